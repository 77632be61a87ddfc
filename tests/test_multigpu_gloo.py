"""world_size-2/3 gloo runs of the N>1 data path on CPU: slab bounds, padded all-gather, assembly.
The slabs themselves are rendered with the oracle here (no GPU); on GPUs the same SlabExchange
carries the HIP renderer's output (bench.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, W, H, ts, out, collective="all_gather", balanced=False):
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd"))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gsplat import multigpu, synth
    from oracle import gs_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gs_oracle.set_num_threads(2)
    s = synth.bicycle_like(6000)
    u = synth.orbit_camera(9, W, H).uniforms(W, H)
    bounds = None
    if balanced:  # slabs of unequal width: columns weighted by the whole frame's instance counts (rank 0 decides)
        ntx = multigpu.num_tile_columns(W, ts)
        tb = torch.zeros(world + 1, dtype=torch.int64)
        if rank == 0:
            full0 = gs_oracle.render(s, u, W, H, ts, want_f32=False)
            tc = np.diff(np.concatenate([[0], full0["ranges"].astype(np.int64)]))
            tb = torch.tensor(multigpu.balanced_bounds(tc[: (tc.size // ntx) * ntx].reshape(-1, ntx).sum(0), world), dtype=torch.int64)
        dist.broadcast(tb, src=0)
        bounds = [int(v) for v in tb.tolist()]
    x = multigpu.SlabExchange(W, H, ts, world, rank, torch.device("cpu"), bounds=bounds, collective=collective)
    r = gs_oracle.render(s, u, W, H, ts, cols=x.cols, want_f32=False)
    b, e = x.pixels[rank]
    slab = np.ascontiguousarray(r["rgba8"][:, b:e])
    x.send[: slab.size] = torch.from_numpy(slab.reshape(-1))
    x.exchange()
    img = x.assemble().numpy() if (collective == "all_gather" or rank == 0) else None
    tot = torch.tensor([r["num_intersections"]], dtype=torch.int64)
    dist.all_reduce(tot)
    if rank == 0:
        full = gs_oracle.render(s, u, W, H, ts, want_f32=False)
        np.save(out, np.array([int(np.array_equal(img, full["rgba8"])), int(tot.item() == full["num_intersections"])]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,ts", [(2, 320, 160, 16), (3, 200, 96, 8)])
def test_slab_exchange_gloo(tmp_path, world, W, H, ts):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, W, H, ts, out), nprocs=world, join=True)
    ok = np.load(out)
    assert ok[0] == 1, "assembled frame differs from the single-device frame"
    assert ok[1] == 1, "slab intersection counts do not add up"


@pytest.mark.parametrize("world,W,H,ts", [(2, 320, 160, 16), (3, 200, 96, 8)])
def test_root_gather_and_balanced_slabs_gloo(tmp_path, world, W, H, ts):
    """The collective bench.py uses at N > 1 (slabs to the presenting rank only) over slabs of unequal width."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    port = 29500 + (os.getpid() % 2000) + 20 + world
    mp.spawn(_worker, args=(world, port, W, H, ts, out, "gather", True), nprocs=world, join=True)
    ok = np.load(out)
    assert ok[0] == 1, "assembled frame differs from the single-device frame"
    assert ok[1] == 1, "slab intersection counts do not add up"


def test_balanced_bounds():
    from itertools import combinations
    from gsplat import multigpu
    rng = np.random.default_rng(5)
    assert multigpu.balanced_bounds(np.ones(120), 8) == multigpu.slab_bounds(1920, 16, 8)
    for ntx, world in [(7, 3), (9, 4), (12, 5), (10, 1), (6, 6)]:
        cost = rng.integers(0, 50, ntx).astype(np.float64)
        b = multigpu.balanced_bounds(cost, world)
        assert b[0] == 0 and b[-1] == ntx and len(b) == world + 1 and all(x < y for x, y in zip(b, b[1:]))
        got = max(cost[x:y].sum() for x, y in zip(b, b[1:]))
        best = min(max(cost[x:y].sum() for x, y in zip((0,) + c, c + (ntx,))) for c in combinations(range(1, ntx), world - 1))
        assert got == best
    hill = np.exp(-0.5 * ((np.arange(120) - 60) / 25.0) ** 2)  # a scene's centre columns carry most instances
    b = multigpu.balanced_bounds(hill, 8)
    w = np.diff(b)
    assert w[0] > w[3] and w[-1] > w[4], "border slabs must come out wider than centre slabs"
    with pytest.raises(ValueError):
        multigpu.balanced_bounds(np.ones(4), 5)


def test_slab_bounds():
    from gsplat import multigpu
    assert multigpu.slab_bounds(1920, 16, 8) == [0, 15, 30, 45, 60, 75, 90, 105, 120]
    assert multigpu.slab_bounds(3840, 16, 8)[1] == 30
    b = multigpu.slab_bounds(200, 16, 3)  # 13 columns
    assert b[0] == 0 and b[-1] == 13 and all(x < y for x, y in zip(b, b[1:]))
    assert multigpu.slab_pixels(b, 200, 16)[-1][1] == 200
    with pytest.raises(ValueError):
        multigpu.slab_bounds(64, 16, 5)


def _gpu_worker(rank, world, port, W, H, ts, out):
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd"))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import gsplat
    from gsplat import _abi, multigpu, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = synth.bicycle_like(30000)
    u = synth.orbit_camera(9, W, H).uniforms(W, H)
    x = multigpu.SlabExchange(W, H, ts, world, rank, torch.device("cpu"))
    r = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, gsplat.PackedGaussians(s), ts, cols=x.cols, flags=_abi.GS_FLAG_EXACT_BLEND)
    r.render_uniforms(u)
    r.wait()
    slab = r.read_rgba8()
    x.send[: slab.size] = torch.from_numpy(slab.reshape(-1))
    x.exchange()
    img = x.assemble().numpy()
    tot = torch.tensor([r.stats()["num_intersections"]], dtype=torch.int64)
    dist.all_reduce(tot)
    r.destroy()
    if rank == 0:
        full = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, gsplat.PackedGaussians(s), ts, flags=_abi.GS_FLAG_EXACT_BLEND)
        full.render_uniforms(u)
        full.wait()
        ok = [int(np.array_equal(img, full.read_rgba8())), int(tot.item() == full.stats()["num_intersections"])]
        full.destroy()
        np.save(out, np.array(ok))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_slab_ranks_on_one_gpu(tmp_path):
    """Three processes share the one GPU of the test box (<= 6 allowed), each renders its tile-column slab
    with the HIP path; the gathered frame equals the single-process frame byte for byte."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    port = 29500 + (os.getpid() % 2000) + 7
    mp.spawn(_gpu_worker, args=(3, port, 400, 208, 16, out), nprocs=3, join=True)
    ok = np.load(out)
    assert ok[0] == 1 and ok[1] == 1


def _gpu_stream_worker(rank, world, port, W, H, ts, out, overlap):
    """The flow of bench.py --gpus N: blend straight into the send buffer (gs_render_to), all-gather, device-side assembly,
    all ordered by one created torch stream, several frames back to back without host waits in between."""
    import torch
    import torch.distributed as dist
    import gsplat
    from gsplat import multigpu, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.cuda.set_stream(torch.cuda.Stream(dev))  # not the default stream: its handle 0 means "own stream" to gs_create
    s = synth.bicycle_like(60000)
    us = [synth.orbit_camera(k, W, H).uniforms(W, H) for k in range(5)]
    b = multigpu.slab_bounds(W, ts, world)
    r = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, gsplat.PackedGaussians(s), ts, cols=(b[rank], b[rank + 1]),
                        stream=torch.cuda.current_stream(dev).cuda_stream)
    x = multigpu.SlabExchange(W, H, ts, world, rank, dev, renderer=r)
    ovl = multigpu.OverlappedExchange(x, torch.cuda.current_stream(dev)) if overlap else None
    for u in us:  # no host synchronisation between the frames
        if ovl is not None:  # all-gather on a side stream, overlapped with the next frame
            r.render_uniforms(u, out_ptr=ovl.send_ptr())
            ovl.submit(assemble=(rank == 0))
        else:
            r.render_uniforms(u, out_ptr=x.send.data_ptr())
            x.exchange()
            if rank == 0:
                x.assemble()
    if ovl is not None:
        ovl.finish(assemble=(rank == 0))
    r.wait()
    torch.cuda.synchronize(dev)
    dist.barrier()
    if rank == 0:
        img = x.image.cpu().numpy()
        full = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, gsplat.PackedGaussians(s), ts)
        full.render_uniforms(us[-1])
        full.wait()
        np.save(out, np.array([int(np.array_equal(img, full.read_rgba8()))]))
        full.destroy()
    r.destroy()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [False, True])
def test_device_side_exchange_is_ordered_with_the_frame(tmp_path, overlap):
    """bench.py's N>1 step on one GPU with two processes: the gathered, assembled LAST frame equals the whole-canvas frame
    (it would be a stale or torn slab if the collective were not ordered after the blend)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    port = 29500 + (os.getpid() % 2000) + 11
    mp.spawn(_gpu_stream_worker, args=(2, port + int(overlap), 640, 368, 16, out, overlap), nprocs=2, join=True)
    assert np.load(out)[0] == 1


def _gpu_pipeline_worker(rank, world, port, W, H, ts, out, K):
    """bench.py's N > 1 step: K slab contexts per rank in flight (multigpu.PipelinedSlabs), slabs balanced by instance count,
    gather to rank 0, assembly on the communication stream; seven frames without a host wait."""
    import torch
    import torch.distributed as dist
    import gsplat
    from gsplat import _abi, multigpu, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    s = synth.bicycle_like(60000)
    us = [synth.orbit_camera(k, W, H).uniforms(W, H) for k in range(7)]
    pg = gsplat.PackedGaussians(s)
    owner = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, ts)
    ntx = multigpu.num_tile_columns(W, ts)
    tb = torch.zeros(world + 1, dtype=torch.int64)
    if rank == 0:
        owner.render_uniforms(us[0])
        owner.wait()
        tc = np.diff(np.concatenate([[0], owner.read_buffer(_abi.GS_BUF_RANGES).astype(np.int64)])).astype(np.float64)
        tb = torch.tensor(multigpu.balanced_bounds(tc[: (tc.size // ntx) * ntx].reshape(-1, ntx).sum(0), world), dtype=torch.int64)
    dist.broadcast(tb, src=0)
    bounds = [int(v) for v in tb.tolist()]
    x = multigpu.SlabExchange(W, H, ts, world, rank, dev, bounds=bounds, collective="gather")
    mk = lambda stream, share: gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, ts, cols=x.cols, stream=stream, share_with=share)
    pipe = multigpu.PipelinedSlabs(x, mk, K, owner=owner)
    for u in us:
        pipe.submit(u)
    pipe.finish()
    torch.cuda.synchronize(dev)
    dist.barrier()
    if rank == 0:
        owner.render_uniforms(us[-1])
        owner.wait()
        np.save(out, np.array([int(np.array_equal(x.image.cpu().numpy(), owner.read_rgba8())), int(len(set(np.diff(bounds))) > 1)]))
    pipe.destroy()
    owner.destroy()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("K", [1, 3])
def test_pipelined_slabs_on_one_gpu(tmp_path, K):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    port = 29500 + (os.getpid() % 2000) + 30 + K
    mp.spawn(_gpu_pipeline_worker, args=(2, port, 640, 368, 16, out, K), nprocs=2, join=True)
    ok = np.load(out)
    assert ok[0] == 1, "the assembled last frame differs from the whole-canvas frame"


def _gpu_rccl_worker(rank, world, port, W, H, ts, out, collective):
    """The N > 1 step over RCCL itself with a world of one rank (a one-GPU box cannot hold two RCCL ranks): the collective
    is really called on the communication stream (always_collective), three slab contexts in flight."""
    import torch
    import torch.distributed as dist
    import gsplat
    from gsplat import multigpu, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    s = synth.bicycle_like(60000)
    us = [synth.orbit_camera(k, W, H).uniforms(W, H) for k in range(8)]
    pg = gsplat.PackedGaussians(s)
    owner = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, ts)
    tb = torch.tensor(multigpu.slab_bounds(W, ts, 1), dtype=torch.int64, device=dev)
    dist.broadcast(tb, src=0)
    x = multigpu.SlabExchange(W, H, ts, 1, 0, dev, bounds=[int(v) for v in tb.tolist()], collective=collective)
    x.always_collective = True
    mk = lambda stream, share: gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, ts, cols=x.cols, stream=stream, share_with=share)
    pipe = multigpu.PipelinedSlabs(x, mk, 3, owner=owner)
    for u in us:
        pipe.submit(u)
    pipe.finish()
    torch.cuda.synchronize(dev)
    dist.barrier()
    owner.render_uniforms(us[-1])
    owner.wait()
    ok = int(np.array_equal(x.image.cpu().numpy(), owner.read_rgba8()))
    t = torch.tensor([ok], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    np.save(out, np.array([int(t.item())]))
    pipe.destroy()
    owner.destroy()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("collective", ["gather", "all_gather"])
def test_pipelined_slabs_over_rccl_world_of_one(tmp_path, collective):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    port = 29500 + (os.getpid() % 2000) + 40 + len(collective)
    mp.spawn(_gpu_rccl_worker, args=(1, port, 640, 368, 16, out, collective), nprocs=1, join=True)
    assert np.load(out)[0] == 1, "the frame gathered and assembled over RCCL differs from the whole-canvas frame"


def _gpu_groups_worker(rank, world, port, W, H, ts, out, groups, backend="gloo"):
    """multigpu.FrameGroupSlabs: groups of ranks render alternate frames, every frame = world/groups slabs gathered to rank 0."""
    import torch
    import torch.distributed as dist
    import gsplat
    from gsplat import _abi, multigpu, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    s = synth.bicycle_like(60000)
    us = [synth.orbit_camera(k, W, H).uniforms(W, H) for k in range(9)]
    pg = gsplat.PackedGaussians(s)
    owner = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, ts)
    S = world // groups
    bounds = multigpu.slab_bounds(W, ts, S)
    cols = (bounds[rank % S], bounds[rank % S + 1])
    mk = lambda stream, share: gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, ts, cols=cols, stream=stream, share_with=share)
    fg = multigpu.FrameGroupSlabs(W, H, ts, world, rank, dev, groups, bounds, mk, frames_in_flight=2, owner=owner)
    ok = 1
    for k, u in enumerate(us):
        fg.submit(u)
        if k in (4, len(us) - 1):  # an odd and an even frame: both groups' frames must reach rank 0 intact
            fg.finish()
            torch.cuda.synchronize(dev)
            dist.barrier()
            if rank == 0:
                owner.render_uniforms(u)
                owner.wait()
                ok &= int(np.array_equal(fg.x.image.cpu().numpy(), owner.read_rgba8()))
    if rank == 0:
        np.save(out, np.array([ok]))
    fg.destroy()
    owner.destroy()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_frame_groups_on_one_gpu(tmp_path):
    """Four processes on the one GPU: two groups of two slabs render alternate frames."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    port = 29500 + (os.getpid() % 2000) + 50
    mp.spawn(_gpu_groups_worker, args=(4, port, 640, 368, 16, out, 2), nprocs=4, join=True)
    assert np.load(out)[0] == 1, "a frame gathered from a frame group differs from the whole-canvas frame"


@pytest.mark.gpu
def test_frame_groups_over_rccl_world_of_one(tmp_path):
    """FrameGroupSlabs' own code -- dist.new_group, the gather on a sub-communicator, the assembly on the communication stream --
    on the nccl (= RCCL) backend with a world of one rank: all the one-GPU box can execute of it (more ranks over RCCL need more
    GPUs; over gloo: test_frame_groups_on_one_gpu)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    port = 29500 + (os.getpid() % 2000) + 60
    mp.spawn(_gpu_groups_worker, args=(1, port, 640, 368, 16, out, 1, "nccl"), nprocs=1, join=True)
    assert np.load(out)[0] == 1, "the frame gathered by FrameGroupSlabs over RCCL differs from the whole-canvas frame"
