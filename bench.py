#!/usr/bin/env python3
"""bench.py -- frames/s of the forward splat-render path on N MI355X (BASELINE.json metric).

A "step" is one whole frame (process_gaussians -> scan -> write_tile_ids -> radix sort -> ranges ->
blend; for N > 1 plus the RCCL all-gather of the per-rank tile-column slabs and their assembly) for
a camera that moves every step along a fixed orbit, with the splats already resident in HBM.

    python bench.py                       # N=1, config B: 6.1 M splats @ 1920x1080
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  The per-stage device times come from hipEvents that the library
records on its own stream inside the timed region (GS_FLAG_TIMING).  After the timed region (N = 1) the
last camera is rendered again in the other blend mode and with the reference's binning (self_check), and
the CPU oracle renders ONE whole frame of the same scene and camera on the host cores: that is both the
cpu_baseline (a reported baseline, not a target) and the check of the frame the bench timed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
VALU_PEAK_TFLOPS = 157.3   # peak vector fp32 (MI355X_MICROARCH.md)

CONFIGS = {
    # BASELINE.json configs[1], [2], [4]; "bicycle-like" synthetic, no real scene offline
    "B": dict(n=6_100_000, width=1920, height=1080, name="bicycle-like synthetic 6.1M gaussians @1920x1080"),
    "C": dict(n=6_100_000, width=3840, height=2160, name="bicycle-like synthetic 6.1M gaussians @3840x2160"),
    "E": dict(n=50_000_000, width=1920, height=1080, name="bicycle-like synthetic 50M gaussians @1920x1080"),
    "A": dict(n=10_000, width=256, height=256, name="bicycle-like synthetic 10k gaussians @256x256"),
}


def algorithmic_bytes(st, W, H, T):
    """SURVEY.md 8(d) per-frame algorithmic bytes, per stage, verbatim: N gaussians, Nv visible, I instances, Ip staged entries, T
    tiles, P pixels, p = sort passes over the INSTANCES the build really executes (the tight row pipeline writes every instance
    once: p = 1; its pass over the row items is moved bytes of "emit", below, not algorithmic bytes)."""
    N, Nv, I, Ip, p = st["num_gaussians"], st["num_visible"], st["num_intersections"], st["num_processed"], st["sort_passes"]
    return {
        "preprocess": 12 * (N - Nv) + 236 * Nv + 4 * N + 56 * Nv,
        "scan": 8 * N,
        "emit": 24 * Nv + 8 * I,
        "sort": (4 + 16 * p) * I,
        "ranges": 4 * I + 4 * T,
        "blend": 40 * Ip + 4 * W * H,
    }


def moved_bytes(st, W, H, T, tile16):
    """What THIS build's kernels read + write per frame (beside SURVEY 8(d)'s figures).
    Tight row pipeline (S row-item slots, R row items, NT = N / 2048 sort chunks, C = R / 512 + rows expansion chunks):
    projection reads the 12-byte position of every gaussian and 32 + 192 bytes of a visible one, writes the count word, the 64-byte
    GaussianData, the arena address and 12 bytes per slot; the gaussian-level sort reads the count words twice and the addresses once,
    writes / scans / reads its 8 KB-per-chunk table and writes a 16-byte record per visible gaussian; the row sort reads those and
    the slots and writes 12 bytes per item; count / scan / expansion read the items twice, a 1 KB table per chunk three times and
    write 4 bytes per instance (and the ranges); the blend reads 4 bytes per staged entry per walker that stages it (<= 4) and 40
    bytes of GaussianData per (walker, entry) it evaluates, and writes 4 bytes per pixel.
    Reference binning, depth-ordered on 16-bit tile ids: as round 2 (emission 6 bytes per instance, 12 per pair and sweep)."""
    N, Nv, I, Ip, p, Ev = st["num_gaussians"], st["num_visible"], st["num_intersections"], st["num_processed"], st["sort_passes"], st["num_evaluated"]
    if st.get("tight_binning"):
        S, R = st["num_row_slots"], st["num_row_items"]
        NT, C = (N + 2047) // 2048, R // 512 + H // 16 + 1
        return {"preprocess": 12 * N + 224 * Nv + 4 * N + 68 * Nv + 12 * S, "scan": 12 * N + 4 * 8192 * NT + 16 * Nv,
                "emit": 16 * Nv + 12 * S + 12 * R, "sort": 24 * R + 3 * 1024 * C + 4 * I + 4 * T, "ranges": 0,
                "blend": 16 * Ip + 40 * Ev + 4 * W * H}
    if st.get("depth_ordered") and tile16:
        NT = (N + 2047) // 2048
        return {"preprocess": 12 * N + 224 * Nv + 4 * N + 64 * Nv, "scan": 8 * N + 4 * 8192 * NT + 16 * Nv,
                "emit": 64 * Nv + 6 * I, "sort": 12 * p * I, "ranges": 2 * I + 4 * T, "blend": 16 * Ip + 40 * Ev + 4 * W * H}
    return None


STAGE_KERNELS = {"preprocess": ["gs_preprocess_kernel"], "scan": ["gs_gsort_scatter_kernel", "gs_scan_kernel"], "emit": ["gs_rows_sort_kernel", "gs_emit_balanced_kernel", "gs_emit_kernel"],
                 "sort": ["gs_rows_expand_kernel", "gs_sort_sweep_kernel<unsigned short>", "gs_sort_sweep_kernel<unsigned int>", "gs_sort_sweep_kernel"],
                 "ranges": ["gs_ranges16_kernel", "gs_ranges_kernel"],
                 "blend": ["gs_blend_quad_kernel", "gs_blend_wave_kernel", "gs_blend_kernel"]}


PMC_FILE = os.path.join("profiles", "r03_pmc.json")


def pmc_traffic(stage, workload):
    """HBM bytes per launch of the stage's main kernel.  NOT measured by this run: PMC counters need their own rocprofv3
    passes (gpurun refuses them beside tracing), so the figure is replayed from the committed passes of tools/pmc_run.sh
    on the same workload (PMC_FILE; the `source` key of the result says so).  MI355X_MICROARCH.md (HBM): bytes =
    (FETCH_SIZE + WRITE_SIZE) * 1024, and on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced stream (16 bytes per lane),
    so it is doubled for the streaming stages -- but NOT for the blend: its reads are gathers of 8 + 12 + 16 bytes of a 64-byte record,
    and tools/microbench/fetch_calib.hip (known bytes: profiles/r03_fetch_calib.txt) shows FETCH_SIZE counts those at 1.003x."""
    path = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(path):
        return None, None
    js = json.load(open(path))
    if js.get("workload") != workload:
        return None, None
    for k in STAGE_KERNELS[stage]:
        c = next((v for name, v in js["kernels"].items() if name.startswith(k)), None)  # template arguments follow the name
        if c and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            ff = 1.0 if stage == "blend" else 2.0
            how = ("FETCH_SIZE + WRITE_SIZE (record gathers are counted at 1.003x known bytes: profiles/r03_fetch_calib.txt)" if stage == "blend"
                   else "2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE half-count for 16 B/lane streams)")
            return (ff * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, {"kernel": k, "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
                                                                       "correction": how,
                                                                       "source": PMC_FILE + " (committed rocprofv3 --pmc passes of the same workload, not this run)"}
    return None, None


def cpu_baseline(host_scene, n_full, W, H, ts, uniforms, max_n):
    """Times the CPU oracle (oracle/gs_oracle.c, OpenMP over the box's host cores) on ONE whole frame of the SAME scene the
    GPU rendered (the device tensor copied to the host), same camera as the last timed frame.  Scenes above max_n gaussians
    are cut to their first max_n (stated in `sample`) so that the default run stays within minutes."""
    from oracle import gs_oracle
    gs_oracle.build()
    n = min(max_n, n_full)
    splats = host_scene[:n]
    cores = gs_oracle.get_num_threads()
    t0 = time.perf_counter()
    out = gs_oracle.render(splats, uniforms, W, H, ts, want_f32=False)
    dt = time.perf_counter() - t0
    res = {
        "value": 1.0 / dt, "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": "1 whole frame of %s, %dx%d, the camera of the last timed step; %d intersections (reference binning); "
                  "CPU restatement of the reference pipeline (oracle/gs_oracle.c, OpenMP, no early exit: the reference has none)"
                  % ("the same scene, all %d gaussians" % n_full if n == n_full else "the FIRST %d of %d gaussians of the same scene" % (n, n_full),
                     W, H, out["num_intersections"]),
        "seconds_per_frame": dt,
    }
    return res, (out if n == n_full else None)


def flight_pass(r, _abi, uniforms, args, k):
    """The same K steps with the context limited to k frames in flight (GS_OPT_FRAMES_IN_FLIGHT), outside the timed region."""
    r.set_option(_abi.GS_OPT_FRAMES_IN_FLIGHT, k)
    for i in range(args.warmup):
        r.render_uniforms(uniforms[i % 64])
    r.wait()
    r.set_option(_abi.GS_OPT_RESET_TIMING, 0)
    t0 = time.perf_counter()
    for i in range(args.steps):
        r.render_uniforms(uniforms[(args.warmup + i) % 64])
    r.wait()
    dt = time.perf_counter() - t0
    return {"frames_in_flight": k, "value": args.steps / dt, "unit": "frames/s", "ms_per_step": dt / args.steps * 1e3, "steps": args.steps}, r.stats()


_OPT_EMIT_ORDER, _OPT_BLEND_ABLATION = 4, 1  # gs_abi.h GS_OPT_EMIT_ORDER / GS_OPT_BLEND_ABLATION
_KEEP = {}


def self_check(gsplat, _abi, r, W, H, ts, device, u_last, args):
    """Outside the timed region (N = 1): the camera of the last timed step is rendered again (a) by the timed context, (b) by a
    context in the other blend mode (EXACT <-> fused), (c) with the reference's rect binning and the other emission order;
    (b) gives `exact_blend` (frames/s of the bit-exact mode over a short loop) and the fused-vs-exact pixel statistics,
    (c) must give the same bytes as (a)/(b) in the same blend mode.  bench.py's CPU leg then compares the EXACT frame with the
    oracle's frame of the same scene and camera."""
    import numpy as np
    pg = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians)
    pg.numGaussians, pg.gaussiansBuffer = r.numGaussians, None
    r.render_uniforms(u_last)
    r.wait()
    timed_img = r.read_rgba8()
    other_flags = 0 if args.exact else _abi.GS_FLAG_EXACT_BLEND
    o = gsplat.Renderer(gsplat.Canvas(W, H), None, device, pg, ts, flags=other_flags, share_with=r)
    if args.tile_cull >= 0:
        o.set_option(_abi.GS_OPT_TILE_CULL, args.tile_cull)
    k = max(10, min(args.steps, 40))
    us = [u_last] * 3
    for u in us:
        o.render_uniforms(u)
        o.wait()
    from gsplat import synth
    orbit = [synth.orbit_camera(i, W, H).uniforms(W, H) for i in range(64)]
    t0 = time.perf_counter()
    for i in range(k):
        o.render_uniforms(orbit[i % 64])
    o.wait()
    other_fps = k / (time.perf_counter() - t0)
    o.render_uniforms(u_last)
    o.wait()
    other_img = o.read_rgba8()
    exact_img, fused_img = (timed_img, other_img) if args.exact else (other_img, timed_img)
    _KEEP["exact_frame"] = exact_img
    # (c) reference binning + the reference's emission order, same blend mode as `o` (EXACT unless --exact)
    o.set_option(_abi.GS_OPT_TILE_CULL, 0)
    o.set_option(_abi.GS_OPT_EMIT_ORDER, 1)
    o.render_uniforms(u_last)
    o.wait()
    ref_binning_same = bool(np.array_equal(o.read_rgba8(), other_img))
    stref = o.stats()  # the reference's binning of the same camera: its instance and staged-entry counts
    _KEEP["reference_binning"] = {"intersections": stref["num_intersections"], "processed": stref["num_processed"]}
    o.destroy()
    d = np.abs(exact_img[..., :3].astype(np.int32) - fused_img[..., :3].astype(np.int32)).max(axis=2)
    detail = {"fused_vs_exact_pixels_off_by_more_than_1_lsb": int((d > 1).sum()), "fused_vs_exact_max_lsb": int(d.max()),
              "pixels": int(d.size), "tight_binning_frame_equals_reference_binning_frame": ref_binning_same}
    ok = bool(ref_binning_same and (d > 1).mean() <= 0.02)
    name = "fused_blend" if args.exact else "exact_blend"
    return {"frame_verified": ok, "frame_verified_detail": detail,
            name: {"value": other_fps, "unit": "frames/s", "steps": k, "note": "same scene and orbit, the other blend mode, short loop outside the timed region"}}



def reference_binning_leg(gsplat, _abi, r, W, H, ts, device, uniforms, args, eflag):
    """Outside the timed region (N = 1): the same scene, orbit and blend mode rendered with the REFERENCE's binning (every tile of the
    3-sigma rect, process_gaussians.wgsl:74-86) in the reference's emission order (gaussian index, full-key sort): the like-for-like
    figure beside `value`, whose tight row pipeline bins a provably harmless subset of those instances.  Byte-equal frames in the
    exact blend mode (self_check, tests/test_gpu_scale.py)."""
    pg = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians)
    pg.numGaussians, pg.gaussiansBuffer = r.numGaussians, None
    o = gsplat.Renderer(gsplat.Canvas(W, H), None, device, pg, ts, flags=eflag, share_with=r)
    o.set_option(_abi.GS_OPT_TILE_CULL, 0)
    o.set_option(_abi.GS_OPT_EMIT_ORDER, 1)
    out = {"binning": "reference rect, gaussian-index emission order, full-key sort", "unit": "frames/s"}
    for fif, key in ((0, "value"), (1, "one_frame_in_flight")):
        if fif:
            o.set_option(_abi.GS_OPT_FRAMES_IN_FLIGHT, fif)
        for k in range(0, 64, 1 if not fif else 64):  # capacity of every member of the ring for the whole orbit
            o.render_uniforms(uniforms[k])
            o.wait()
        n = max(10, min(args.steps, 60))
        for k in range(5):
            o.render_uniforms(uniforms[k])
        o.wait()
        t0 = time.perf_counter()
        for k in range(n):
            o.render_uniforms(uniforms[(args.warmup + k) % 64])
        o.wait()
        dt = time.perf_counter() - t0
        out[key] = n / dt
        out["ms_per_step" if not fif else "one_frame_ms_per_step"] = dt / n * 1e3
        out["steps"] = n
    st = o.stats()
    out.update(intersections=st["num_intersections"], sort_passes=st["sort_passes"], frames_in_flight=3)
    o.destroy()
    return out


def copy_probe(dev, nbytes=512 << 20, iters=10):
    """Device-to-device copy of `nbytes` (outside the timed region): the practical HBM ceiling of this box, reported
    beside the 8 TB/s nominal peak the roofline fractions are quoted against."""
    import torch
    a = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
    b = torch.empty_like(a)
    a.zero_()
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    us = e0.elapsed_time(e1) * 1e3 / iters
    del a, b
    return {"bytes_copied": nbytes, "us": round(us, 1), "GBps_read_plus_write": round(2 * nbytes / (us * 1e-6) / 1e9, 1),
            "nominal_peak_GBps": HBM_PEAK_GBS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="B", choices=sorted(CONFIGS))
    ap.add_argument("--gaussians", type=int, default=0, help="override the gaussian count")
    ap.add_argument("--tile", type=int, default=16)
    ap.add_argument("--blend-ablation", type=int, default=0, help="profiling only: see GS_OPT_BLEND_ABLATION")
    ap.add_argument("--emit-order", type=int, default=-1, help="GS_OPT_EMIT_ORDER override (0 depth-ordered, 1 index order, 2 auto = default)")
    ap.add_argument("--grid", type=int, default=0, help="GS_OPT_PERSISTENT_GRID override (workgroups of the ticket-loop kernels)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0 (with --backend gloo)")
    ap.add_argument("--no-timing", action="store_true", help="do not bracket stages with hipEvents")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-max", type=int, default=10_000_000, help="CPU baseline: render the whole scene on the host unless it has more gaussians than this")
    ap.add_argument("--exact", action="store_true", help="time the bit-exact blend (GS_FLAG_EXACT_BLEND) instead of the default fused one")
    ap.add_argument("--tile-cull", type=int, default=-1, help="GS_OPT_TILE_CULL override (1 = tight binning, default; 0 = the reference's rect binning)")
    ap.add_argument("--collective", default="gather", choices=["gather", "all_gather"],
                    help="N>1: slabs go to the presenting rank only (RCCL send/recv to rank 0, default) or to every rank")
    ap.add_argument("--graph", type=int, default=-1,
                    help="GS_OPT_FRAME_GRAPH (1: replay the captured frame with one hipGraphLaunch; contexts with stage timing issue "
                         "their launches directly whatever this says).  Default 0: measured no faster on this stack (config A 14.1 k "
                         "vs 14.6 k frames/s: hipGraphLaunch of 16 nodes costs the host what 16 launches do)")
    ap.add_argument("--force-multi", action="store_true", help="rehearsal: run the N > 1 code path with a world of one rank")
    ap.add_argument("--frame-groups", type=int, default=0,
                    help="N>1: G groups of N/G ranks render alternate frames, every frame being N/G column slabs gathered to "
                         "rank 0 (multigpu.FrameGroupSlabs).  1 = every frame is N column slabs (SURVEY 8e as written).  0 (default) = "
                         "automatic: N/2 groups of two ranks from 4 GPUs on, else 1 -- a rank's frame has a fixed cost (the O(N) cull, "
                         "launches too small to fill the chip), so two wide slabs per frame use 8 GPUs better than eight narrow ones "
                         "(one-GPU projection at 1080p: 6.7x against 3.6x, profiles/r03_slab_per_rank.txt)")
    ap.add_argument("--even-slabs", action="store_true",
                    help="N>1: equal tile-column slabs instead of slabs balanced by the instance counts of a calibration pass")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run checks (N>1: assembled frame vs a whole-canvas render; N=1: self_check)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames in flight in the timed region (0 = default: 3).  N=1: GS_OPT_FRAMES_IN_FLIGHT of the context; N>1: "
                         "slab contexts per rank (multigpu.PipelinedSlabs).  The same steps are repeated outside the timed region "
                         "strictly one frame after the other (`one_frame_in_flight`), which is also where `stages` and `roofline` "
                         "are measured")
    ap.add_argument("--ply", default=os.environ.get("GS_PLY", ""),
                    help="render this 3DGS .ply (native loader) instead of the synthetic scene; also taken from $GS_PLY (SURVEY.md 8d)")
    ap.add_argument("--proj-chunks", type=int, default=0, help="GS_OPT_PROJ_CHUNKS (tuning): cull chunks per workgroup of the tight projection")
    ap.add_argument("--lib", default="", help="A/B only: another build of libgsplat_hip.so (sets $GSPLAT_LIB)")
    args = ap.parse_args()
    if args.lib:
        os.environ["GSPLAT_LIB"] = os.path.abspath(args.lib)

    import numpy as np
    import torch
    import torch.distributed as dist

    import gsplat
    from gsplat import _abi, synth

    cfg = dict(CONFIGS[args.config])
    if args.gaussians:
        cfg["n"] = args.gaussians
    N, W, H, ts = cfg["n"], cfg["width"], cfg["height"], args.tile
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    # --force-multi: the N > 1 code path (slab contexts in flight, collective on the communication stream, assembly) in a
    # single process with a world of one: the only way to put the RCCL calls through their paces on a one-GPU box
    multi = world > 1 or args.force_multi
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:  # --force-multi without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29571"), RANK="0", WORLD_SIZE="1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)

    seed = synth.BASE_SEED + {"A": 0, "B": 1, "C": 2, "E": 4}[args.config]
    from gsplat import multigpu

    ply_records = None
    if args.ply:  # a real scene when the box has one (never the case offline)
        ply_records = gsplat.PackedGaussians.from_ply(args.ply).gaussiansBuffer
        N = cfg["n"] = int(ply_records.shape[0])
        cfg["name"] = "%s (%d gaussians) @%dx%d" % (os.path.basename(args.ply), N, W, H)
        splats = torch.from_numpy(ply_records).to(dev)
    else:
        splats = synth.bicycle_like_torch(N, seed, dev)  # every rank holds the full replica
    eflag = _abi.GS_FLAG_EXACT_BLEND if args.exact else 0
    flags = (0 if args.no_timing else _abi.GS_FLAG_TIMING) | eflag
    pg = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians)
    pg.numGaussians, pg.gaussiansBuffer, pg.sphericalHarmonicsDegree = N, splats, 3
    # the context that uploads the splats.  N = 1: the one that is timed.  N > 1: a whole-canvas context that only holds the
    # replica, calibrates the slab bounds and renders the frame the assembled slabs are checked against; the timed slab
    # contexts borrow its splats.
    r = gsplat.Renderer(gsplat.Canvas(W, H), None, local_rank, pg, ts, flags=flags if not multi else eflag)
    keep_scene = not multi and not args.no_cpu and ply_records is None  # the CPU baseline renders the same bits
    host_scene = splats.cpu().numpy() if keep_scene else ply_records
    del splats
    pg.gaussiansBuffer = None
    torch.cuda.empty_cache()

    def options(rr):
        if args.tile_cull >= 0:
            rr.set_option(_abi.GS_OPT_TILE_CULL, args.tile_cull)
        if args.grid:
            rr.set_option(_abi.GS_OPT_PERSISTENT_GRID, args.grid)
        if args.emit_order >= 0:
            rr.set_option(_abi.GS_OPT_EMIT_ORDER, args.emit_order)
        if args.blend_ablation:
            rr.set_option(_abi.GS_OPT_BLEND_ABLATION, args.blend_ablation)
        if args.graph >= 0:
            rr.set_option(_abi.GS_OPT_FRAME_GRAPH, args.graph)
        if args.proj_chunks:
            rr.set_option(_abi.GS_OPT_PROJ_CHUNKS, args.proj_chunks)

    options(r)
    if args.frames_in_flight > 0 and not multi:
        r.set_option(_abi.GS_OPT_FRAMES_IN_FLIGHT, args.frames_in_flight)

    uniforms = [synth.orbit_camera(k, W, H).uniforms(W, H) for k in range(64)]
    owner = xch = pipe = None
    if args.frame_groups == 0:
        args.frame_groups = world // 2 if (multi and world >= 4 and world % 2 == 0) else 1
    if args.frame_groups < 1 or world % args.frame_groups:
        raise SystemExit("--frame-groups must divide the number of ranks")
    nslabs = world // args.frame_groups  # slabs per frame
    bounds = multigpu.slab_bounds(W, ts, nslabs)  # tile-column slabs
    if multi:
        owner = r
        if not args.even_slabs:
            # slabs balanced by load: instances per tile column, summed over 8 cameras of the orbit, from whole-canvas frames
            # (rank 0 decides, everybody takes its answer)
            ntx_b = multigpu.num_tile_columns(W, ts)
            col = np.zeros(ntx_b, dtype=np.float64)
            if rank == 0:
                for k in range(0, 64, 8):
                    owner.render_uniforms(uniforms[k])
                    owner.wait()
                    tc = np.diff(np.concatenate([[0], owner.read_buffer(_abi.GS_BUF_RANGES).astype(np.int64)])).astype(np.float64)
                    col += tc[: (tc.size // ntx_b) * ntx_b].reshape(-1, ntx_b).sum(axis=0)  # instances per tile, summed over the rows
                bounds = multigpu.balanced_bounds(col, world // args.frame_groups)
            tb = torch.tensor(bounds, dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
            dist.broadcast(tb, src=0)
            bounds = [int(v) for v in tb.tolist()]
        cols = (bounds[rank % nslabs], bounds[rank % nslabs + 1])

        def make_slab_renderer(stream_handle, share_with, fl=eflag):
            rr = gsplat.Renderer(gsplat.Canvas(W, H), None, local_rank, pg, ts, flags=fl, cols=cols, stream=stream_handle, share_with=share_with)
            options(rr)
            return rr

        # K frames in flight per rank: K slab contexts on their own streams, collectives in frame order on one more stream
        K = args.frames_in_flight if args.frames_in_flight > 0 else 3
        if args.frame_groups > 1:
            pipe = multigpu.FrameGroupSlabs(W, H, ts, world, rank, dev, args.frame_groups, bounds, make_slab_renderer, K, owner=owner)
            xch = pipe.x
        else:
            xch = multigpu.SlabExchange(W, H, ts, world, rank, dev, bounds=bounds, collective=args.collective)
            xch.always_collective = args.force_multi  # a world of one would otherwise copy instead of calling the collective
            pipe = multigpu.PipelinedSlabs(xch, make_slab_renderer, K, owner=owner)
        r = pipe.renderers[0]

    def step(k):
        u = uniforms[k % 64]
        if not multi:
            r.render_uniforms(u)
        else:
            pipe.submit(u)  # blend into the send buffer, gather to the presenting rank, assembly there: all enqueued, no host wait

    trouble = {}

    def sync():
        try:
            if pipe is not None:
                pipe.finish()
            else:
                r.wait()
        except _abi.GsError as e:
            if e.code != -9:  # GS_ERR_TRUNCATED: frames of this batch were rendered from truncated lists (capacity now grown)
                raise
            trouble["truncated"] = str(e)
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # capacity calibration, outside warm-up and timing: every camera of the orbit once, waited for one by one, so that the
    # (key,value) arrays have grown to the largest frame of the orbit before frames are enqueued back to back (a frame that
    # overflows while others are queued behind it cannot be re-rendered: gs_wait reports GS_ERR_TRUNCATED)
    for rr in (pipe.renderers if pipe is not None else [r]):
        for k in range(64):
            rr.render_uniforms(uniforms[k])
            rr.wait()
    for k in range(args.warmup):
        step(k)
    sync()
    trouble.clear()
    r.set_option(_abi.GS_OPT_RESET_TIMING, 0)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    sync()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = r.stats()

    slab_strict = None
    if multi and rank == 0 and not args.no_timing:
        # outside the timed region: rank 0's slab strictly one frame after the other on a context with per-stage hipEvents (no
        # exchange): where `stages` and `roofline` of an N > 1 line come from
        rt = make_slab_renderer(None, owner, flags)
        for k in range(64):
            rt.render_uniforms(uniforms[k])
            rt.wait()
        slab_strict = flight_pass(rt, _abi, uniforms, args, 1)
        rt.destroy()

    verified = None
    if multi and not args.no_verify:
        # every rank: the slab it sent last (still in its send buffer) against its columns of a whole-canvas render here
        last_k = args.warmup + args.steps - 1
        own_k, own_slot = (pipe.last_own if args.frame_groups > 1 else (last_k, last_k % pipe.K))  # the last frame THIS rank rendered
        owner.render_uniforms(uniforms[own_k % 64])
        owner.wait()
        whole_local = torch.from_numpy(owner.read_rgba8())
        b0, e0 = xch.pixels[rank % nslabs]
        mine = pipe.send[own_slot][: H * (e0 - b0) * 4].view(H, e0 - b0, 4).cpu()
        slab_ok = bool(torch.equal(mine, whole_local[:, b0:e0]))
        if not slab_ok:
            print("verify: rank %d: its own slab differs from the whole-canvas render in %d pixels" % (rank, int((mine != whole_local[:, b0:e0]).any(dim=2).sum())), file=sys.stderr, flush=True)
        ok_t = torch.tensor([int(slab_ok)], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        slabs_ok = bool(int(ok_t.item()))
    if multi and rank == 0 and not args.no_verify:
        # outside the timed region: the assembled last frame must equal a whole-canvas render of the same camera on this GPU
        # (the slabs only filter the key emission: byte-for-byte equality, tests/test_gpu_parity.py::test_slab_union...)
        full = owner
        full.render_uniforms(uniforms[(args.warmup + args.steps - 1) % 64])
        full.wait()
        whole = torch.from_numpy(full.read_rgba8())
        got = xch.image.cpu()
        verified = bool(torch.equal(whole, got))
        if not verified:
            d = (whole != got).any(dim=2)
            bad = torch.nonzero(d.any(dim=0)).flatten()
            print("verify: %d differing pixels, columns %d..%d, slab pixel bounds %s" % (int(d.sum()), int(bad.min()), int(bad.max()), xch.pixels), file=sys.stderr, flush=True)
            print("verify: per-slab differing pixels vs the last frame: %s" % [int(d[:, b:e].sum()) for b, e in xch.pixels], file=sys.stderr, flush=True)

    if multi:
        # whole-frame statistics are the sums over the slabs
        v = torch.tensor([st["num_visible"], st["num_intersections"], st["num_processed"]], dtype=torch.int64, device=dev)
        per = torch.zeros((world, 3), dtype=torch.int64, device=dev)
        per[rank] = v
        dist.all_reduce(per)  # (one row per rank; a sum of one-hot rows is a gather every backend has)
        dist.all_reduce(v)
        tot_vis, tot_I, tot_Ip = (int(x) // args.frame_groups for x in v.tolist())  # (each group's ranks describe a frame of their own)
        per_rank = [{"rank": g, "columns": [bounds[g % nslabs], bounds[g % nslabs + 1]], "visible": int(per[g][0]), "intersections": int(per[g][1])} for g in range(world)]
        tr = torch.tensor([1 if trouble else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        if int(tr.item()) and not trouble:
            trouble["truncated"] = "a rank other than 0 rendered frames from truncated lists"
    else:
        tot_vis, tot_I, tot_Ip = st["num_visible"], st["num_intersections"], st["num_processed"]

    if rank == 0:
        T = st["num_tiles"]
        line = {
            "metric": "frames/sec", "value": args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "ply" if args.ply else "synthetic",
            # frames are enqueued back to back (no host wait inside the timed region, one gs_wait at its end); the context keeps
            # this many of them in flight (GS_OPT_FRAMES_IN_FLIGHT: a shadow context with its own stream and per-frame arrays)
            "frames_in_flight": pipe.K if pipe is not None else st["frames_in_flight"],
            "config": {"workload": cfg["name"], "gaussians": N, "width": W, "height": H, "tile_size": ts,
                       "parallelism": ("%stile-column slabs x%d (%s, bounds %s) + RCCL %s of the rgba8 slabs to rank 0, %d frames in flight per rank"
                                       % (("%d groups of ranks rendering alternate frames, each frame: " % args.frame_groups) if args.frame_groups > 1 else "",
                                          nslabs, "even" if args.even_slabs else "balanced by instance count", bounds, args.collective, pipe.K))
                       if multi else "single GPU",
                       "visible": tot_vis, "intersections": tot_I, "processed": tot_Ip, "block_evaluated": st["num_evaluated"],
                       "sort_passes": st["sort_passes"], "depth_ordered_emission": bool(st["depth_ordered"]),
                       "tight_binning": bool(st["tight_binning"]), "blend_mode": "exact" if args.exact else "fused",
                       "camera": "64-step orbit, moved every frame"},
        }
        if multi:
            line["per_rank"] = per_rank
        if verified is not None:
            line["slab_frame_equals_single_gpu_frame"] = verified
            line["every_rank_slab_equals_its_columns_of_the_single_gpu_frame"] = slabs_ok
        stq, stage_source = st, "the timed region"
        if not multi and st["frames_in_flight"] > 1 and not args.no_timing:
            # the timed region kept several frames in flight: the kernels of consecutive frames overlap there and stretch each other, so
            # a stage's hipEvent bracket is not a kernel duration any more.  The same steps are run once more, outside the timed
            # region, strictly one frame after the other: its per-stage times (and the frames/s of that discipline) are what
            # `stages` and `roofline` are computed from; the overlapped brackets are kept as `stages_overlapped_us`.
            line["stages_overlapped_us"] = {k_: round(v, 2) for k_, v in st["stage_us_mean"].items()}
            line["one_frame_in_flight"], stq = flight_pass(r, _abi, uniforms, args, 1)
            r.set_option(_abi.GS_OPT_FRAMES_IN_FLIGHT, st["frames_in_flight"])
            stage_source = "the same steps strictly one frame after the other (one_frame_in_flight), outside the timed region"
        if slab_strict is not None:
            line["one_frame_in_flight"], stq = slab_strict
            line["one_frame_in_flight"]["note"] = "rank 0's slab alone, no exchange"
            stage_source = "rank 0's slab strictly one frame after the other, outside the timed region"
        if not args.no_timing and stq["frames_timed"]:
            st_main, st = st, stq
            ab = algorithmic_bytes(st, W, H, T)
            stages = {}
            for name, us in st["stage_us_mean"].items():
                if name == "ranges" and st.get("tight_binning"):
                    # the tight row pipeline has no ranges kernel (they fall out of the expansion's scan): SURVEY's bytes of the stage
                    # are priced together with "sort", over the time of both brackets
                    stages[name] = {"us": round(us, 2), "alg_bytes": int(ab[name]), "GBps": None, "hbm_frac": None, "priced_with": "sort"}
                    continue
                a_, u_ = ab[name], us
                if name == "sort" and st.get("tight_binning"):
                    a_, u_ = ab["sort"] + ab["ranges"], us + st["stage_us_mean"]["ranges"]
                gbs = a_ / (u_ * 1e-6) / 1e9 if u_ > 0 else 0.0
                stages[name] = {"us": round(us, 2), "alg_bytes": int(a_), "GBps": round(gbs, 1),
                                "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
            dom = max(st["stage_us_mean"], key=lambda k_: st["stage_us_mean"][k_])
            dus = st["stage_us_mean"][dom]
            ach = ab[dom] / (dus * 1e-6) / 1e9
            traffic, tdetail = pmc_traffic(dom, cfg["name"]) if not multi and not args.gaussians else (None, None)
            line["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_detail": tdetail,
                                "alg_bytes_per_launch": int(ab[dom]),
                                "launch_us": round(dus, 2), "frames_timed": st["frames_timed"], "measured_in": stage_source,
                                "rank0_slab_only": multi}
            bus = st["stage_us_mean"]["blend"]
            if bus > 0:
                # 22 flop + 1 exp (counted 2) per pixel x entry actually evaluated (64 pixels per surviving (8x8 block, entry) pair)
                flops = 24.0 * 64.0 * st["num_evaluated"]
                line["roofline"]["blend_valu"] = {"achieved": round(flops / (bus * 1e-6) / 1e12, 2), "peak": VALU_PEAK_TFLOPS,
                                                  "unit": "TFLOP/s", "frac": round(flops / (bus * 1e-6) / 1e12 / VALU_PEAK_TFLOPS, 4)}
            ntx_ = int(np.ceil(np.float32(W) / np.float32(ts)))
            mb = moved_bytes(st, W, H, T, (T + ntx_) < 0xFFFF)
            if mb:
                for name in stages:
                    stages[name]["moved_bytes_this_build"] = int(mb[name])
                    stages[name]["moved_GBps"] = round(mb[name] / (st["stage_us_mean"][name] * 1e-6) / 1e9, 1) if st["stage_us_mean"][name] > 0 else 0.0
            line["stages"] = stages
            line["frame_us_device"] = round(st["frame_us_mean"], 2)
            st = st_main
        line["capacity"] = {"entries": st["capacity"], "max_intersections_seen": st["max_intersections_seen"],
                            "truncated_frames": st["truncated_frames"]}
        if st["truncated_frames"] or trouble:
            line["valid"] = False  # a timed frame was rendered from truncated lists: the number does not count
            line["invalid_reason"] = trouble.get("truncated", "truncated frames")
        if not multi and not args.no_verify:
            line.update(self_check(gsplat, _abi, r, W, H, ts, local_rank, uniforms[(args.warmup + args.steps - 1) % 64], args))
            line["reference_binning"] = reference_binning_leg(gsplat, _abi, r, W, H, ts, local_rank, uniforms, args, eflag)
        if not multi and not args.no_cpu:
            u_last = uniforms[(args.warmup + args.steps - 1) % 64]
            line["cpu_baseline"], ref = cpu_baseline(host_scene, N, W, H, ts, u_last, args.cpu_max)
            if ref is not None and "exact_frame" in _KEEP:
                same = bool(np.array_equal(_KEEP["exact_frame"], ref["rgba8"]))
                line["frame_verified"] = same and line.get("frame_verified", True)
                line["frame_verified_detail"]["exact_frame_equals_cpu_oracle_frame"] = same
            line["copy_probe"] = copy_probe(dev)
            # the reference itself (WGSL on a WebGPU runtime, TypeScript host) cannot run on this box
            line["webgpu_baseline"] = "unavailable (no WebGPU runtime, no TypeScript toolchain, no network)"
        print(json.dumps(line), flush=True)
    if pipe is not None:
        pipe.destroy()
        owner.destroy()
    else:
        r.destroy()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
