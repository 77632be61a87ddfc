#!/usr/bin/env python3
"""Per-rank THROUGHPUT of an N-GPU slab with K frames in flight (K contexts sharing the splats, each on its own stream),
measured on one GPU: what one rank of an N-GPU run can sustain when its frames overlap.  No collective here.
Usage: slab_flight_probe.py [world=8] [K=3] [config=B|C] [emit_order=2] [balanced=1] [ranks=ends|all]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT)
import torch
import gsplat
from gsplat import _abi, synth, multigpu
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfgname = sys.argv[3] if len(sys.argv) > 3 else "B"
order = int(sys.argv[4]) if len(sys.argv) > 4 else 2
balanced = int(sys.argv[5]) if len(sys.argv) > 5 else 1
which = sys.argv[6] if len(sys.argv) > 6 else "ends"
N = 6_100_000
W, H = (1920, 1080) if cfgname == "B" else (3840, 2160)
sp = synth.bicycle_like_torch(N, synth.BASE_SEED + (1 if cfgname == "B" else 2), "cuda")
pg = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians); pg.numGaussians, pg.gaussiansBuffer = N, sp
b = multigpu.slab_bounds(W, 16, world)
us = [synth.orbit_camera(k, W, H).uniforms(W, H) for k in range(64)]
if balanced and world > 1:  # as bench.py does: instances per tile column over 8 cameras of the orbit, from whole-canvas frames
    import numpy as np
    full = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16)
    ntx = multigpu.num_tile_columns(W, 16)
    col = np.zeros(ntx)
    for k in range(0, 64, 8):
        full.render_uniforms(us[k]); full.wait()
        tc = np.diff(np.concatenate([[0], full.read_buffer(_abi.GS_BUF_RANGES).astype(np.int64)])).astype(np.float64)
        col += tc[: (tc.size // ntx) * ntx].reshape(-1, ntx).sum(axis=0)
    b = multigpu.balanced_bounds(col, world)
    full.destroy()
print("bounds", b, flush=True)
for rank in (range(world) if which == "all" else sorted({0, world // 2, world - 1})):
    cols = (b[rank], b[rank + 1]) if world > 1 else None
    streams = [torch.cuda.Stream() for _ in range(K)]
    rs = []
    for k in range(K):
        rs.append(gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16, cols=cols, stream=streams[k].cuda_stream,
                                  share_with=rs[0] if k else None))
        rs[-1].set_option(_abi.GS_OPT_EMIT_ORDER, order)
    for k in range(64):  # capacity calibration
        for r in rs:
            r.render_uniforms(us[k]); r.wait()
    res = {}
    for kk in sorted({1, K}):
        for k in range(10): rs[k % kk].render_uniforms(us[k])
        for r in rs: r.wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 200
        for k in range(steps): rs[k % kk].render_uniforms(us[(10 + k) % 64])
        for r in rs: r.wait()
        torch.cuda.synchronize()
        res[kk] = (time.perf_counter() - t0) / steps * 1e6
    st = rs[0].stats()
    print("world", world, "rank", rank, "cfg", cfgname, "order", order, "depth_ordered", st["depth_ordered"], "I", st["num_intersections"], "vis", st["num_visible"],
          "us/frame:", {("%d in flight" % k): round(v, 1) for k, v in res.items()}, flush=True)
    for r in reversed(rs): r.destroy()
