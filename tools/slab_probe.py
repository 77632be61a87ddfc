#!/usr/bin/env python3
"""Per-rank cost of an N-GPU slab (what one rank of an 8-GPU run does), measured on one GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import gsplat
from gsplat import _abi, synth, multigpu
N, W, H = 6_100_000, 1920, 1080
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
order = int(sys.argv[2]) if len(sys.argv) > 2 else 2  # GS_OPT_EMIT_ORDER
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # GS_OPT_PROJ_CHUNKS
sp = synth.bicycle_like_torch(N, synth.BASE_SEED + 1, "cuda")
pg = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians); pg.numGaussians, pg.gaussiansBuffer = N, sp
b = multigpu.slab_bounds(W, 16, world)
us = [synth.orbit_camera(k, W, H).uniforms(W, H) for k in range(64)]
for rank in sorted({0, world // 2, world - 1}):
    r = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16, flags=_abi.GS_FLAG_TIMING, cols=(b[rank], b[rank + 1]))
    r.set_option(_abi.GS_OPT_EMIT_ORDER, order)
    if chunks: r.set_option(_abi.GS_OPT_PROJ_CHUNKS, chunks)
    for k in range(10): r.render_uniforms(us[k])
    r.wait(); r.set_option(_abi.GS_OPT_RESET_TIMING, 0)
    for k in range(40): r.render_uniforms(us[10 + k])
    r.wait(); st = r.stats()
    print("order", order, "depth_ordered", st["depth_ordered"], "rank", rank, "of", world, "I", st["num_intersections"], "vis", st["num_visible"], "frame_us", round(st["frame_us_mean"], 1),
          {k: round(v, 1) for k, v in st["stage_us_mean"].items()})
    r.destroy()
