#!/usr/bin/env python3
"""Throughput with K frames in flight: K contexts (own streams, own per-frame buffers) rendered round-robin."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import gsplat
from gsplat import _abi, synth
N, W, H = 6_100_000, 1920, 1080
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = 200
sp = synth.bicycle_like_torch(N, synth.BASE_SEED + 1, "cuda")
pg = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians); pg.numGaussians, pg.gaussiansBuffer = N, sp
rs = [gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16, flags=0) for _ in range(K)]
us = [synth.orbit_camera(k, W, H).uniforms(W, H) for k in range(64)]
for k in range(20):
    rs[k % K].render_uniforms(us[k % 64])
for r in rs: r.wait()
t0 = time.perf_counter()
for k in range(steps):
    r = rs[k % K]
    if k >= K: r.wait()          # the frame this context rendered K steps ago must be complete before its buffers are reused
    r.render_uniforms(us[(20 + k) % 64])
for r in rs: r.wait()
dt = time.perf_counter() - t0
print("frames in flight", K, "fps", round(steps / dt, 1), "ms/frame", round(dt / steps * 1e3, 3))
for r in rs: r.destroy()
