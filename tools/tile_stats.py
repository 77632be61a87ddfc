#!/usr/bin/env python3
"""Prints the distribution of per-tile list lengths for a bench config (GPU needed)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import gsplat
from gsplat import _abi, synth
N, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 6_100_000, 1920, 1080
sp = synth.bicycle_like_torch(N, synth.BASE_SEED + 1, "cuda")
pg = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians); pg.numGaussians, pg.gaussiansBuffer = N, sp
r = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16, flags=_abi.GS_FLAG_TIMING)
r.render_uniforms(synth.orbit_camera(0, W, H).uniforms(W, H)); r.wait()
rg = r.read_buffer(_abi.GS_BUF_RANGES).astype(np.int64)
ln = np.diff(np.concatenate([[0], rg]))
print("tiles", ln.size, "mean", ln.mean(), "max", ln.max(), "p50", np.percentile(ln, 50), "p90", np.percentile(ln, 90), "p99", np.percentile(ln, 99))
g = ln.reshape(68, 120)
print("row-sums (every 8 rows):", g.sum(1)[::8])
print("top tiles:", np.sort(ln)[-10:])
print(r.stats())
