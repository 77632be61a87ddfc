#!/usr/bin/env python3
"""GPU box, profiling build (GSPLAT_LIB=.../libgsplat_hip_prof.so): where the waves of the tight projection spend their cycles."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GSPLAT_LIB", os.path.join(ROOT, "gaussian-splatting-wgpu_amd", "lib", "libgsplat_hip_prof.so"))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import gsplat
from gsplat import _abi, synth
N, W, H = 6_100_000, 1920, 1080
sp = synth.bicycle_like_torch(N, synth.BASE_SEED + 1, "cuda")
pg = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians); pg.numGaussians, pg.gaussiansBuffer = N, sp
r = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16, flags=_abi.GS_FLAG_TIMING)
lib = ctypes.CDLL(os.environ["GSPLAT_LIB"])
u = synth.orbit_camera(0, W, H).uniforms(W, H)
for _ in range(3):
    r.render_uniforms(u); r.wait()
out = (ctypes.c_ulonglong * 8)()
lib.gs_prof_preprocess(out, 1)
K = 10
for k in range(K):
    r.render_uniforms(synth.orbit_camera(k, W, H).uniforms(W, H)); r.wait()
lib.gs_prof_preprocess(out, 0)
st = r.stats()
v = np.array(list(out), dtype=np.float64)
waves = v[5] / K
names = ["cull", "projection arithmetic", "slot scan + cursor bump (2 barriers)", "row-item loop", "colour + record"]
tot = v[:5].sum()
print("preprocess stage us", round(st["stage_us"]["preprocess"], 1), "waves per launch", waves)
for n, c in zip(names, v[:5]):
    print("  %-40s %5.1f %%   %8.0f cycles per wave" % (n, 100 * c / tot, c / v[5]))
print("  total cycles per wave %.0f" % (tot / v[5]))
