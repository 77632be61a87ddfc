#!/usr/bin/env python3
"""GPU box, profiling build: how much of an 8x8 block does an evaluated entry touch (lanes with alpha >= 1/255, 4x4 quads holding one)?"""
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
for k in range(3):
    r.render_uniforms(synth.orbit_camera(k, W, H).uniforms(W, H)); r.wait()
r.set_option(_abi.GS_OPT_BLEND_ABLATION, 32)
out = (ctypes.c_ulonglong * 11)()
lib.gs_prof_blend_footprint(out, 1)
for k in range(4):
    r.render_uniforms(synth.orbit_camera(k * 16, W, H).uniforms(W, H)); r.wait()
lib.gs_prof_blend_footprint(out, 0)
ev, lanes, quads, none, live, kept, dead, q1, tailn, tailbits, nokeep = [float(x) for x in out]
print("evaluations %.3g: lanes with alpha >= 1/255: %.1f of 64 (%.1f %%); 4x4 quads touched: %.2f of 4; evaluations touching nothing: %.1f %%" % (
    ev, lanes / ev, 100 * lanes / ev / 64, quads / ev, 100 * none / ev))
print("live lanes (pixel not final) per evaluation: %.1f of 64; kept lanes (both tests): %.1f; evaluations whose alpha >= 1/255 lanes are all final: %.1f %%; evaluations with <= 16 live lanes: %.1f %%" % (
    live / ev, kept / ev, 100 * dead / ev, 100 * q1 / ev))
print("parked entries while <= 16 pixels are live: %.3g (%.1f %% of the evaluations); blocks of the tile their mask names: %.2f of 4" % (tailn, 100 * tailn / ev, tailbits / max(tailn, 1)))
print("evaluations that keep no lane (alpha or transmittance test fails on every live pixel): %.1f %%" % (100 * nokeep / ev))
