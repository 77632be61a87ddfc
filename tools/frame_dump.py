#!/usr/bin/env python3
"""GPU box: product frames (f32 tap) of a scaled config-B scene with the library in $GSPLAT_LIB -> npy.  args: n W H out [ablation] [fused]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gsplat
from gsplat import _abi, synth
import gpu_checks as gc
n, W, H, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
abl = int(sys.argv[5]) if len(sys.argv) > 5 else 0
fused = len(sys.argv) > 6
s = synth.bicycle_like(n, synth.BASE_SEED + 1)
r = gc.make_renderer(s, W, H, 16, flags=0 if fused else _abi.GS_FLAG_EXACT_BLEND)
if abl: r.set_option(_abi.GS_OPT_BLEND_ABLATION, abl)
imgs = []
for step in (0, 21, 40):
    u = synth.orbit_camera(step, W, H).uniforms(W, H)
    r.render_uniforms(u); r.wait()
    imgs.append(r.read_buffer(_abi.GS_BUF_RGB_F32, np.float32).reshape(H, W, 3).copy())
np.save(out, np.stack(imgs))
st = r.stats()
print(out, "evaluated", st["num_evaluated"], "I", st["num_intersections"], "processed", st["num_processed"])
