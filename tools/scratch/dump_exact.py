#!/usr/bin/env python3
"""scratch: EXACT product frame (f32 tap) of a scaled config-B scene with the library in $GSPLAT_LIB -> npy"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gsplat
from gsplat import _abi, synth
import gpu_checks as gc
n, W, H, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
s = synth.bicycle_like(n, synth.BASE_SEED + 1)
u = synth.orbit_camera(0, W, H).uniforms(W, H)
r = gc.make_renderer(s, W, H, 16, flags=_abi.GS_FLAG_EXACT_BLEND)
r.render_uniforms(u); r.wait()
f32 = r.read_buffer(_abi.GS_BUF_RGB_F32, np.float32).reshape(H, W, 3)
np.save(out, f32)
print(out, r.stats()["num_evaluated"], r.stats()["num_intersections"])
