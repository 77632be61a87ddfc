#!/usr/bin/env python3
"""CPU-only logic check of the tight binning (gs_tight.h) against the oracle's exact contribution masks.
Usage: python tools/tight_check/run.py [n] [W] [H] [ts] [camera step] [scale_modifier]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT)
import numpy as np
from gsplat import synth
from oracle import gs_oracle as o
n, W, H, ts, step = (int(x) for x in (sys.argv[1:6] + ["200000", "1920", "1080", "16", "0"][len(sys.argv) - 1:]))
mod = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
d = tempfile.mkdtemp()
src = open(os.path.join(ROOT, "gaussian-splatting-wgpu_amd", "csrc", "gs_tight.h")).read().replace('#include "gs_device.h"', "")
open(os.path.join(d, "gs_tight_host.h"), "w").write(src)
exe = os.path.join(d, "check")
subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", d, os.path.join(ROOT, "tools", "tight_check", "host_check.cpp"), "-o", exe])
s = synth.bicycle_like(n, synth.BASE_SEED + 1)
u = synth.orbit_camera(step, W, H).uniforms(W, H).copy()
u[39] = np.float32(mod)
gd, cnt = o.preprocess(s, u, W, H, ts)
off, I = o.scan(cnt)
k, v = o.emit(gd, off, cnt, I, W, ts)
sk, sv = o.sort(k, v)
m = o.instance_masks(gd, sk, sv, W, H, ts)
for name, a in (("gd", gd), ("sk", sk), ("sv", sv), ("m", m)):
    a.astype(np.uint32).tofile(os.path.join(d, name + ".bin"))
sys.exit(subprocess.call([exe] + [os.path.join(d, x + ".bin") for x in ("gd", "sk", "sv", "m")] + [str(W), str(H), str(ts)]))
