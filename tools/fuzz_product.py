#!/usr/bin/env python3
"""GPU box: randomized product-path frames against the oracle (whole canvas: lists proven a harmless subset + EXACT image bit-equal;
a random tile-column slab of the same frame: EXACT image bit-equal on its columns).  A one-off hunt for rare cases, beyond the fixed
seeds of tests/.  Usage: tools/fuzz_product.py [cases=120] [seed0=0]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.zeros(1, device="cuda")
from gsplat import _abi, synth
from oracle import gs_oracle as o
import gpu_checks as gc

o.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for k in range(seed0, seed0 + cases):
    rng = np.random.default_rng(1000 + k)
    ts = int(rng.choice([8, 16, 16, 16, 32]))
    W = int(rng.integers(17, 1400)); H = int(rng.integers(17, 900))
    if rng.random() < 0.15: W = ts * int(rng.integers(1, 60)); H = ts * int(rng.integers(1, 40))  # exact multiples of the tile
    n = int(rng.choice([1, 7, 64, 300, 3000, 20000, 60000]))
    step = int(rng.integers(0, 64))
    mod = float(rng.choice([0.3, 1.0, 1.0, 1.0, 2.0, 4.0]))
    s = synth.bicycle_like(n, synth.BASE_SEED + 100 + k)
    if rng.random() < 0.3:  # some opaque, some nearly transparent splats
        idx = rng.integers(0, n, max(n // 10, 1)); s[idx, 12] = rng.choice([8.0, -6.0, -5.5], idx.size).astype(np.float32)
    u = synth.orbit_camera(step, W, H).uniforms(W, H).copy()
    u[39] = np.float32(mod)
    tag = "case %d: n %d %dx%d ts %d step %d mod %g" % (k, n, W, H, ts, step, mod)
    try:
        ref = o.render(s, u, W, H, ts)
        if ref["num_intersections"] > 60_000_000: print(tag, "skipped (too many instances)"); continue
        r = gc.make_renderer(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND)
        r.render_uniforms(u); r.wait()
        st = r.stats()
        if st["tight_binning"]:
            gc.check_product_lists(r, ref, o, W, H, ts)
        gc.check_image(r, ref, True)
        r.destroy()
        ntx = -(-W // ts)
        if ntx >= 2:
            c0 = int(rng.integers(0, ntx - 1)); c1 = int(rng.integers(c0 + 1, ntx + 1))
            r = gc.make_renderer(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND, cols=(c0, c1))
            r.render_uniforms(u); r.wait()
            gc.check_image(r, ref, True)
            r.destroy()
        print(tag, "ok  I", ref["num_intersections"], "tight", st["tight_binning"], flush=True)
    except Exception as e:
        bad += 1
        print(tag, "FAILED", repr(e)[:300], flush=True)
        traceback.print_exc(limit=2)
print("%d cases, %d failed, %.0f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
