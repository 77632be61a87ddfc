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
    n = int(rng.choice([1, 7, 64, 300, 3000, 20000, 60000, 250000]))
    if os.environ.get("FUZZ_BIG"):  # a few large cases: canvases up to 4K, up to 3 M splats
        n = int(rng.choice([500_000, 1_200_000, 3_000_000])); W = int(rng.integers(1200, 3841)); H = int(rng.integers(700, 2161))
    step = int(rng.integers(0, 64))
    mod = float(rng.choice([0.3, 1.0, 1.0, 1.0, 2.0, 4.0]))
    s = synth.bicycle_like(n, synth.BASE_SEED + 100 + k)
    if rng.random() < 0.3:  # some opaque, some nearly transparent splats
        idx = rng.integers(0, n, max(n // 10, 1)); s[idx, 12] = rng.choice([8.0, -6.0, -5.5], idx.size).astype(np.float32)
    mut = rng.random()
    if mut < 0.12:    # everything far away: depth bucket 999 for all -- the order inside a tile is the gaussian index alone
        s[:, 0:3] *= np.float32(rng.choice([6.0, 15.0]))
    elif mut < 0.2:   # non-finite and extreme values here and there
        # (not: non-finite SH coefficients or opacity logits.  WGSL lets an implementation assume NaNs and infinities away; the oracle's
        # IEEE reading makes `0 * inf` poison every pixel of every tile of such a splat's rect, which no culling renderer reproduces)
        for col, val in ((0, np.nan), (2, np.inf), (4, 30.0), (8, 0.0), (12, 1e30), (12, -1e30), (5, -40.0), (9, np.inf)):
            idx = rng.integers(0, n, max(n // 200, 1)); s[idx, col] = np.float32(val)
        idx = rng.integers(0, n, max(n // 200, 1)); s[idx, 8:12] = 0.0  # zero quaternions: 0 / 0
    elif mut < 0.3:   # a share of very large splats (rects of hundreds of tiles, long row runs)
        idx = rng.integers(0, n, max(n // 50, 1)); s[idx, 4:7] += np.float32(rng.choice([2.0, 3.5]))
    u = synth.orbit_camera(step, W, H).uniforms(W, H).copy()
    u[39] = np.float32(mod)
    tag = "case %d: n %d %dx%d ts %d step %d mod %g" % (k, n, W, H, ts, step, mod)
    try:
        ref = o.render(s, u, W, H, ts)
        if ref["num_intersections"] > 60_000_000: print(tag, "skipped (too many instances)"); continue
        exact = rng.random() < 0.6 or 0.12 <= mut < 0.2  # (non-finite records: the fused arithmetic cannot follow the oracle's NaNs)
        cull = 0 if rng.random() < 0.15 else 1
        r = gc.make_renderer(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND if exact else 0)
        r.set_option(_abi.GS_OPT_TILE_CULL, cull)
        if not cull: r.set_option(_abi.GS_OPT_EMIT_ORDER, int(rng.integers(0, 3)))
        if rng.random() < 0.3: r.set_option(_abi.GS_OPT_PROJ_CHUNKS, int(rng.choice([2, 4, 8])))
        if rng.random() < 0.4:  # frames in flight: two other cameras first, no wait in between
            for st2 in (int(rng.integers(0, 64)), int(rng.integers(0, 64))):
                u2 = synth.orbit_camera(st2, W, H).uniforms(W, H).copy(); u2[39] = np.float32(mod)
                r.render_uniforms(u2)
        r.render_uniforms(u)
        try:
            r.wait()
        except _abi.GsError as e:
            if e.code != -9: raise  # an earlier frame of the batch outgrew a capacity: reported, the last frame is complete
        st = r.stats()
        if st["tight_binning"] and exact:
            gc.check_product_lists(r, ref, o, W, H, ts)
        if exact and 0.12 <= mut < 0.2:  # NaN pixels: equal where both are NaN (sign and payload of a NaN are not defined), bit-equal elsewhere
            f32 = r.read_buffer(_abi.GS_BUF_RGB_F32, np.float32).reshape(H, W, 3)
            na, nb = np.isnan(f32), np.isnan(ref["rgbf"])
            assert (na == nb).all(), "%d values are NaN on one side only" % int((na != nb).sum())
            assert (f32.view(np.uint32)[~na] == ref["rgbf"].view(np.uint32)[~na]).all()
        elif exact:
            gc.check_image(r, ref, True)
        else:
            refi = o.render(s, u, W, H, ts, want_illcond=True)
            gc.check_image(r, refi, False, max_ill=0.6)
        r.destroy()
        ntx = -(-W // ts)
        if ntx >= 2:
            c0 = int(rng.integers(0, ntx - 1)); c1 = int(rng.integers(c0 + 1, ntx + 1))
            r = gc.make_renderer(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND, cols=(c0, c1))
            r.render_uniforms(u); r.wait()
            if 0.12 <= mut < 0.2:
                f32 = r.read_buffer(_abi.GS_BUF_RGB_F32, np.float32).reshape(H, -1, 3)
                want = ref["rgbf"][:, r.slab_x0:r.slab_x0 + r.slab_width]
                na, nb = np.isnan(f32), np.isnan(want)
                assert (na == nb).all() and (f32.view(np.uint32)[~na] == want.view(np.uint32)[~na]).all()
            else:
                gc.check_image(r, ref, True)
            r.destroy()
        print(tag, "ok  I", ref["num_intersections"], "tight", st["tight_binning"], flush=True)
    except Exception as e:
        bad += 1
        print(tag, "FAILED", repr(e)[:300], flush=True)
        traceback.print_exc(limit=2)
print("%d cases, %d failed, %.0f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
