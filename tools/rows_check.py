#!/usr/bin/env python3
"""GPU box: the tight row pipeline (k_rows.hip) against the oracle on a few frames, with readable diagnostics.
Usage: python tools/rows_check.py [n] [W] [H] [ts] [step]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import gsplat
from gsplat import _abi, synth
from oracle import gs_oracle as o
import gpu_checks

o.build()
cases = [(10240, 256, 256, 16, 3), (60000, 640, 360, 16, 11), (30000, 640, 360, 8, 5), (40000, 1024, 768, 32, 7), (200000, 1920, 1080, 16, 0)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in sys.argv[1:6])]
for (n, W, H, ts, step) in cases:
    s = synth.bicycle_like(n)
    u = synth.orbit_camera(step, W, H).uniforms(W, H)
    ref = o.render(s, u, W, H, ts)
    r = gpu_checks.make_renderer(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND)
    r.render_uniforms(u)
    r.wait()
    st = r.stats()
    print("case", (n, W, H, ts, step), {k: st[k] for k in ("num_visible", "num_intersections", "num_row_items", "num_row_slots", "row_capacity", "capacity", "tight_binning")},
          "ref I", ref["num_intersections"], flush=True)
    try:
        gpu_checks.check_stages(r, ref, True, debug=False, oracle=o, W=W, H=H, ts=ts)
        print("  OK", flush=True)
    except AssertionError as e:
        print("  FAIL", str(e)[:600], flush=True)
        keys = r.read_buffer(_abi.GS_BUF_KEYS); vals = r.read_buffer(_abi.GS_BUF_VALUES); rng = r.read_buffer(_abi.GS_BUF_RANGES)
        print("  keys", keys[:12], "vals", vals[:12] & 0x0FFFFFFF, "ranges", rng[:12], "sorted", bool((np.diff(keys.astype(np.int64)) >= 0).all()))
        img = r.read_rgba8()
        print("  image differs in", int((img != ref["rgba8"]).any(axis=2).sum()), "pixels of", img.shape[0] * img.shape[1])
    r.destroy()
