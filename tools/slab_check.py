#!/usr/bin/env python3
"""One process, one GPU: renders every slab of an N-way split into send buffers, assembles them on the device and compares
with the whole-canvas frame.  usage: python tools/slab_check.py <gaussians> <world> <emit order 0|1|2>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import gsplat
from gsplat import _abi, synth, multigpu
N, W, H, world = int(sys.argv[1]), 1920, 1080, int(sys.argv[2])
order = int(sys.argv[3])
sp = synth.bicycle_like_torch(N, synth.BASE_SEED + 1, "cuda")
pg = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians); pg.numGaussians, pg.gaussiansBuffer = N, sp
u = synth.orbit_camera(24, W, H).uniforms(W, H)
for flags in (0,):
    full = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16, flags=flags)
    full.render_uniforms(u); full.wait(); whole = full.read_rgba8()
    b = multigpu.slab_bounds(W, 16, world)
    xs = [multigpu.SlabExchange(W, H, 16, world, g, torch.device("cuda")) for g in range(world)]
    rs = []
    for g in range(world):
        r = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16, flags=flags, cols=(b[g], b[g + 1]))
        r.set_option(_abi.GS_OPT_EMIT_ORDER, order)
        for rep in range(3):
            r.render_uniforms(u, out_ptr=xs[g].send.data_ptr()); r.wait()
        rs.append(r)
    x0 = xs[0]; x0.renderer = rs[0]
    for g in range(world):
        x0.gathered[g * x0.stride:(g + 1) * x0.stride].copy_(xs[g].send)
    img = x0.assemble().cpu().numpy()
    torch.cuda.synchronize()
    d = (img != whole).any(axis=2)
    print("N", N, "world", world, "order", order, [r.stats()["depth_ordered"] for r in rs], [r.stats()["num_intersections"] for r in rs], "flags", flags, "differing pixels", int(d.sum()), "columns", np.unique(np.nonzero(d)[1])[:20] if d.any() else [])
    # per-slab direct read
    parts = [r.read_rgba8() for r in rs]
    cat = np.concatenate(parts, axis=1)
    print("  direct slab reads equal whole:", np.array_equal(cat, whole), "assembled equals concat:", np.array_equal(img, cat))
    for r in rs: r.destroy()
    full.destroy()
