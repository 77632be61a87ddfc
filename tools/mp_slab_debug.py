#!/usr/bin/env python3
"""Debug: W processes on ONE GPU, each renders its slab (on a torch stream, as bench.py does) and a whole frame, compares locally."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT)
import numpy as np


def worker(rank, world):
    import torch
    import gsplat
    from gsplat import _abi, synth, multigpu
    N, W, H = 6_100_000, 1920, 1080
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    sp = synth.bicycle_like_torch(N, synth.BASE_SEED + 1, dev)
    pg = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians); pg.numGaussians, pg.gaussiansBuffer = N, sp
    b = multigpu.slab_bounds(W, 16, world)
    r = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16, cols=(b[rank], b[rank + 1]), stream=torch.cuda.current_stream(dev).cuda_stream)
    pgv = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians); pgv.numGaussians, pgv.gaussiansBuffer = N, None
    full = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pgv, 16, share_with=r)
    send = torch.zeros(H * (b[rank + 1] - b[rank]) * 16 * 4, dtype=torch.uint8, device=dev)
    bad = 0
    for k in range(12):
        u = synth.orbit_camera(k, W, H).uniforms(W, H)
        r.render_uniforms(u, out_ptr=send.data_ptr())
        if k % 3 == 2:
            r.wait()
            torch.cuda.synchronize(dev)
            full.render_uniforms(u); full.wait()
            whole = full.read_rgba8()
            mine = send.cpu().numpy().reshape(H, -1, 4)
            x0 = b[rank] * 16
            d = (mine != whole[:, x0:x0 + mine.shape[1]]).any(axis=2).sum()
            bad += int(d)
            print("rank", rank, "frame", k, "differing", int(d), "I", r.stats()["num_intersections"], flush=True)
    r.destroy(); full.destroy()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    mp.spawn(worker, args=(world,), nprocs=world, join=True)
