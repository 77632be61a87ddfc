#!/usr/bin/env python3
"""Writes a synthetic 3DGS .ply (bicycle-like scene) for exercising `bench.py --ply` / $GS_PLY and the native loader.
usage: python tools/make_ply.py out.ply [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gsplat import synth
from ply_bench import write_ply
if __name__ == "__main__":
    write_ply(sys.argv[1], synth.bicycle_like(int(sys.argv[2]) if len(sys.argv) > 2 else 300000))
    print("wrote", sys.argv[1])
