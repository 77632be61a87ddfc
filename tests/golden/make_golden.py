#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/gs_oracle.c).

The reference (WGSL + TypeScript in a browser) cannot be executed in this environment and ships no
fixtures of its own, so these vectors pin the ORACLE, not the reference: they guard
against silent drift of the canonical semantics and let the GPU box check itself without
/root/reference.  Inputs are regenerated from the seed (gsplat.synth, numpy Philox), so only the
expected outputs are stored.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd"))
sys.path.insert(0, ROOT)
from gsplat import synth  # noqa: E402
from oracle import gs_oracle  # noqa: E402

CASES = {
    # name: (n, W, H, tile, orbit step)   -- cfg-A of BASELINE.json and its 256-multiple twin
    "cfgA_10000_256": (10000, 256, 256, 16, 3),
    "cfgA_10240_256": (10240, 256, 256, 16, 3),
    "ragged_3001_200x120_t8": (3001, 200, 120, 8, 9),
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for name, (n, W, H, ts, step) in CASES.items():
        s = synth.bicycle_like(n)
        u = synth.orbit_camera(step, W, H).uniforms(W, H)
        r = gs_oracle.render(s, u, W, H, ts)
        np.savez_compressed(
            os.path.join(here, name + ".npz"),
            params=np.array([n, W, H, ts, step], dtype=np.int64),
            uniforms=u,
            input_sha256=np.array(sha(s)),
            tile_counts=r["tile_counts"],
            num_intersections=np.array(r["num_intersections"], dtype=np.int64),
            gdata_sha256=np.array(sha(r["gdata"])),
            sorted_keys=r["sorted_keys"],
            sorted_values=r["sorted_values"],
            ranges=r["ranges"],
            rgba8=r["rgba8"],
            rgbf_sha256=np.array(sha(r["rgbf"])),
        )
        print(name, "I =", r["num_intersections"], "bytes =", os.path.getsize(os.path.join(here, name + ".npz")))


if __name__ == "__main__":
    main()
