#!/usr/bin/env python3
"""GPU box: randomized PLY files (property order, float / uchar types, unused properties, SH degree 0..3, vertex counts around the
64 Ki-vertex chunks of the streaming loader) -- the streamed device scene (gs_upload_ply) must equal the scene uploaded from the
packed records of gs_ply_load, bit for bit (tap 12).  Usage: tools/fuzz_ply.py [cases=60] [seed0=0]"""
import ctypes, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.zeros(1, device="cuda")
import gsplat
from gsplat import _abi, synth
u = synth.orbit_camera(3, 64, 64).uniforms(64, 64)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d = tempfile.mkdtemp()
bad = 0
for k in range(seed0, seed0 + cases):
    rng = np.random.default_rng(77 + k)
    n = int(rng.choice([1, 2, 63, 4097, 65535, 65536, 65537, 131071, 131072, 131073, 200001]))
    deg = int(rng.integers(0, 4))
    names = ["x", "y", "z", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3", "opacity", "f_dc_0", "f_dc_1", "f_dc_2"] + \
            ["f_rest_%d" % i for i in range(3 * ((deg + 1) ** 2 - 1))]
    extra = ["nx", "ny", "nz", "junk_a", "junk_b"][: int(rng.integers(0, 6))]
    allp = names + extra
    rng.shuffle(allp)
    types = {p: ("u1" if rng.random() < 0.15 else "<f4") for p in allp}
    dt = np.dtype([(p, types[p]) for p in allp], align=False)
    arr = np.zeros(n, dtype=dt)
    for p in allp:
        arr[p] = rng.integers(0, 256, n).astype(np.uint8) if types[p] == "u1" else rng.standard_normal(n).astype(np.float32)
    path = os.path.join(d, "f%d.ply" % k)
    with open(path, "wb") as f:
        hdr = "ply\nformat binary_little_endian 1.0\n" + ("comment fuzz %d\n" % k if rng.random() < 0.5 else "") + "element vertex %d\n" % n
        hdr += "".join("property %s %s\n" % ("uchar" if types[p] == "u1" else "float", p) for p in allp) + "end_header\n"
        f.write(hdr.encode()); f.write(arr.tobytes())
    tag = "case %d: n %d degree %d, %d properties (%d uchar), stride %d" % (k, n, deg, len(allp), sum(t == "u1" for t in types.values()), dt.itemsize)
    try:
        pg = gsplat.PackedGaussians.from_ply(path)
        assert pg.numGaussians == n and pg.sphericalHarmonicsDegree == deg
        r = gsplat.Renderer(gsplat.Canvas(64, 64), None, 0, pg, 16)
        r.render_uniforms(u); r.wait()  # (taps need a frame)
        want = r.read_buffer(12).copy()
        cnt = ctypes.c_uint64()
        _abi.check(_abi.load().gs_upload_ply(r._ctx, path.encode(), ctypes.byref(cnt)))
        assert cnt.value == n
        r.render_uniforms(u); r.wait()
        got = r.read_buffer(12)
        assert got.shape == want.shape
        npad = (n + 63) & ~63  # scene tap: planes px, py, pz, smax [npad each] | geo [npad][8] | sh [n][48]; the pad entries are never written
        valid = np.zeros(got.size, dtype=bool)
        for pl in range(4): valid[pl * npad: pl * npad + n] = True
        valid[4 * npad: 4 * npad + 8 * n] = True
        valid[4 * npad + 8 * npad:] = True
        assert (got[valid] == want[valid]).all(), "%d words differ" % int((got[valid] != want[valid]).sum())
        r.destroy()
        print(tag, "ok", flush=True)
    except Exception as e:
        bad += 1
        print(tag, "FAILED", repr(e)[:300], flush=True)
    os.remove(path)
print("%d cases, %d failed" % (cases, bad))
sys.exit(1 if bad else 0)
