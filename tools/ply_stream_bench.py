#!/usr/bin/env python3
"""GPU box: streaming .ply upload (gs_upload_ply) of an n-gaussian synthetic scene: seconds, peak host RSS above the baseline,
and -- beside it -- the gs_ply_load + gs_upload_splats route through N x 320-byte records.  Usage: tools/ply_stream_bench.py [n]"""
import ctypes, os, resource, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import gsplat
from gsplat import _abi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_100_000
d = tempfile.mkdtemp(dir=os.environ.get("GS_TMP", None))
path = os.path.join(d, "scene.ply")
names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + ["f_rest_%d" % i for i in range(45)] + \
        ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
with open(path, "wb") as f:  # written chunk by chunk: the generator itself must not hold the scene
    f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join("property float %s\n" % p for p in names) + "end_header\n").encode())
    for c0 in range(0, n, synth.CHUNK):
        m = min(synth.CHUNK, n - c0)
        full = synth.bicycle_like(synth.CHUNK, synth.BASE_SEED + 1 + c0 // synth.CHUNK)[:m]  # any data will do: chunk-seeded
        cols = {"x": full[:, 0], "y": full[:, 1], "z": full[:, 2], "nx": 0 * full[:, 0], "ny": 0 * full[:, 0], "nz": 0 * full[:, 0],
                "opacity": full[:, 12], "scale_0": full[:, 4], "scale_1": full[:, 5], "scale_2": full[:, 6],
                "rot_0": full[:, 8], "rot_1": full[:, 9], "rot_2": full[:, 10], "rot_3": full[:, 11]}
        for c in range(3):
            cols["f_dc_%d" % c] = full[:, 16 + c]
            for i in range(15):
                cols["f_rest_%d" % (c * 15 + i)] = full[:, 16 + 4 * (i + 1) + c]
        f.write(np.stack([cols[p] for p in names], axis=1).astype("<f4").tobytes())
        del full, cols
size = os.path.getsize(path)
L = _abi.load()
cfg = _abi.GsConfig(); cfg.struct_size = ctypes.sizeof(cfg); cfg.width, cfg.height, cfg.tile_size, cfg.device = 1920, 1080, 16, 0
ctx = ctypes.c_void_p(); _abi.check(L.gs_create(ctypes.byref(cfg), ctypes.byref(ctx)))
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
cnt = ctypes.c_uint64()
t0 = time.perf_counter(); _abi.check(L.gs_upload_ply(ctx, path.encode(), ctypes.byref(cnt))); t1 = time.perf_counter()
rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("gs_upload_ply (streaming): %d gaussians, file %.0f MB, %.2f s = %.2f M gaussians/s (%.2f GB/s of file); peak host RSS grew by %.0f MB (file + 256 MB = %.0f MB)"
      % (cnt.value, size / 1e6, t1 - t0, n / (t1 - t0) / 1e6, size / (t1 - t0) / 1e9, (rss1 - rss0) / 1024.0, size / 1e6 + 256))
if n <= 10_000_000:
    r, nn, dd = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_int32()
    t0 = time.perf_counter(); _abi.check(L.gs_ply_load(path.encode(), ctypes.byref(r), ctypes.byref(nn), ctypes.byref(dd)))
    _abi.check(L.gs_upload_splats(ctx, r, nn)); t1 = time.perf_counter()
    L.gs_ply_free(r)
    rss2 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    print("gs_ply_load + gs_upload_splats (N x 320-byte records on the host): %.2f s; peak host RSS grew by %.0f MB more" % (t1 - t0, (rss2 - rss1) / 1024.0))
_abi.check(L.gs_destroy(ctx))
os.remove(path)
