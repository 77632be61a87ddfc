import os, sys, ctypes
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import gsplat
from gsplat import _abi, synth
N, W, H = 6_100_000, 1920, 1080
sp = synth.bicycle_like_torch(N, synth.BASE_SEED + 1, "cuda")
pg = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians); pg.numGaussians, pg.gaussiansBuffer = N, sp
r = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16)
for k in range(6):
    r.render_uniforms(synth.orbit_camera(k, W, H).uniforms(W, H)); r.wait()
L = _abi.load()
buf = np.zeros(8192 * 8, dtype=np.uint64)
L.gs_debug_sweep_stamps.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
print("rc", L.gs_debug_sweep_stamps(buf.ctypes.data, buf.nbytes))
st = buf.reshape(8192, 8).astype(np.int64)
ok = (st[:, 0] > 0) & (st[:, 7] > st[:, 0])
s = st[ok]
print("tiles stamped", ok.sum())
d = np.diff(s, axis=1)
names = ["load+rank", "barrier1", "owner+lookback", "barrier2", "gbase+vals+reorder", "barrier3", "stores"]
tick_ns = 10.0  # s_memtime / readcyclecounter runs at 100 MHz on gfx9
for i, n in enumerate(names):
    print("%-20s median %8.2f us   p90 %8.2f us" % (n, np.median(d[:, i]) * tick_ns / 1000, np.percentile(d[:, i], 90) * tick_ns / 1000))
tot = s[:, 7] - s[:, 0]
print("tile total median %.2f us p90 %.2f us" % (np.median(tot) * tick_ns / 1000, np.percentile(tot, 90) * tick_ns / 1000))
span = (s[:, 7].max() - s[:, 0].min()) * tick_ns / 1000
print("span first start -> last end %.1f us; tile starts per us %.1f" % (span, len(s) / span))
