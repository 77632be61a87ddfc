"""Fused-vs-canonical blend error report (GPU + oracle): max |rgb - oracle| over well-conditioned pixels, ill fraction.
usage: python tools/fused_error.py   (runs a few seeded scenes incl. a thin-splat stress scene)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd"))
sys.path.insert(0, ROOT)
from gsplat import _abi, Renderer, Canvas, PackedGaussians
from gsplat.synth import bicycle_like, orbit_camera
from oracle import gs_oracle


def run(name, recs, W, H, cam_i=0):
    u = orbit_camera(cam_i, W, H).uniforms(W, H)
    ref = gs_oracle.render(recs, u, W, H, 16, want_illcond=True)
    r = Renderer(Canvas(W, H), None, 0, PackedGaussians(recs), 16, flags=_abi.GS_FLAG_F32_TAP)
    r.render_uniforms(u)
    r.wait()
    f32 = r.read_buffer(_abi.GS_BUF_RGB_F32, np.float32).reshape(H, W, 3)
    ill = ref["illcond"].astype(bool)
    err = np.abs(f32 - ref["rgbf"]).max(axis=2)
    print("%-28s N=%d I=%d  max err (well-conditioned) %.3g  p99.9 %.3g  ill fraction %.4g  max err overall %.3g" % (
        name, len(recs), ref["sorted_keys"].size if "sorted_keys" in ref else -1, err[~ill].max(initial=0), np.quantile(err[~ill], 0.999),
        ill.mean(), err.max()), flush=True)
    r.destroy()


if __name__ == "__main__":
    recs = bicycle_like(200000, seed=3)
    run("bicycle-like 200k @640x360", recs, 640, 360)
    run("bicycle-like 200k @1920x1080", recs, 1920, 1080, 5)
    for name, k in (("thin splats (scale x0.1)", 0.1), ("fat splats (scale x4)", 4.0)):
        sc = bicycle_like(200000, seed=4).copy()
        sc[:, 4:7] += np.float32(np.log(k))  # log-scales (ply.ts: scale_0..2)
        run(name, sc, 640, 360)
