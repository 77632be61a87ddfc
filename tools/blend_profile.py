#!/usr/bin/env python3
"""Per-walker timeline of the blend kernel (GS_OPT_BLEND_ABLATION bit 16): lifetimes, concurrency over time, work per walker."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GSPLAT_LIB", os.path.join(ROOT, "gaussian-splatting-wgpu_amd", "lib", "libgsplat_hip_prof.so"))  # the stamps exist in the profiling build only
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import gsplat
from gsplat import _abi, synth
N, W, H = 6_100_000, 1920, 1080
cull = int(sys.argv[1]) if len(sys.argv) > 1 else 1
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sp = synth.bicycle_like_torch(N, synth.BASE_SEED + 1, "cuda")
pg = gsplat.PackedGaussians.__new__(gsplat.PackedGaussians); pg.numGaussians, pg.gaussiansBuffer = N, sp
r = gsplat.Renderer(gsplat.Canvas(W, H), None, 0, pg, 16, flags=_abi.GS_FLAG_TIMING)
r.set_option(_abi.GS_OPT_TILE_CULL, cull)
u = synth.orbit_camera(0, W, H).uniforms(W, H)
for _ in range(3):
    r.render_uniforms(u); r.wait()
r.set_option(_abi.GS_OPT_BLEND_ABLATION, 0x10000 | extra)
r.render_uniforms(u); r.wait()
p = r.read_buffer(11).reshape(-1, 4).astype(np.int64)
st = r.stats()
t0, t1, ev, stg = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
live = stg > 0
t0, t1, ev, stg = t0[live], t1[live], ev[live], stg[live]
base = t0.min()
dur = (t1 - t0) * 0.01  # us (100 MHz)
end = (t1.max() - base) * 0.01
print("cull", cull, "blend stage us", round(st["stage_us"]["blend"], 1), "walkers", live.sum(), "kernel span us", round(end, 1))
print("walker duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
print("evals per walker: mean %.0f p99 %.0f max %d | staged: mean %.0f max %d" % (ev.mean(), np.percentile(ev, 99), ev.max(), stg.mean(), stg.max()))
print("ns per eval (duration/evals), by decile of start time:")
order = np.argsort(t0)
for k in range(10):
    sl = order[k * len(order) // 10:(k + 1) * len(order) // 10]
    print("  decile %d: start %.0f..%.0f us, mean dur %.1f us, mean evals %.0f, us/eval %.3f" % (k, (t0[sl].min() - base) * 0.01, (t0[sl].max() - base) * 0.01, dur[sl].mean(), ev[sl].mean(), dur[sl].sum() / max(ev[sl].sum(), 1)))
# concurrency timeline
edges = np.linspace(0, end, 25)
for a, b in zip(edges[:-1], edges[1:]):
    mid = base + (a + b) * 0.5 / 0.01
    print("  t=%6.0f us: %5d walkers resident" % ((a + b) / 2, int(((t0 <= mid) & (t1 > mid)).sum())))
worst = np.argsort(-dur)[:5]
print("longest walkers: dur", dur[worst].round(1), "evals", ev[worst], "staged", stg[worst], "start", ((t0[worst] - base) * 0.01).round(0))
