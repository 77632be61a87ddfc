#!/usr/bin/env python3
"""GPU box: random SEQUENCES of calls on one context -- new scenes of other sizes, option changes (binning, emission order, frames in
flight, frame graph, projection chunks, workgroup-per-tile blend), bursts of frames without a wait (a debug frame among them), frames presented through
gs_render_host with tickets waited for in any order (every sink checked), reads in between -- the last frame
of every burst against the oracle (EXACT, bit for bit).  Hunts life-cycle bugs (stale captures, ring members with old arrays,
capacities).  Usage: tools/fuzz_sequence.py [sequences=30] [seed0=0]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import gsplat
from gsplat import _abi, synth
from oracle import gs_oracle as o
import gpu_checks as gc

o.build()
seqs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
L = _abi.load()
for q in range(seed0, seed0 + seqs):
    rng = np.random.default_rng(31337 + q)
    ts = int(rng.choice([8, 16, 16, 32]))
    W = int(rng.integers(64, 900)); H = int(rng.integers(64, 600))
    log = []
    try:
        n = int(rng.choice([200, 5000, 40000]))
        s = synth.bicycle_like(n, synth.BASE_SEED + 7 * q)
        r = gc.make_renderer(s, W, H, ts, flags=_abi.GS_FLAG_EXACT_BLEND)
        for op in range(int(rng.integers(6, 14))):
            c = rng.random()
            if c < 0.15:
                n = int(rng.choice([1, 300, 8000, 90000]))
                s = synth.bicycle_like(n, synth.BASE_SEED + 7 * q + op + 1)
                r.wait()
                _abi.check(L.gs_upload_splats(r._ctx, np.ascontiguousarray(s).ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n)))
                log.append("upload %d" % n)
            elif c < 0.45:
                key, val = [(_abi.GS_OPT_TILE_CULL, int(rng.integers(0, 2))), (_abi.GS_OPT_EMIT_ORDER, int(rng.integers(0, 3))),
                            (_abi.GS_OPT_FRAMES_IN_FLIGHT, int(rng.integers(1, 5))), (_abi.GS_OPT_FRAME_GRAPH, int(rng.integers(0, 2))),
                            (_abi.GS_OPT_PROJ_CHUNKS, int(rng.choice([0, 2, 4, 8]))), (_abi.GS_OPT_BLEND_ABLATION, int(rng.choice([0, 8])))][int(rng.integers(0, 6))]
                r.wait()
                r.set_option(key, val)
                log.append("opt %d=%d" % (key, val))
            elif c < 0.60:
                # pipelined presentation: k frames through gs_render_host into sinks of their own (plain and debug frames of other
                # cameras enqueued in between), tickets waited for in random order, EVERY sink against the oracle
                k = int(rng.integers(1, 6))
                us = [synth.orbit_camera(int(rng.integers(0, 64)), W, H).uniforms(W, H).copy() for _ in range(k)]
                others = [synth.orbit_camera(int(rng.integers(0, 64)), W, H).uniforms(W, H).copy() for _ in range(k)]
                for u in us + others:  # capacities first (a ticket's frame is not re-rendered): every member of the ring sees every camera
                    for _ in range(4): r.render_uniforms(u)
                    r.wait()
                nbytes = W * H * 4
                sinks, tickets = [], []
                for j, u in enumerate(us):
                    pp = ctypes.c_void_p(); _abi.check(L.gs_host_alloc(nbytes, ctypes.byref(pp))); sinks.append(pp)
                    t = ctypes.c_uint64()
                    uu = np.ascontiguousarray(u, dtype=np.float32)
                    _abi.check(L.gs_render_host(r._ctx, uu.ctypes.data, pp, nbytes, ctypes.byref(t)))
                    tickets.append(t.value)
                    x = rng.random()
                    if x < 0.25: r.render_uniforms(others[j])
                    elif x < 0.35: r.render_uniforms(others[j], debug=True)
                log.append("host %d" % k)
                for j in rng.permutation(k):
                    _abi.check(L.gs_wait_ticket(r._ctx, tickets[int(j)]))
                    got = np.ctypeslib.as_array(ctypes.cast(sinks[int(j)], ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,)).reshape(H, W, 4).copy()
                    ref = o.render(s, us[int(j)], W, H, ts)
                    np.testing.assert_array_equal(got, ref["rgba8"])
                r.wait()
                for pp in sinks: L.gs_host_free(pp)
            else:
                k = int(rng.integers(1, 5))
                mod = float(rng.choice([0.5, 1.0, 1.0, 2.5]))
                us = []
                for _ in range(k):
                    u = synth.orbit_camera(int(rng.integers(0, 64)), W, H).uniforms(W, H).copy(); u[39] = np.float32(mod); us.append(u)
                # gs_render_debug (reference binning, index order, every tap) as one frame of the burst, anywhere in it
                dbg_at = int(rng.integers(0, k)) if rng.random() < 0.2 else -1
                for j, u in enumerate(us): r.render_uniforms(u, debug=(j == dbg_at))
                dbg = dbg_at == k - 1
                try:
                    r.wait()
                except _abi.GsError as e:
                    if e.code != -9: raise
                    log.append("(truncated)")
                log.append("burst %d" % k)
                ref = o.render(s, us[-1], W, H, ts)
                gc.check_image(r, ref, True)
                if dbg: gc.check_stages(r, ref, exact_image=True)
                if rng.random() < 0.3 and not dbg:
                    st = r.stats(); assert st["num_gaussians"] == n
                    if st["tight_binning"]: gc.check_product_lists(r, ref, o, W, H, ts)
        r.destroy()
        print("sequence %d (%dx%d ts %d): ok  %s" % (q, W, H, ts, " | ".join(log)), flush=True)
    except Exception as e:
        bad += 1
        print("sequence %d (%dx%d ts %d): FAILED %s  after: %s" % (q, W, H, ts, repr(e)[:300], " | ".join(log)), flush=True)
        try: r.destroy()
        except Exception: pass
print("%d sequences, %d failed" % (seqs, bad))
sys.exit(1 if bad else 0)
