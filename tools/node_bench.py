#!/usr/bin/env python3
"""GPU box: frames/s of the Node host's Renderer.animate() at config B (tests/js/host_check.js bench): the reference-shaped loop
(await every frame) without and with a frame sink, and the pipelined mode (canvas.pipeline = 3, pinned sinks).
Usage: tools/node_bench.py [n] [frames]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from gsplat import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_100_000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
W, H, ts = 1920, 1080, 16
d = tempfile.mkdtemp()
path = os.path.join(d, "scene.ply")
names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + ["f_rest_%d" % i for i in range(45)] + \
        ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
scene = synth.bicycle_like(n, synth.BASE_SEED + 1)  # config B's distribution (numpy generator; 1.95 GB at 6.1 M)
with open(path, "wb") as f:
    f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join("property float %s\n" % p for p in names) + "end_header\n").encode())
    step = 1 << 20
    for c0 in range(0, n, step):
        full = scene[c0:c0 + step]
        cols = {"x": full[:, 0], "y": full[:, 1], "z": full[:, 2], "nx": 0 * full[:, 0], "ny": 0 * full[:, 0], "nz": 0 * full[:, 0],
                "opacity": full[:, 12], "scale_0": full[:, 4], "scale_1": full[:, 5], "scale_2": full[:, 6],
                "rot_0": full[:, 8], "rot_1": full[:, 9], "rot_2": full[:, 10], "rot_3": full[:, 11]}
        for c in range(3):
            cols["f_dc_%d" % c] = full[:, 16 + c]
            for i in range(15):
                cols["f_rest_%d" % (c * 15 + i)] = full[:, 16 + 4 * (i + 1) + c]
        f.write(np.stack([cols[p] for p in names], axis=1).astype("<f4").tobytes())
del scene
orbit = np.stack([synth.orbit_camera(k, W, H).uniforms(W, H) for k in range(64)]).astype("<f4")
opath = os.path.join(d, "orbit.bin")
orbit.tofile(opath)
r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "host_check.js"), "bench", path, str(W), str(H), str(ts), opath, str(frames)],
                   capture_output=True, text=True, timeout=900)
os.remove(path)
if r.returncode != 0:
    print("node bench failed:", r.stderr[-2000:])
    sys.exit(1)
out = json.loads(r.stdout.strip().splitlines()[-1])
out["workload"] = "bicycle-like synthetic %d gaussians @%dx%d, 64-step orbit, %d frames per mode" % (n, W, H, frames)
print(json.dumps(out, indent=1))
