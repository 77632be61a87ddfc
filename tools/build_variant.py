#!/usr/bin/env python3
"""A/B of compile-time knobs: builds gaussian-splatting-wgpu_amd/lib/variants/<name>.so from the product sources with extra -D flags
(objects of the sources that do not mention a flag's macro are shared with the product build).  bench.py / tools pick it with
GSPLAT_LIB=<path>.  Usage: tools/build_variant.py <name> -DRA_MINW=4 -DRA_GRP=6 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd", "csrc"))
import build as B
name, flags = sys.argv[1], sys.argv[2:]
B.build()
macros = [f[2:].split("=")[0] for f in flags if f.startswith("-D")]
vdir = os.path.join(B.OUT, "variants"); odir = os.path.join(vdir, "obj_" + name)
os.makedirs(odir, exist_ok=True)
objs = []
for src in B.SOURCES:
    path = os.path.join(B.HERE, src)
    text = open(path).read() + "".join(open(os.path.join(B.HERE, h)).read() for h in ("gs_device.h", "gs_tight.h", "gs_kernels.h"))
    if any(m in text for m in macros):
        obj = os.path.join(odir, src.replace(".hip", ".o"))
        subprocess.check_call([B.HIPCC] + B.FLAGS + B.FILE_FLAGS.get(src, []) + flags + ["-c", path, "-o", obj])
    else:
        obj = os.path.join(B.OUT, "obj", src.replace(".hip", ".o"))
    objs.append(obj)
so = os.path.join(vdir, name + ".so")
subprocess.check_call([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", so] + objs)
print(so)
