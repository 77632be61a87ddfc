#!/usr/bin/env python3
"""Builds libgsplat_hip.so (gfx950 kernels + runtime + C ABI) and, when node headers are present,
the N-API addon.  hipcc cross-compiles without a GPU.  Outputs stay in-tree (git-ignored)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "lib")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -ffp-contract=off: one IEEE rounding per written operation (canonical semantics);
# fused operations are spelled __builtin_fmaf where they are wanted.
# -fno-slp-vectorize: the SLP vectoriser turns adjacent scalar f32 multiplies/adds into v_pk_mul_f32 / v_pk_fma_f32, which issue
# slower than the two scalar instructions they replace on gfx950 and need register pairs (extra moves, waits right behind
# loads): projection 282 -> 243 us, blend 738 -> 712 us with the flag (profiles/README.md).
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
FILE_FLAGS = {}  # per-file flags (none at present)
SOURCES = ["k_preprocess.hip", "k_binning.hip", "k_gsort.hip", "k_rows.hip", "k_sort.hip", "k_blend.hip", "gs_runtime.hip"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, profiling=False):
    """profiling=True: -DGS_PROFILING, a separate libgsplat_hip_prof.so with the blend's ablation branches and per-walker stamps
    (tools/blend_profile.py, tools/sweep.sh); the product library does not contain them."""
    os.makedirs(OUT, exist_ok=True)
    objdir = os.path.join(OUT, "obj_prof" if profiling else "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(HERE, h) for h in ("gs_device.h", "gs_kernels.h", "gs_tight.h")] + [
        os.path.join(HERE, "..", "..", "include", "gsplat", "gs_abi.h")]

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(HERE, src)
        if force or _stale(obj, [path] + headers):
            cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + (["-DGS_PROFILING"] if profiling else []) + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    so = os.path.join(OUT, "libgsplat_hip_prof.so" if profiling else "libgsplat_hip.so")
    if force or _stale(so, objs):
        subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", so] + objs)
    return os.path.abspath(so)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, profiling="--profiling" in sys.argv))


def build_napi(force=False, verbose=False):
    """Plain-C N-API addon (no node-gyp): gcc against /usr/include/node, linked to libgsplat_hip.so
    with an $ORIGIN rpath.  Skipped (returns None) when the node headers are not installed."""
    inc = "/usr/include/node"
    if not os.path.exists(os.path.join(inc, "node_api.h")):
        return None
    so = build(force=force, verbose=verbose)
    src = os.path.join(HERE, "napi", "gs_napi.c")
    out = os.path.join(OUT, "gsplat_napi.node")
    if force or _stale(out, [src, so, os.path.join(HERE, "..", "..", "include", "gsplat", "gs_abi.h")]):
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-DNODE_GYP_MODULE_NAME=gsplat_napi", "-I" + inc, src, "-o", out,
               "-L" + OUT, "-lgsplat_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return os.path.abspath(out)
