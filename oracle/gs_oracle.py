"""ctypes wrapper of the CPU oracle (oracle/gs_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py; never by the product package.  Every function cites the reference stage it checks.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgs_oracle.so")

_u32p = ctypes.POINTER(ctypes.c_uint32)
_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "gs_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.gso_expf.restype = ctypes.c_float
        L.gso_expf.argtypes = [ctypes.c_float]
        L.gso_preprocess.restype = None
        L.gso_preprocess.argtypes = [_f32p, ctypes.c_uint64, _f32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                     ctypes.c_uint32, ctypes.c_uint32, _u32p, _u32p]
        L.gso_scan.restype = ctypes.c_uint32
        L.gso_scan.argtypes = [_u32p, ctypes.c_uint64, _u32p]
        L.gso_emit.restype = None
        L.gso_emit.argtypes = [_u32p, _u32p, _u32p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                               ctypes.c_uint32, _u32p, _u32p]
        L.gso_sort.restype = None
        L.gso_sort.argtypes = [_u32p, _u32p, ctypes.c_uint64, _u32p, _u32p]
        L.gso_ranges.restype = None
        L.gso_ranges.argtypes = [_u32p, ctypes.c_uint64, ctypes.c_uint32, _u32p]
        L.gso_blend.restype = None
        L.gso_blend.argtypes = [_u32p, _u32p, _u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                ctypes.c_uint32, _u8p, _f32p, _u8p, _u32p]
        L.gso_instance_masks.restype = None
        L.gso_instance_masks.argtypes = [_u32p, _u32p, _u32p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, _u32p]
        L.gso_num_threads.restype = ctypes.c_int
        L.gso_set_num_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def num_tiles(W, H, ts):
    """ceil(f32(W)/f32(ts)) as in process_gaussians.wgsl:79."""
    ntx = int(np.ceil(np.float32(W) / np.float32(ts)))
    nty = int(np.ceil(np.float32(H) / np.float32(ts)))
    return ntx, nty


def expf(x):
    return float(lib().gso_expf(float(np.float32(x))))


def preprocess(splats, uniforms, W, H, ts=16, cols=None):
    """process_gaussians.wgsl:35-106.  splats: float32 [N,80]; uniforms: float32 [40].
    Returns (gdata uint32 [N,16], tile_counts uint32 [N])."""
    splats = np.ascontiguousarray(splats, dtype=np.float32).reshape(-1, 80)
    uniforms = np.ascontiguousarray(uniforms, dtype=np.float32).reshape(40)
    n = splats.shape[0]
    ntx, _ = num_tiles(W, H, ts)
    c0, c1 = cols if cols is not None else (0, ntx)
    gdata = np.zeros((n, 16), dtype=np.uint32)
    counts = np.zeros(n, dtype=np.uint32)
    lib().gso_preprocess(_p(splats, _f32p), n, _p(uniforms, _f32p), W, H, ts, c0, c1, _p(gdata, _u32p), _p(counts, _u32p))
    return gdata, counts


def scan(counts):
    """exclusive_scan.ts:208-325.  Returns (offsets, total)."""
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    offsets = np.zeros_like(counts)
    total = lib().gso_scan(_p(counts, _u32p), counts.size, _p(offsets, _u32p))
    return offsets, int(total)


def emit(gdata, offsets, counts, total, W, ts=16, cols=None, H=None):
    """write_tile_ids.wgsl:18-35.  Returns (keys, values) of length total."""
    ntx = int(np.ceil(np.float32(W) / np.float32(ts)))
    c0, c1 = cols if cols is not None else (0, ntx)
    keys = np.zeros(total, dtype=np.uint32)
    values = np.zeros(total, dtype=np.uint32)
    lib().gso_emit(_p(gdata, _u32p), _p(offsets, _u32p), _p(counts, _u32p), gdata.shape[0], W, ts, c0, c1,
                   _p(keys, _u32p), _p(values, _u32p))
    return keys, values


def sort(keys, values):
    """sort.ts:341-350 (stable ascending by full key).  Returns sorted copies."""
    k = np.array(keys, dtype=np.uint32, copy=True)
    v = np.array(values, dtype=np.uint32, copy=True)
    tk = np.empty_like(k)
    tv = np.empty_like(v)
    lib().gso_sort(_p(k, _u32p), _p(v, _u32p), k.size, _p(tk, _u32p), _p(tv, _u32p))
    return k, v


def ranges(sorted_keys, T):
    """compute_ranges.wgsl:5-29 (canonical result)."""
    sorted_keys = np.ascontiguousarray(sorted_keys, dtype=np.uint32)
    r = np.zeros(T, dtype=np.uint32)
    lib().gso_ranges(_p(sorted_keys, _u32p), sorted_keys.size, T, _p(r, _u32p))
    return r


def blend(gdata, sorted_values, rng, W, H, ts=16, cols=None, want_f32=True, want_illcond=False, want_processed=False):
    """compute_tiles.wgsl:30-75.  Returns dict(rgba8 [H,W,4], rgbf [H,W,3], illcond [H,W], processed [T])."""
    ntx, nty = num_tiles(W, H, ts)
    c0, c1 = cols if cols is not None else (0, ntx)
    rgba8 = np.zeros((H, W, 4), dtype=np.uint8)
    rgbf = np.zeros((H, W, 3), dtype=np.float32) if want_f32 else None
    ill = np.zeros((H, W), dtype=np.uint8) if want_illcond else None
    proc = np.zeros(ntx * nty, dtype=np.uint32) if want_processed else None
    sorted_values = np.ascontiguousarray(sorted_values, dtype=np.uint32)
    lib().gso_blend(_p(gdata, _u32p), _p(sorted_values, _u32p), _p(rng, _u32p), W, H, ts, c0, c1, _p(rgba8, _u8p),
                    _p(rgbf, _f32p), _p(ill, _u8p), _p(proc, _u32p))
    return {"rgba8": rgba8, "rgbf": rgbf, "illcond": ill, "processed": proc}


def instance_masks(gdata, keys, values, W, H, ts=16):
    """Checker for the product path's tight binning: per instance of a (key, value) list, one bit per 8x8 pixel block of
    its tile that holds a pixel passing `power <= 0 && alpha >= 1/255` (compute_tiles.wgsl:57-63).  mask 0 = the instance
    cannot change any pixel."""
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    values = np.ascontiguousarray(values, dtype=np.uint32)
    masks = np.zeros(keys.size, dtype=np.uint32)
    lib().gso_instance_masks(_p(gdata, _u32p), _p(keys, _u32p), _p(values, _u32p), keys.size, W, H, ts, _p(masks, _u32p))
    return masks


def render(splats, uniforms, W, H, ts=16, cols=None, **blend_kw):
    """Whole frame, Renderer.animate (renderer.ts:349-593).  Returns every intermediate buffer."""
    ntx, nty = num_tiles(W, H, ts)
    gdata, counts = preprocess(splats, uniforms, W, H, ts, cols)
    offsets, total = scan(counts)
    keys, values = emit(gdata, offsets, counts, total, W, ts, cols)
    skeys, svalues = sort(keys, values)
    rng = ranges(skeys, ntx * nty)
    out = blend(gdata, svalues, rng, W, H, ts, cols, **blend_kw)
    out.update(gdata=gdata, tile_counts=counts, offsets=offsets, num_intersections=total, keys=keys, values=values,
               sorted_keys=skeys, sorted_values=svalues, ranges=rng)
    return out


def set_num_threads(n):
    lib().gso_set_num_threads(int(n))


def get_num_threads():
    return int(lib().gso_num_threads())
