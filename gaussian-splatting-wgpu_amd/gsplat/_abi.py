"""ctypes binding of include/gsplat/gs_abi.h (libgsplat_hip.so).

There is no CPU fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# $GSPLAT_LIB: another build of the same library (A/B measurements of compiler flags); never a different implementation
LIB_PATH = os.environ.get("GSPLAT_LIB") or os.path.abspath(os.path.join(_HERE, "..", "lib", "libgsplat_hip.so"))

GS_FLAG_EXACT_BLEND = 0x1
GS_FLAG_F32_TAP = 0x2
GS_FLAG_TIMING = 0x4

GS_STAGE_NAMES = ("preprocess", "scan", "emit", "sort", "ranges", "blend")

(GS_BUF_TILE_COUNTS, GS_BUF_TILE_OFFSETS, GS_BUF_GAUSSIAN_DATA, GS_BUF_KEYS_UNSORTED, GS_BUF_VALUES_UNSORTED, GS_BUF_KEYS,
 GS_BUF_VALUES, GS_BUF_RANGES, GS_BUF_RGBA8, GS_BUF_RGB_F32, GS_BUF_BLOCK_MASKS) = range(11)

GS_OPT_BLEND_ABLATION = 1
GS_OPT_PERSISTENT_GRID = 2
GS_OPT_RESET_TIMING = 3
GS_OPT_EMIT_ORDER = 4
GS_OPT_UNFUSED = 5
GS_OPT_DEBUG_VIEW = 6
GS_OPT_TILE_CULL = 7
GS_OPT_FRAMES_IN_FLIGHT = 8
GS_OPT_FRAME_GRAPH = 9
GS_OPT_PROJ_CHUNKS = 10

# every symbol include/gsplat/gs_abi.h declares
ABI_SYMBOLS = ("gs_last_error", "gs_abi_version", "gs_create", "gs_destroy", "gs_upload_splats", "gs_upload_splats_device",
               "gs_share_splats",
               "gs_ply_load", "gs_ply_free", "gs_upload_ply",
               "gs_render", "gs_render_debug", "gs_render_to", "gs_wait", "gs_render_host", "gs_wait_ticket", "gs_host_alloc", "gs_host_free", "gs_read_rgba8", "gs_read_buffer", "gs_device_ptr",
               "gs_get_stats", "gs_set_option", "gs_slab_width", "gs_assemble_slabs", "gs_sort_pairs_u32",
               "gs_exclusive_scan_u32")


class GsConfig(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("tile_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("col_begin", ctypes.c_uint32),
                ("col_end", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("max_intersections", ctypes.c_uint64),
                ("stream", ctypes.c_void_p)]


class GsStats(ctypes.Structure):
    _fields_ = [("num_gaussians", ctypes.c_uint64), ("num_visible", ctypes.c_uint64), ("num_intersections", ctypes.c_uint64),
                ("num_processed", ctypes.c_uint64), ("num_tiles", ctypes.c_uint32), ("sort_passes", ctypes.c_uint32),
                ("frames", ctypes.c_uint64), ("stage_us", ctypes.c_float * 6), ("frame_us", ctypes.c_float),
                ("stage_us_mean", ctypes.c_float * 6), ("frame_us_mean", ctypes.c_float), ("frames_timed", ctypes.c_uint32), ("depth_ordered", ctypes.c_uint32), ("num_evaluated", ctypes.c_uint64),
                ("capacity", ctypes.c_uint64), ("max_intersections_seen", ctypes.c_uint64), ("truncated_frames", ctypes.c_uint64),
                ("tight_binning", ctypes.c_uint32), ("frames_in_flight", ctypes.c_uint32), ("graph_frames", ctypes.c_uint64),
                ("num_row_items", ctypes.c_uint64), ("num_row_slots", ctypes.c_uint64), ("row_capacity", ctypes.c_uint64)]


class GsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gsplat error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Loads libgsplat_hip.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libgsplat_hip.so not built: run `python gaussian-splatting-wgpu_amd/csrc/build.py` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    vp, u64, i32, u32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint32
    L.gs_last_error.restype = ctypes.c_char_p
    L.gs_abi_version.restype = i32
    L.gs_create.argtypes = [ctypes.POINTER(GsConfig), ctypes.POINTER(vp)]
    L.gs_destroy.argtypes = [vp]
    L.gs_upload_splats.argtypes = [vp, vp, u64]
    L.gs_upload_splats_device.argtypes = [vp, vp, u64]
    L.gs_share_splats.argtypes = [vp, vp]
    L.gs_ply_load.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp), ctypes.POINTER(u64), ctypes.POINTER(i32)]
    L.gs_ply_free.argtypes = [vp]
    L.gs_upload_ply.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(u64)]
    L.gs_render.argtypes = [vp, vp]
    L.gs_render_debug.argtypes = [vp, vp]
    L.gs_render_to.argtypes = [vp, vp, vp]
    L.gs_wait.argtypes = [vp]
    L.gs_render_host.argtypes = [vp, vp, vp, u64, ctypes.POINTER(u64)]
    L.gs_wait_ticket.argtypes = [vp, u64]
    L.gs_host_alloc.argtypes = [u64, ctypes.POINTER(vp)]
    L.gs_host_free.argtypes = [vp]
    L.gs_read_rgba8.argtypes = [vp, vp, u64]
    L.gs_read_buffer.argtypes = [vp, i32, vp, u64, ctypes.POINTER(u64)]
    L.gs_device_ptr.argtypes = [vp, i32, ctypes.POINTER(vp)]
    L.gs_get_stats.argtypes = [vp, ctypes.POINTER(GsStats)]
    L.gs_set_option.argtypes = [vp, i32, ctypes.c_int64]
    L.gs_slab_width.argtypes = [vp, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.gs_assemble_slabs.argtypes = [vp, vp, ctypes.POINTER(u32), u32, u64, vp]
    L.gs_sort_pairs_u32.argtypes = [i32, vp, vp, u64, u32]
    L.gs_exclusive_scan_u32.argtypes = [i32, vp, u64, ctypes.POINTER(u64)]
    for name in ABI_SYMBOLS:
        if name not in ("gs_last_error", "gs_ply_free", "gs_host_free"):
            getattr(L, name).restype = i32
    L.gs_ply_free.restype = None
    L.gs_host_free.restype = None
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise GsError(rc, load().gs_last_error().decode("utf-8", "replace"))


def sort_pairs(keys, values=None, key_bits=32, device=0):
    """GPUSorter.sort (radix_sort/sort.ts:341-350) on host arrays; returns sorted copies."""
    k = np.array(keys, dtype=np.uint32, copy=True)
    v = None if values is None else np.array(values, dtype=np.uint32, copy=True)
    check(load().gs_sort_pairs_u32(device, k.ctypes.data, None if v is None else v.ctypes.data, k.size, key_bits))
    return k, v


def exclusive_scan(data, device=0):
    """ExclusiveScanner.scan (exclusive_scan.ts:208-325): returns (offsets, total)."""
    d = np.array(data, dtype=np.uint32, copy=True)
    total = ctypes.c_uint64(0)
    check(load().gs_exclusive_scan_u32(device, d.ctypes.data, d.size, ctypes.byref(total)))
    return d, int(total.value)


def load_ply(path):
    """Native PackedGaussians (ply.ts:162-228): returns (records float32 [n,80], sh_degree)."""
    L = load()
    rec, n, deg = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_int32()
    check(L.gs_ply_load(str(path).encode(), ctypes.byref(rec), ctypes.byref(n), ctypes.byref(deg)))
    try:
        arr = np.ctypeslib.as_array(ctypes.cast(rec, ctypes.POINTER(ctypes.c_float)), shape=(max(n.value, 1) * 80,))[: n.value * 80]
        return arr.reshape(n.value, 80).copy(), deg.value
    finally:
        L.gs_ply_free(rec)
