"""Python mirror of the reference's ``Renderer`` (renderer.ts:35-594) on top of the C ABI.

Same constructor shape -- Renderer(canvas, interactiveCamera, device, gaussians, tileSize) -- where
``canvas`` is anything with width/height, ``device`` is a HIP device ordinal and ``gaussians`` has
``numGaussians`` and ``gaussiansBuffer`` (the 320-byte records PackedGaussians builds).
"""
import ctypes

import numpy as np

from . import _abi
from ._abi import GsConfig, GsStats, check


class Canvas:
    def __init__(self, width, height):
        self.width = int(width)
        self.height = int(height)


class PackedGaussians:
    """Holds the packed 320-byte records (ply.ts:32-47 fields the renderer consumes)."""

    def __init__(self, records):
        rec = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 80)
        self.numGaussians = rec.shape[0]
        self.gaussiansBuffer = rec
        self.sphericalHarmonicsDegree = 3


    @staticmethod
    def from_ply(path):
        """Native loader (gs_ply_load): the reference's PackedGaussians(arrayBuffer) for a .ply on disk."""
        rec, deg = _abi.load_ply(path)
        pg = PackedGaussians(rec)
        pg.sphericalHarmonicsDegree = deg
        return pg


class InteractiveCamera:
    """camera.ts:193-308 without the DOM callbacks: dirty flag + camera."""

    def __init__(self, camera):
        self._camera = camera
        self._dirty = True

    def setNewCamera(self, camera):
        self._camera = camera
        self._dirty = True

    def isDirty(self):
        return self._dirty

    def getCamera(self):
        self._dirty = False
        return self._camera


class Renderer:
    def __init__(self, canvas, interactiveCamera, device, gaussians, tileSize=16, *, flags=0, cols=None,
                 max_intersections=0, stream=None, share_with=None):
        """share_with: another Renderer on the same device whose resident splats this one renders (gs_share_splats)
        instead of uploading `gaussians` again."""
        self.canvas = canvas
        self.interactiveCamera = interactiveCamera
        self.device = int(device)
        self.tileSize = int(tileSize)
        self.numGaussians = gaussians.numGaussians
        self.numIntersections = 0
        self.numFrames = 0
        self._L = _abi.load()
        cfg = GsConfig()
        cfg.struct_size = ctypes.sizeof(GsConfig)
        cfg.width, cfg.height, cfg.tile_size = canvas.width, canvas.height, self.tileSize
        cfg.device = self.device
        cfg.col_begin, cfg.col_end = cols if cols is not None else (0, 0)
        cfg.flags = flags
        cfg.max_intersections = int(max_intersections)
        cfg.stream = stream
        self._ctx = ctypes.c_void_p()
        check(self._L.gs_create(ctypes.byref(cfg), ctypes.byref(self._ctx)))
        self.flags = flags
        buf = gaussians.gaussiansBuffer if share_with is None else None
        self._owner = share_with  # keeps the owner of borrowed splats alive
        if share_with is not None:
            check(self._L.gs_share_splats(self._ctx, share_with._ctx))
        elif hasattr(buf, "data_ptr"):  # a device tensor: no PCIe copy
            # the repack runs on the context's stream, which is not ordered against the stream that produced the tensor:
            # the records must be complete before the call (a scene still being generated was repacked half-written when
            # several processes shared one GPU)
            import torch
            torch.cuda.current_stream(buf.device).synchronize()
            check(self._L.gs_upload_splats_device(self._ctx, buf.data_ptr(), self.numGaussians))
        else:
            arr = np.ascontiguousarray(buf, dtype=np.float32)
            check(self._L.gs_upload_splats(self._ctx, arr.ctypes.data, self.numGaussians))
        x0, w = ctypes.c_uint32(), ctypes.c_uint32()
        check(self._L.gs_slab_width(self._ctx, ctypes.byref(x0), ctypes.byref(w)))
        self.slab_x0, self.slab_width = x0.value, w.value

    # -- frame -------------------------------------------------------------------------------------
    def render_uniforms(self, uniforms, debug=False, out_ptr=None):
        u = np.ascontiguousarray(uniforms, dtype=np.float32).reshape(40)
        if out_ptr is not None:
            check(self._L.gs_render_to(self._ctx, u.ctypes.data, out_ptr))
        elif debug:
            check(self._L.gs_render_debug(self._ctx, u.ctypes.data))
        else:
            check(self._L.gs_render(self._ctx, u.ctypes.data))
        self.numFrames += 1

    def animate(self, debug=False):
        """One Renderer.animate() tick (renderer.ts:349-593): renders only when the camera is dirty."""
        if self._ctx is None:
            raise RuntimeError("renderer destroyed")
        if not self.interactiveCamera.isDirty():
            return False
        cam = self.interactiveCamera.getCamera()
        self.render_uniforms(cam.uniforms(self.canvas.width, self.canvas.height), debug=debug)
        return True

    def wait(self):
        check(self._L.gs_wait(self._ctx))

    def set_option(self, key, value):
        check(self._L.gs_set_option(self._ctx, key, int(value)))

    # -- outputs -----------------------------------------------------------------------------------
    def read_rgba8(self):
        out = np.empty((self.canvas.height, self.slab_width, 4), dtype=np.uint8)
        check(self._L.gs_read_rgba8(self._ctx, out.ctypes.data, out.nbytes))
        return out

    def read_buffer(self, which, dtype=np.uint32):
        n = ctypes.c_uint64()
        check(self._L.gs_read_buffer(self._ctx, which, None, 0, ctypes.byref(n)))
        out = np.empty(n.value // np.dtype(dtype).itemsize, dtype=dtype)
        if n.value:
            check(self._L.gs_read_buffer(self._ctx, which, out.ctypes.data, out.nbytes, None))
        return out

    def device_ptr(self, which):
        p = ctypes.c_void_p()
        check(self._L.gs_device_ptr(self._ctx, which, ctypes.byref(p)))
        return p.value

    def stats(self):
        s = GsStats()
        check(self._L.gs_get_stats(self._ctx, ctypes.byref(s)))
        d = {k: getattr(s, k) for k in ("num_gaussians", "num_visible", "num_intersections", "num_processed", "num_tiles",
                                        "sort_passes", "frames", "frame_us", "frame_us_mean", "frames_timed", "num_evaluated", "depth_ordered",
                                        "capacity", "max_intersections_seen", "truncated_frames", "tight_binning", "frames_in_flight", "graph_frames",
                                        "num_row_items", "num_row_slots", "row_capacity")}
        d["stage_us"] = {n: s.stage_us[i] for i, n in enumerate(_abi.GS_STAGE_NAMES)}
        d["stage_us_mean"] = {n: s.stage_us_mean[i] for i, n in enumerate(_abi.GS_STAGE_NAMES)}
        self.numIntersections = d["num_intersections"]
        return d

    def assemble(self, d_slabs_ptr, col_bounds, slab_stride_bytes, d_image_ptr):
        cb = (ctypes.c_uint32 * len(col_bounds))(*col_bounds)
        check(self._L.gs_assemble_slabs(self._ctx, d_slabs_ptr, cb, len(col_bounds) - 1, slab_stride_bytes, d_image_ptr))

    def destroy(self):
        """Renderer.destroy (renderer.ts:90-94); safe to call twice and before the first frame."""
        if self._ctx is not None:
            check(self._L.gs_destroy(self._ctx))
            self._ctx = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class PipelinedRenderer:
    """K frames in flight: K contexts (one uploads the splats, the others borrow them) rendered round-robin, each on its
    own stream with its own per-frame buffers, so frame k's blend overlaps frame k+1's binning and sort.  The reference
    keeps one frame in flight (Renderer.animate awaits every stage, renderer.ts:394-587); results per frame are identical.
    `render_uniforms` returns the slot used; `wait(slot)` / `read_rgba8(slot)` address a frame; `wait()` drains all."""

    def __init__(self, canvas, interactiveCamera, device, gaussians, tileSize=16, *, frames_in_flight=2, **kw):
        if frames_in_flight < 1:
            raise ValueError("frames_in_flight must be >= 1")
        first = Renderer(canvas, interactiveCamera, device, gaussians, tileSize, **kw)
        self.renderers = [first] + [Renderer(canvas, interactiveCamera, device, gaussians, tileSize, share_with=first, **kw)
                                    for _ in range(frames_in_flight - 1)]
        for r in self.renderers:  # this class IS the explicit form of the library's own ring: its members render one frame at a time
            r.set_option(_abi.GS_OPT_FRAMES_IN_FLIGHT, 1)
        self.canvas, self.interactiveCamera = canvas, interactiveCamera
        self._next = 0
        self._busy = [False] * frames_in_flight
        self.numFrames = 0

    def render_uniforms(self, uniforms):
        slot = self._next
        r = self.renderers[slot]
        if self._busy[slot]:
            r.wait()  # the frame this context rendered K steps ago must be complete before its buffers are reused
        r.render_uniforms(uniforms)
        self._busy[slot] = True
        self._next = (slot + 1) % len(self.renderers)
        self.numFrames += 1
        return slot

    def animate(self):
        if not self.interactiveCamera.isDirty():
            return None
        cam = self.interactiveCamera.getCamera()
        return self.render_uniforms(cam.uniforms(self.canvas.width, self.canvas.height))

    def wait(self, slot=None):
        for k, r in enumerate(self.renderers):
            if (slot is None or slot == k) and self._busy[k]:
                r.wait()
                self._busy[k] = False

    def read_rgba8(self, slot):
        self.wait(slot)
        return self.renderers[slot].read_rgba8()

    def set_option(self, key, value):
        for r in self.renderers:
            r.set_option(key, value)

    def destroy(self):
        for r in reversed(self.renderers):  # borrowers first
            r.destroy()
