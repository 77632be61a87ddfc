"""Tile-column slab sharding across the GPUs of one node (no counterpart in the
reference, which drives a single GPUDevice: app.ts:14-23).

One process per GPU.  Rank g owns tile columns [bounds[g], bounds[g+1]) and renders only those:
the slab filters the key emission, tile ids and keys are the global ones, so sort, ranges and blend
are purely local and the union of the slabs is the single-GPU image byte for byte.  The only
exchange is ONE all-gather of the rgba8 slabs (RCCL over xGMI when the backend is "nccl"), followed
by a de-interleave of the column slabs into the row-major frame.
"""
import numpy as np


def num_tile_columns(width, tile_size):
    return int(np.ceil(np.float32(width) / np.float32(tile_size)))


def slab_bounds(width, tile_size, world):
    """Tile-column boundaries, world+1 entries; as even as the column count allows."""
    ntx = num_tile_columns(width, tile_size)
    if world > ntx:
        raise ValueError("more ranks (%d) than tile columns (%d)" % (world, ntx))
    return [ntx * g // world for g in range(world + 1)]


def slab_pixels(bounds, width, tile_size):
    """Pixel [begin, end) of every slab."""
    return [(b0 * tile_size, min(width, b1 * tile_size)) for b0, b1 in zip(bounds[:-1], bounds[1:])]


class SlabExchange:
    """Gathers per-rank slab images and assembles the frame.  Works on any torch.distributed backend
    (tested with gloo on CPU); on GPUs the assembly runs in the library (gs_assemble_slabs)."""

    def __init__(self, width, height, tile_size, world, rank, device, renderer=None):
        import torch
        self.torch = torch
        self.W, self.H, self.ts, self.world, self.rank = width, height, tile_size, world, rank
        self.bounds = slab_bounds(width, tile_size, world)
        self.pixels = slab_pixels(self.bounds, width, tile_size)
        self.max_w = max(e - b for b, e in self.pixels)
        self.stride = height * self.max_w * 4  # bytes per rank in the gathered buffer
        self.device = device
        self.renderer = renderer
        self.send = torch.zeros(self.stride, dtype=torch.uint8, device=device)
        self.gathered = torch.zeros(self.stride * world, dtype=torch.uint8, device=device)
        self.image = torch.zeros((height, width, 4), dtype=torch.uint8, device=device)

    @property
    def cols(self):
        return (self.bounds[self.rank], self.bounds[self.rank + 1])

    def exchange(self, send=None, gathered=None):
        """all-gather of the send buffers (self.send -> self.gathered unless given); every rank ends up with every slab."""
        import torch.distributed as dist
        send = self.send if send is None else send
        gathered = self.gathered if gathered is None else gathered
        if self.world == 1:
            gathered.copy_(send)
        elif dist.get_backend() == "gloo":
            # rehearsal path (CPU tests, several ranks on one GPU): gloo stages device tensors through the host on streams of its
            # own, so the frame has to be complete before it reads `send` and its copies back into `gathered` have to be complete
            # before the assembly reads them (with 4 ranks on one GPU the unsynchronised form assembled stale slabs)
            if send.is_cuda:
                self.torch.cuda.synchronize(send.device)
            parts = list(gathered.view(self.world, self.stride).unbind(0))
            dist.all_gather(parts, send)
            if send.is_cuda:
                self.torch.cuda.synchronize(send.device)
        else:
            dist.all_gather_into_tensor(gathered, send)

    def assemble(self, gathered=None):
        """Column slabs -> row-major frame (self.image)."""
        gathered = self.gathered if gathered is None else gathered
        if self.renderer is not None and gathered.is_cuda:
            self.renderer.assemble(gathered.data_ptr(), self.bounds, self.stride, self.image.data_ptr())
        else:
            for g, (b, e) in enumerate(self.pixels):
                w = e - b
                slab = gathered[g * self.stride: g * self.stride + self.H * w * 4].view(self.H, w, 4)
                self.image[:, b:e, :] = slab
        return self.image


class OverlappedExchange:
    """The all-gather of frame k runs on a side stream while frame k+1 is rendered: two send/gather buffer pairs, the blend of
    frame k writes pair k%2, the collective of that pair waits for the blend's event, the assembly of frame k is queued on the
    render stream after frame k+1 (its gather has had a whole frame to finish).  `finish()` assembles the last frame."""

    def __init__(self, xch, render_stream):
        torch = xch.torch
        self.x, self.torch = xch, torch
        self.render_stream = render_stream
        self.comm_stream = torch.cuda.Stream(xch.device)
        self.send = [xch.send, torch.zeros_like(xch.send)]
        self.gathered = [xch.gathered, torch.zeros_like(xch.gathered)]
        self.ev_render = [torch.cuda.Event(), torch.cuda.Event()]
        self.ev_gather = [None, None]
        self.k = 0
        self.unassembled = None

    def send_ptr(self):
        """Where the blend of the next frame must write; orders that blend after the collective that last read the buffer."""
        slot = self.k & 1
        if self.ev_gather[slot] is not None:
            self.render_stream.wait_event(self.ev_gather[slot])
        return self.send[slot].data_ptr()

    def submit(self, assemble):
        """Call after the frame was enqueued on the render stream."""
        torch = self.torch
        slot = self.k & 1
        self.ev_render[slot].record(self.render_stream)
        self.comm_stream.wait_event(self.ev_render[slot])
        with torch.cuda.stream(self.comm_stream):
            self.x.exchange(self.send[slot], self.gathered[slot])
            ev = torch.cuda.Event()
            ev.record(self.comm_stream)
        self.ev_gather[slot] = ev
        if assemble and self.unassembled is not None:
            self._assemble(self.unassembled)
        self.unassembled = slot
        self.k += 1

    def _assemble(self, slot):
        self.render_stream.wait_event(self.ev_gather[slot])
        self.x.assemble(self.gathered[slot])

    def finish(self, assemble):
        if self.unassembled is not None:
            if assemble:
                self._assemble(self.unassembled)
            else:
                self.render_stream.wait_event(self.ev_gather[self.unassembled])
            self.unassembled = None
