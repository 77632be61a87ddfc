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

    def exchange(self):
        """all-gather of the send buffers; every rank ends up with every slab."""
        import torch.distributed as dist
        if self.world == 1:
            self.gathered.copy_(self.send)
        elif dist.get_backend() == "gloo":
            parts = list(self.gathered.view(self.world, self.stride).unbind(0))
            dist.all_gather(parts, self.send)
        else:
            dist.all_gather_into_tensor(self.gathered, self.send)

    def assemble(self):
        """Column slabs -> row-major frame (self.image)."""
        if self.renderer is not None and self.gathered.is_cuda:
            self.renderer.assemble(self.gathered.data_ptr(), self.bounds, self.stride, self.image.data_ptr())
        else:
            for g, (b, e) in enumerate(self.pixels):
                w = e - b
                slab = self.gathered[g * self.stride: g * self.stride + self.H * w * 4].view(self.H, w, 4)
                self.image[:, b:e, :] = slab
        return self.image
