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


def balanced_bounds(col_cost, world):
    """Tile-column boundaries (world+1 entries) of the contiguous partition whose most expensive slab is cheapest.
    `col_cost[x]` = cost of tile column x, e.g. its instance count in a calibration frame (the per-tile counts of a whole-canvas
    render summed over the rows): a rank's frame time is a fixed part plus a part proportional to its instances, and the
    centre columns of a scene hold several times the instances of the border columns.  Every slab gets >= 1 column."""
    cost = np.asarray(col_cost, dtype=np.float64).reshape(-1)
    ntx = cost.shape[0]
    if world > ntx:
        raise ValueError("more ranks (%d) than tile columns (%d)" % (world, ntx))
    pre = np.concatenate([[0.0], np.cumsum(cost)])
    # best[g][x] = smallest possible maximum over g slabs covering columns [0, x)
    inf = float("inf")
    best = np.full((world + 1, ntx + 1), inf)
    cut = np.zeros((world + 1, ntx + 1), dtype=np.int64)
    best[0][0] = 0.0
    for g in range(1, world + 1):
        for x in range(g, ntx - (world - g) + 1):
            for y in range(g - 1, x):  # the last slab is [y, x)
                v = max(best[g - 1][y], pre[x] - pre[y])
                if v < best[g][x]:
                    best[g][x], cut[g][x] = v, y
    bounds = [ntx]
    for g in range(world, 0, -1):
        bounds.append(int(cut[g][bounds[-1]]))
    return bounds[::-1]


def slab_pixels(bounds, width, tile_size):
    """Pixel [begin, end) of every slab."""
    return [(b0 * tile_size, min(width, b1 * tile_size)) for b0, b1 in zip(bounds[:-1], bounds[1:])]


class SlabExchange:
    """Gathers per-rank slab images and assembles the frame.  Works on any torch.distributed backend
    (tested with gloo on CPU); on GPUs the assembly runs in the library (gs_assemble_slabs)."""

    def __init__(self, width, height, tile_size, world, rank, device, renderer=None, bounds=None, collective="all_gather", root=0):
        """bounds: tile-column boundaries (default: even slabs).  collective: "all_gather" (every rank ends up with every slab) or
        "gather" (only `root`, the presenting rank, receives them: RCCL send/recv to the root, 1/world of the fabric bytes)."""
        import torch
        self.torch = torch
        self.W, self.H, self.ts, self.world, self.rank = width, height, tile_size, world, rank
        self.bounds = list(bounds) if bounds is not None else slab_bounds(width, tile_size, world)
        if len(self.bounds) != world + 1 or self.bounds[0] != 0 or self.bounds[-1] != num_tile_columns(width, tile_size) \
                or any(a >= b for a, b in zip(self.bounds, self.bounds[1:])):
            raise ValueError("bad slab bounds %r" % (self.bounds,))
        if collective not in ("all_gather", "gather"):
            raise ValueError("collective must be all_gather or gather")
        self.collective, self.root = collective, root
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
        if self.world == 1 and not getattr(self, "always_collective", False):
            gathered.copy_(send)
        elif self.collective == "gather" and not (dist.get_backend() == "gloo" and send.is_cuda):
            # only the presenting rank needs the slabs (gloo cannot gather device tensors: the rehearsal path below all-gathers)
            parts = list(gathered.view(self.world, self.stride).unbind(0)) if self.rank == self.root else None
            dist.gather(send, parts, dst=self.root)
        elif dist.get_backend() == "gloo":
            # rehearsal path (CPU tests, several ranks on one GPU): device tensors are staged through the host here, with
            # blocking copies around a CPU all-gather, so the rehearsal checks the frame logic and nothing of gloo's own
            # device-tensor handling
            if send.is_cuda:
                self.torch.cuda.synchronize(send.device)  # the frame is complete
                host = send.cpu()
                parts = [self.torch.empty_like(host) for _ in range(self.world)]
                dist.all_gather(parts, host)
                gathered.copy_(self.torch.cat(parts))
                self.torch.cuda.synchronize(send.device)
            else:
                dist.all_gather(list(gathered.view(self.world, self.stride).unbind(0)), send)
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


class PipelinedSlabs:
    """K frames in flight on every rank.  A rank's slab frame is bound by latencies, not by throughput (its longest tile list, a
    dozen dependent launches), so one frame at a time leaves most of the GPU idle: 418-508 us per frame at 8 slabs of config B
    against 211-275 us with three frames in flight (tools/slab_flight_probe.py).  K slab contexts share the resident splats
    (gs_share_splats), each on its own stream with its own send buffer; frame k is rendered by context k % K, its collective
    and (on the root) the assembly are queued on ONE communication stream in frame order -- the same order on every rank, as
    RCCL requires -- after the frame's event, and the context's next frame (k + K) waits for that collective to have read the
    send buffer.  No host synchronisation inside the frame loop.
    make_renderer(stream_handle, share_with) -> a slab Renderer on that stream (share_with = None: it uploads the splats)."""

    def __init__(self, xch, make_renderer, frames_in_flight=3, owner=None):
        torch = xch.torch
        self.x, self.torch = xch, torch
        self.K = int(frames_in_flight)
        if self.K < 1:
            raise ValueError("frames_in_flight must be >= 1")
        self.streams = [torch.cuda.Stream(xch.device) for _ in range(self.K)]
        self.comm = torch.cuda.Stream(xch.device)
        self.renderers = []
        for k in range(self.K):
            self.renderers.append(make_renderer(self.streams[k].cuda_stream, owner if owner is not None else (self.renderers[0] if k else None)))
        share = owner if owner is not None else self.renderers[0]
        # the assembly (gs_assemble_slabs) runs on its context's stream: a context of its own on the communication stream
        self.assembler = make_renderer(self.comm.cuda_stream, share) if xch.rank == xch.root else None
        if self.assembler is not None:
            xch.renderer = self.assembler
        self.send = [xch.send] + [torch.zeros_like(xch.send) for _ in range(self.K - 1)]
        self.ev_render = [torch.cuda.Event() for _ in range(self.K)]
        self.ev_free = [None] * self.K
        self.k = 0

    def submit(self, uniforms):
        torch = self.torch
        slot = self.k % self.K
        s = self.streams[slot]
        if self.ev_free[slot] is not None:
            s.wait_event(self.ev_free[slot])  # the collective of frame k - K has read this send buffer
        self.renderers[slot].render_uniforms(uniforms, out_ptr=self.send[slot].data_ptr())  # the blend writes the send buffer
        self.ev_render[slot].record(s)
        self.comm.wait_event(self.ev_render[slot])
        with torch.cuda.stream(self.comm):
            self.x.exchange(self.send[slot], self.x.gathered)
            if self.assembler is not None:
                self.x.assemble(self.x.gathered)
            ev = torch.cuda.Event()
            ev.record(self.comm)
        self.ev_free[slot] = ev
        self.k += 1
        return slot

    def finish(self):
        """Drains every context and the communication stream; raises what gs_wait raises (GS_ERR_TRUNCATED ...)."""
        err = None
        for r in self.renderers:
            try:
                r.wait()
            except Exception as e:  # keep draining: the other contexts still hold frames
                err = err or e
        self.comm.synchronize()
        if err is not None:
            raise err

    def destroy(self):
        if self.assembler is not None:
            self.assembler.destroy()
            self.assembler = None
        for r in reversed(self.renderers):
            r.destroy()
        self.renderers = []


class FrameGroupSlabs:
    """G groups of S = world / G ranks render ALTERNATE frames; every frame is still S tile-column slabs gathered to rank 0 and
    assembled there.  A rank's frame costs a fixed part (the O(N) cull and scan, launches too small to fill the chip) plus a part
    proportional to its slab, so fewer, wider slabs per frame use the GPUs better: at 1080p eight slabs project to 3.6x one GPU,
    two groups of four slabs to 4.9x, four groups of two to 6.7x (profiles/r03_slab_per_rank.txt) -- for a frame latency G
    times longer.  bench.py's default from 4 ranks on (--frame-groups 0: world / 2 groups of two ranks); rehearsed over gloo with
    four processes on one GPU and over RCCL with a world of one rank, never run on multi-GPU hardware.

    Frame k belongs to group k % G.  Rank r is slab r % S of group r // S.  Communicator h holds the ranks of group h and, for
    h > 0, rank 0, which receives every frame; rank 0 issues the gathers of ALL frames in frame order on its communication
    stream, the other ranks only those of their own frames.  Within a rank: K slab contexts in flight as in PipelinedSlabs."""

    def __init__(self, width, height, tile_size, world, rank, device, groups, bounds, make_renderer, frames_in_flight=3, owner=None):
        import torch
        import torch.distributed as dist
        if groups < 1 or world % groups:
            raise ValueError("the world (%d) must be a multiple of the frame groups (%d)" % (world, groups))
        self.torch, self.dist = torch, dist
        self.G, self.S = groups, world // groups
        self.rank, self.group, self.slab = rank, rank // self.S, rank % self.S
        # the geometry of ONE frame: S slabs; this rank's slab index inside it
        self.x = SlabExchange(width, height, tile_size, self.S, self.slab, device, bounds=bounds, collective="gather")
        self.pgs = []
        for h in range(groups):  # every rank creates every communicator, in the same order
            ranks = list(range(h * self.S, (h + 1) * self.S))
            self.pgs.append(dist.new_group(ranks if h == 0 else [0] + ranks))
        self.K = int(frames_in_flight)
        self.streams = [torch.cuda.Stream(device) for _ in range(self.K)]
        self.comm = torch.cuda.Stream(device)
        self.renderers = [make_renderer(self.streams[k].cuda_stream, owner) for k in range(self.K)]
        self.assembler = make_renderer(self.comm.cuda_stream, owner) if rank == 0 else None
        if self.assembler is not None:
            self.x.renderer = self.assembler
        self.send = [torch.zeros_like(self.x.send) for _ in range(self.K)]
        self.dummy = torch.zeros_like(self.x.send)
        # rank 0 receives S slabs, behind its own dummy contribution when the frame is another group's
        self.gathered = torch.zeros(self.x.stride * (self.S + 1), dtype=torch.uint8, device=device) if rank == 0 else None
        self.ev_render = [torch.cuda.Event() for _ in range(self.K)]
        self.ev_free = [None] * self.K
        self.k = 0       # frames submitted
        self.mine = 0    # frames this rank rendered
        self.last_own = None  # (frame index, slot) of the last frame this rank rendered

    def _collective(self, h, send):
        """The gather of one frame of group h, on the current (communication) stream."""
        torch, dist, x = self.torch, self.dist, self.x
        n = self.S + (1 if h else 0)  # ranks in communicator h
        if dist.get_backend() == "gloo" and send.is_cuda:  # rehearsal: staged through the host (SlabExchange.exchange)
            torch.cuda.synchronize(send.device)
            host = send.cpu()
            parts = [torch.empty_like(host) for _ in range(n)]
            dist.all_gather(parts, host, group=self.pgs[h])
            if self.rank == 0:
                self.gathered[: n * x.stride].copy_(torch.cat(parts))
            torch.cuda.synchronize(send.device)
        else:
            parts = list(self.gathered[: n * x.stride].view(n, x.stride).unbind(0)) if self.rank == 0 else None
            dist.gather(send, parts, dst=0, group=self.pgs[h])
        if self.rank == 0:
            first = x.stride if h else 0  # skip rank 0's dummy
            x.assemble(self.gathered[first: first + self.S * x.stride])

    def submit(self, uniforms):
        torch = self.torch
        h = self.k % self.G
        if h == self.group:
            slot = self.mine % self.K
            s = self.streams[slot]
            if self.ev_free[slot] is not None:
                s.wait_event(self.ev_free[slot])
            self.renderers[slot].render_uniforms(uniforms, out_ptr=self.send[slot].data_ptr())
            self.ev_render[slot].record(s)
            self.comm.wait_event(self.ev_render[slot])
            with torch.cuda.stream(self.comm):
                self._collective(h, self.send[slot])
                ev = torch.cuda.Event()
                ev.record(self.comm)
            self.ev_free[slot] = ev
            self.last_own = (self.k, slot)
            self.mine += 1
        elif self.rank == 0:
            with torch.cuda.stream(self.comm):
                self._collective(h, self.dummy)
        self.k += 1

    def finish(self):
        err = None
        for r in self.renderers:
            try:
                r.wait()
            except Exception as e:
                err = err or e
        self.comm.synchronize()
        if err is not None:
            raise err

    def destroy(self):
        if self.assembler is not None:
            self.assembler.destroy()
            self.assembler = None
        for r in reversed(self.renderers):
            r.destroy()
        self.renderers = []
