/*
 * gs_abi.h -- C ABI of the MI355X-native forward Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary for the hot path of ldyken53/gaussian-splatting-wgpu: everything
 * `Renderer` (src/renderer.ts:96-102,349-593) asks of WebGPU per scene and per frame goes through
 * the entry points below.  Plain C: opaque handle, plain pointers and sizes, int32 status codes,
 * no C++ types, no exceptions, no torch types.  The reference has no FFI of its own (it is a
 * browser app); the N-API binding a Node host uses is gaussian-splatting-wgpu_amd/csrc/napi and
 * the binding stubs for other hosts are in INTEGRATION.md.
 *
 * Threading: a gs_ctx is not re-entrant: its functions must not be called concurrently on the same ctx (the one exception is
 * gs_wait_ticket, which another thread may call while the owner enqueues); different ctxs may be used from different threads.
 * Frames in flight: a whole-canvas ctx on its own stream keeps up to GS_OPT_FRAMES_IN_FLIGHT (default 3) frames in flight -- a
 * frame enqueued while the previous one is still on the device is rendered by a SHADOW of the ctx (own stream and per-frame
 * arrays, the same resident splats), gs_render taking turns over that ring.  With K frames enqueued: gs_wait waits for ALL of
 * them (and reports a capacity overflow of an earlier one as GS_ERR_TRUNCATED: only the last frame can be re-rendered);
 * gs_read_rgba8, gs_read_buffer, gs_device_ptr and gs_get_stats describe the LAST frame enqueued (its ring member);
 * gs_render_host / gs_wait_ticket address a frame of their own.  A host that calls gs_wait after every gs_render (as
 * Renderer.animate does) never has more than one frame in flight and sees none of this.  Device work of one ring member is
 * ordered on one HIP stream.
 * Ownership: the ctx owns every device allocation; host pointers passed in are copied before the
 * call returns; output host buffers are caller-allocated.
 */
#ifndef GSPLAT_GS_ABI_H
#define GSPLAT_GS_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_ABI_VERSION 3

/* ---- status codes (every function returns one; message via gs_last_error) ------------------ */
#define GS_OK 0
#define GS_ERR_INVALID_ARGUMENT (-1) /* null pointer, bad size, unsupported tile size ...           */
#define GS_ERR_NO_DEVICE (-2)        /* no HIP device / HIP runtime failure at create             */
#define GS_ERR_HIP (-3)              /* a HIP call failed; gs_last_error has hipGetErrorString      */
#define GS_ERR_OUT_OF_MEMORY (-4)
#define GS_ERR_NO_SCENE (-5)         /* gs_render before gs_upload_splats                           */
#define GS_ERR_NO_FRAME (-6)         /* read-back before any gs_render                              */
#define GS_ERR_DEVICE_FAULT (-7)     /* an in-kernel bounded spin gave up (decoupled look-back)     */
#define GS_ERR_CAPACITY (-8)         /* intersections exceed the hard limit (2^30, radix_sort.wgsl:220-221) */
#define GS_ERR_TRUNCATED (-9)        /* gs_wait: a frame enqueued BEFORE the last one overflowed the (key,value) capacity; its
                                        output came from truncated lists and cannot be re-rendered (the last frame is complete and
                                        the capacity has been grown)                                                          */

/* ---- byte layouts fixed by the reference ------------------------------------------------------
 * splat record   320 B  ply.ts:190-198 / process_gaussians.wgsl:1-7
 *     pos f32x3 @0, log_scale f32x3 @16, rot f32x4 (r,x,y,z) @32, opacity_logit f32 @48,
 *     sh 16 x (f32x3, stride 16) @64
 * uniform block  160 B  renderer.ts:15-24,371-392 / process_gaussians.wgsl:16-25
 *     view mat4 col-major @0, proj(=P*V) @64, cam_pos f32x3 @128, tan_fovx @140, tan_fovy @144,
 *     focal_x @148, focal_y @152, scale_modifier @156
 * GaussianData    64 B  process_gaussians.wgsl:8-15
 *     uv f32x2 @0, conic f32x3 @16, depth @28, color f32x3 @32, opacity @44, rect u32x4 @48
 */
#define GS_SPLAT_RECORD_BYTES 320
#define GS_UNIFORM_BYTES 160
#define GS_GAUSSIAN_DATA_BYTES 64

/* ---- configuration: what `new Renderer(canvas, camera, device, gaussians, tileSize)` fixes ---- */
#define GS_FLAG_EXACT_BLEND 0x1u /* blend with the canonical (as-written, unfused) f32 arithmetic: bit-equal to
                                    the CPU oracle; default is fused f32 + hardware exp2 (<=1e-4 per channel)  */
#define GS_FLAG_F32_TAP 0x2u     /* also keep the un-quantised f32 RGB accumulators (GS_BUF_RGB_F32)            */
#define GS_FLAG_TIMING 0x4u      /* bracket every stage with hipEvents; gs_get_stats returns stage microseconds */

typedef struct gs_config {
    uint32_t struct_size;       /* = sizeof(gs_config); lets the struct grow                                     */
    uint32_t width, height;     /* canvas.width / canvas.height (renderer.ts:157,199,366)                       */
    uint32_t tile_size;         /* 8, 16 or 32 (index.html:20-24, app.ts:29,45)                                 */
    int32_t device;             /* HIP device ordinal                                                           */
    uint32_t col_begin, col_end;/* tile-column slab owned by this ctx, [begin,end); 0,0 = whole screen          */
    uint32_t flags;             /* GS_FLAG_*                                                                     */
    uint64_t max_intersections; /* capacity hint for the (key,value) arrays; 0 = derive from the scene         */
    void* stream;               /* hipStream_t to run on; NULL = the ctx creates its own (so the legacy default stream,
                                   whose handle is 0, cannot be passed: give the ctx a created stream when its work must be
                                   ordered with other work, e.g. a collective)                                      */
} gs_config;

/* ---- per-frame statistics (the reference only console.logs these: renderer.ts:406-590) ------- */
enum {
    GS_STAGE_PREPROCESS = 0, /* process_gaussians.wgsl::main                       */
    GS_STAGE_SCAN = 1,       /* ExclusiveScanner.scan (exclusive_scan.ts:208-325)  */
    GS_STAGE_EMIT = 2,       /* write_tile_ids.wgsl::main                          */
    GS_STAGE_SORT = 3,       /* GPUSorter.sort (sort.ts:341-350)                   */
    GS_STAGE_RANGES = 4,     /* compute_ranges.wgsl::main                          */
    GS_STAGE_BLEND = 5,      /* compute_tiles.wgsl::main (+ the render.wgsl blit, which is the identity) */
    GS_STAGE_COUNT = 6
};

typedef struct gs_stats {
    uint64_t num_gaussians;       /* N                                                        */
    uint64_t num_visible;         /* gaussians that passed the cull (tile count > 0)          */
    uint64_t num_intersections;   /* I: what ExclusiveScanner.scan returns (renderer.ts:419)  */
    uint64_t num_processed;       /* list entries the blend read before tile early-exit (per tile: the deepest of its walkers; the
                                     id words are read 192 entries ahead of the records, so this runs up to 191 ahead of the last
                                     entry whose record was fetched) */
    uint32_t num_tiles;           /* T over the whole canvas                                  */
    uint32_t sort_passes;         /* 8-bit radix passes executed                              */
    uint64_t frames;              /* frames rendered by this ctx                              */
    float stage_us[GS_STAGE_COUNT]; /* per-stage device time of the last frame (GS_FLAG_TIMING) */
    float frame_us;               /* first kernel start -> last kernel end (GS_FLAG_TIMING)    */
    float stage_us_mean[GS_STAGE_COUNT]; /* mean over the frames since GS_OPT_RESET_TIMING (at most the last 256) */
    float frame_us_mean;
    uint32_t frames_timed;        /* frames the means cover                                    */
    uint32_t depth_ordered;       /* 1 if the last frame used the depth-ordered pipeline (GS_OPT_EMIT_ORDER)   */
    uint64_t num_evaluated;       /* blend: (8x8 pixel block, entry) pairs evaluated after the block cull */
    uint64_t capacity;            /* entries the (key,value) arrays hold now                                  */
    uint64_t max_intersections_seen; /* largest I of any frame since GS_OPT_RESET_TIMING (as of the last gs_wait) */
    uint64_t truncated_frames;    /* frames since GS_OPT_RESET_TIMING that overflowed the capacity and could not be re-rendered
                                     (only possible when several frames are enqueued per gs_wait)                 */
    uint32_t tight_binning;       /* 1 if the last frame used the opacity-aware (tight) binning of the product path: its
                                     instance lists are then a subset of the reference's (GS_OPT_TILE_CULL)             */
    uint32_t frames_in_flight;    /* contexts of the ring gs_render alternates between now (GS_OPT_FRAMES_IN_FLIGHT)      */
    uint64_t graph_frames;        /* frames replayed from the captured frame graph (GS_OPT_FRAME_GRAPH), summed over the ring */
    uint64_t num_row_items;       /* tight frames (ABI 3): row items = (gaussian, tile row) runs of tiles the lists were expanded from */
    uint64_t num_row_slots;       /*   ... slots reserved for them (items + rows that turned out empty)                               */
    uint64_t row_capacity;        /* row-item slots the context holds now (grown like `capacity`)                                  */
} gs_stats;

/* ---- debug taps: the buffers the reference author inspected by hand (renderer.ts:423-438,504-519) */
enum {
    GS_BUF_TILE_COUNTS = 0,   /* u32[N]     tileCountBuffer                                   */
    GS_BUF_TILE_OFFSETS = 1,  /* u32[N]     tileOffsetBuffer after the scan (needs gs_render_debug)  */
    GS_BUF_GAUSSIAN_DATA = 2, /* 64 B x N   gaussianDataBuffer (culled records are all-zero)  */
    GS_BUF_KEYS_UNSORTED = 3, /* u32[I]     tile_ids as written by write_tile_ids (needs gs_render_debug) */
    GS_BUF_VALUES_UNSORTED = 4,
    GS_BUF_KEYS = 5,          /* u32[I]     sorted tileIDBuffer                                */
    GS_BUF_VALUES = 6,        /* u32[I]     sorted gaussianIDBuffer                            */
    GS_BUF_RANGES = 7,        /* u32[T]     rangesBuffer                                       */
    GS_BUF_RGBA8 = 8,         /* u8[H][Wslab][4] renderTarget (rgba8unorm), this ctx's slab    */
    GS_BUF_RGB_F32 = 9,       /* f32[H][Wslab][3] (GS_FLAG_F32_TAP)                            */
    GS_BUF_BLOCK_MASKS = 10   /* u32[I]     per sorted instance: one bit per 8x8 pixel block of its tile (row-major, tile_size/8
                                 per row) the blend evaluates it for; all blocks when the frame did not use tight binning  */
};

typedef struct gs_ctx gs_ctx;

/* Thread-local message of the last failing call on this thread. */
const char* gs_last_error(void);
int32_t gs_abi_version(void);

/* Replaces `new Renderer(...)` buffer/pipeline setup (renderer.ts:96-324). */
int32_t gs_create(const gs_config* cfg, gs_ctx** out);
/* Replaces Renderer.destroy()/destroyImpl (renderer.ts:90-94,326-347).  Safe before the first frame. */
int32_t gs_destroy(gs_ctx* ctx);

/* Replaces the pointDataBuffer upload (renderer.ts:130-137): takes the exact bytes of
 * PackedGaussians.gaussiansBuffer (ply.ts:204-220), n records of 320 B, host memory.  The records
 * are re-laid-out on the device; the caller's buffer is not referenced after return. */
int32_t gs_upload_splats(gs_ctx* ctx, const void* aos320, uint64_t n);
/* Native PackedGaussians (ply.ts:162-228): parses a binary little-endian 3DGS .ply with the reference's header and
 * property rules and returns n packed 320-byte records (malloc'ed; release with gs_ply_free).  sh_degree may be NULL. */
int32_t gs_ply_load(const char* path, void** records, uint64_t* n, int32_t* sh_degree);
void gs_ply_free(void* records);
/* gs_ply_load + gs_upload_splats. */
int32_t gs_upload_ply(gs_ctx* ctx, const char* path, uint64_t* n);
/* Same, from a device pointer (no PCIe copy).  The records must be COMPLETE when the call is made: the repack runs on the
 * context's stream and is not ordered against whatever stream produced them (synchronise the producer first).  Returns
 * after the repack; the caller may free d_aos320 then. */
int32_t gs_upload_splats_device(gs_ctx* ctx, const void* d_aos320, uint64_t n);
/* `ctx` renders `owner`'s resident splats (same device; read-only during a frame) with its own stream and per-frame
 * buffers: several contexts rendered round-robin keep several frames in flight, so one frame's blend (instruction-issue
 * bound) overlaps the next frame's binning and sort (memory/latency bound).  The reference has one frame in flight
 * (Renderer.animate awaits every stage, renderer.ts:394-587).  `owner` must outlive `ctx` and must not re-upload meanwhile. */
int32_t gs_share_splats(gs_ctx* ctx, gs_ctx* owner);

/* Replaces one Renderer.animate() frame (renderer.ts:349-593): enqueues the whole frame for the
 * 160-byte uniform block and returns without waiting for the device. */
int32_t gs_render(gs_ctx* ctx, const void* uniforms160);
/* As gs_render, additionally keeping the unsorted (key,value) arrays for GS_BUF_*_UNSORTED. */
int32_t gs_render_debug(gs_ctx* ctx, const void* uniforms160);
/* As gs_render, writing the rgba8 slab image straight into caller-owned DEVICE memory
 * (u8[height][slab_width][4]); used to render into a collective's send buffer. */
int32_t gs_render_to(gs_ctx* ctx, const void* uniforms160, void* d_rgba8);
/* Blocks until the frame is complete (the reference awaits onSubmittedWorkDone 8x per frame).
 * Reports device-side faults; grows the (key,value) capacity and re-renders if the frame overflowed.  When several
 * frames were enqueued since the last gs_wait and an EARLIER one overflowed, that frame cannot be re-rendered: the
 * capacity is grown for the following frames and GS_ERR_TRUNCATED is returned (the last frame is complete). */
int32_t gs_wait(gs_ctx* ctx);

/* Pipelined presentation (a host that does not await every frame, unlike renderer.ts:394-587): as gs_render, plus an asynchronous
 * copy of the finished rgba8 slab image (height * slab_width * 4 bytes, <= size) into `host_dst` on the frame's own stream, so
 * frame k+1 is enqueued -- and rendered by the next member of the ring -- while frame k is still being blended and copied.
 * `host_dst` should come from gs_host_alloc (page-locked: the copy then overlaps the rendering; pageable memory also works, slower)
 * and must stay untouched until gs_wait_ticket(*ticket) has returned.  Tickets count up from 1 per root ctx. */
int32_t gs_render_host(gs_ctx* ctx, const void* uniforms160, void* host_dst, uint64_t size, uint64_t* ticket);
/* Blocks until the frame of that ticket and its copy are complete.  May be called from another thread than the one that
 * enqueues (one waiter per ticket).  A frame that overflowed a capacity is NOT re-rendered here: size the capacities first (render
 * the scene's largest views once with gs_render + gs_wait, or pass gs_config.max_intersections); the next gs_wait reports it
 * (GS_ERR_TRUNCATED).  At most 64 tickets may be outstanding. */
int32_t gs_wait_ticket(gs_ctx* ctx, uint64_t ticket);
/* Page-locked host memory for frame sinks. */
int32_t gs_host_alloc(uint64_t bytes, void** out);
void gs_host_free(void* p);

/* Replaces the blit to the canvas (render.wgsl, renderer.ts:549-574): copies the finished rgba8
 * image of this ctx's slab to host memory; size must be height*slab_width*4. */
int32_t gs_read_rgba8(gs_ctx* ctx, void* dst, uint64_t size);
/* Copies one debug tap to host memory.  *written receives the byte count; dst may be NULL to query it. */
int32_t gs_read_buffer(gs_ctx* ctx, int32_t which, void* dst, uint64_t size, uint64_t* written);
/* Device address of a tap (valid until the next gs_render / gs_destroy), for zero-copy consumers. */
int32_t gs_device_ptr(gs_ctx* ctx, int32_t which, void** d_ptr);
int32_t gs_get_stats(gs_ctx* ctx, gs_stats* out);
/* Tuning / profiling knobs. */
#define GS_OPT_BLEND_ABLATION 1  /* bit 3 (8): the workgroup-per-tile blend kernel at tiles 16 and 32 (identical results; default = one
                                    wave per 8x8 pixel block); bit 2 (4): the default kernel without its two parking culls (live box,
                                    transmittance bound: identical results, slower -- the test that they change no bit); bits 8-15: tile-column strip width of the default kernel's XCD mapping
                                    (0 = automatic).  The remaining bits act only in the PROFILING build (csrc/build.py --profiling,
                                    libgsplat_hip_prof.so; the product library ignores them): bits 0/1 break the image (1 skip the
                                    pixel loop, 2 gather from a cache-resident window), bit 5 counts the evaluations' footprint
                                    (tools/blend_footprint.py), bits 6/7 cap the kernel at 2 / 4 waves per SIMD, bit 16 records a
                                    start / end stamp per walker (GS_BUF 11, tools/blend_profile.py)                              */
#define GS_OPT_PERSISTENT_GRID 2 /* workgroups of the ticket-loop kernels (default 4 per CU)                          */
#define GS_OPT_RESET_TIMING 3    /* start a new averaging window for gs_stats.stage_us_mean                           */
#define GS_OPT_EMIT_ORDER 4      /* 1: the reference's gaussian-index emission order + sort by the full key (3-4 radix digits);
                                    0: depth-ordered pipeline: sort the visible GAUSSIANS by depth bucket, emit their instances
                                    in that order (work-balanced), sort the instances by tile only (2 digits);
                                    2 (default): choose per frame from the previous frame's instance count ((radix sweeps saved)
                                    x instances >= 1 M -> 0, else 1).  Sorted keys/values, ranges and image are identical in
                                    every mode.                                                                           */
#define GS_OPT_DEBUG_VIEW 6      /* the developer views commented out in compute_tiles.wgsl:35-38,67-70, drawn over the frame: 0 off,
                                    1 tile borders (last row/column of every tile red), 2 list length/1000 as grey, 3 pixel
                                    position gradient, 4 list length/100 in red+green                                    */
#define GS_OPT_UNFUSED 5         /* 1 (default): projection, scan and emission are three launches; 0: experimental single fused launch
                                    (identical results; measured slower in round 1)                                              */
#define GS_OPT_TILE_CULL 7       /* 1 (default): gs_render / gs_render_to bin TIGHTLY: an instance (gaussian, tile) is emitted only if
                                    some pixel of the tile can reach alpha >= 1/255 (conservative test, so no output bit changes:
                                    compute_tiles.wgsl:60-63 skips such an entry on every pixel), and carries a mask of the 8x8 pixel
                                    blocks it can touch.  GS_BUF_KEYS / VALUES / RANGES / TILE_COUNTS of such a frame describe that
                                    subset.  0: the reference's binning (every tile of the 3-sigma rect, process_gaussians.wgsl:74-86)
                                    as gs_render_debug always uses.                                                              */
#define GS_OPT_FRAMES_IN_FLIGHT 8 /* 1..4, default 3 for a whole-canvas ctx on its own stream (1 for slabs and caller-supplied streams).
                                    Renderer.animate awaits every frame (renderer.ts:404-587), so it never has two in flight; a host
                                    that enqueues frame k+1 before waiting for frame k gets it rendered by a shadow of the ctx (own
                                    stream and per-frame arrays, the same resident splats; created on first need), gs_render
                                    taking turns between them: frame k's blend overlaps frame k+1's binning.  Every frame's result
                                    is what a single context renders; gs_wait waits for all; read-backs, taps and statistics refer
                                    to the LAST frame.  1 = strictly one frame after the other.                                  */
#define GS_OPT_FRAME_GRAPH 9      /* 1: the frame's commands (1 memset, 12-14 kernels, 2 copies) are captured into a hipGraph the first time
                                    and replayed with one hipGraphLaunch afterwards; only the projection's uniforms change between
                                    frames (a kernel-node parameter update).  For hosts bound by launch cost: small scenes (config A:
                                    a frame is ~70 us of launches) and the narrow slabs of an 8-GPU run.  Frames with per-stage
                                    events (GS_FLAG_TIMING), gs_render_debug and the profiler taps are issued directly as before;
                                    the capture is redone when a buffer moves (capacity growth), an option or the emission order
                                    changes.  Renderer.animate re-encodes every pass every frame (renderer.ts:394-587).  Default 0. */
#define GS_OPT_PROJ_CHUNKS 10     /* tuning: 512-gaussian cull chunks per workgroup of the tight projection (2, 4 or 8; 0 = automatic)  */
int32_t gs_set_option(gs_ctx* ctx, int32_t key, int64_t value);
/* Width in pixels of this ctx's slab (= width when the ctx owns the whole screen). */
int32_t gs_slab_width(gs_ctx* ctx, uint32_t* px_begin, uint32_t* px_width);

/* Multi-GPU presentation: scatter `n_slabs` gathered slab images (slab g = u8[height][w_g][4]) into one
 * row-major u8[height][width][4] DEVICE image.  col_bounds (host) has n_slabs+1 tile-column boundaries.
 * slab_stride_bytes = distance between consecutive slabs in d_slabs (what an all-gather of equally
 * sized, padded send buffers produces); 0 = slabs tightly packed in rank order.  Runs on the ctx stream. */
int32_t gs_assemble_slabs(gs_ctx* ctx, const void* d_slabs, const uint32_t* col_bounds, uint32_t n_slabs,
                          uint64_t slab_stride_bytes, void* d_image);

/* ---- stand-alone stages, mirroring the reference's reusable classes ---------------------------- */
/* GPUSorter.sort (radix_sort/sort.ts:341-350): stable ascending sort of n u32 keys with u32 payloads,
 * host buffers, in place.  values may be NULL (keys only, as testSort does: radix_sort/utils.ts:55-81). */
int32_t gs_sort_pairs_u32(int32_t device, uint32_t* keys, uint32_t* values, uint64_t n, uint32_t key_bits);
/* ExclusiveScanner.scan (exclusive_scan.ts:208-325): in-place exclusive scan of n u32 (each < 2^22, the
 * sum < 2^32), returns the total. */
int32_t gs_exclusive_scan_u32(int32_t device, uint32_t* data, uint64_t n, uint64_t* total);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_GS_ABI_H */
