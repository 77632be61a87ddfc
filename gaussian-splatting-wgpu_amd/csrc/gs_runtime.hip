// gs_runtime.hip -- context, device arena, frame sequencing and the C ABI (include/gsplat/gs_abi.h).
//
// Replaces the frame orchestration of Renderer (reference src/renderer.ts:96-347 setup/teardown,
// :349-593 animate) and the host halves of ExclusiveScanner (src/exclusive_scan.ts:208-325) and
// GPUSorter (src/radix_sort/sort.ts:249-350).  Where the reference blocks on the queue 8 times per
// frame, reads I back to the CPU, allocates six sort buffers and clears three buffers per frame,
// this runtime enqueues one frame as ~10 kernels + 1 small memset on one HIP stream with no host
// synchronisation: I stays in device memory, grids are persistent (ticket loops), every buffer is
// allocated once (capacity-based) and grown geometrically only when a frame overflows.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/gsplat/gs_abi.h"
#include "gs_kernels.h"
#include "gs_tight.h"

#define GS_EXPORT extern "C" __attribute__((visibility("default")))

static thread_local char g_err[512] = "";
static int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(e_ == hipErrorOutOfMemory ? GS_ERR_OUT_OF_MEMORY : GS_ERR_HIP, "%s: %s", #expr,      \
                        hipGetErrorString(e_));                                                             \
    } while (0)

// roctx ranges named after the reference's stages (renderer.ts:406,421,467,477,501,546), opened around the launches of each
// stage when GS_FLAG_TIMING is set, so a rocprofv3 --marker-trace lines up with the reference's own console.log timings.
// The library is looked up at run time: no link dependency, silently absent when the profiler SDK is not installed.
#include <dlfcn.h>
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        void* h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_LOCAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_LAZY | RTLD_LOCAL);
        if (h) {
            push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (!push || !pop) push = nullptr, pop = nullptr;
        }
    }
};
Roctx& roctx() { static Roctx r; return r; }
const char* const kStageRange[] = {"gsplat:process_gaussians", "gsplat:exclusive_scan", "gsplat:write_tile_ids", "gsplat:radix_sort",
                                   "gsplat:compute_ranges", "gsplat:compute_tiles"};
}

#define GS_EV_RING 256
struct gs_ctx {
    gs_config cfg{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    GsFrame frame{};
    uint32_t T = 0, passes = 0, key_bits = 0; // passes: 8-bit digits of the full key (reference-order pipeline)
    uint32_t tile_passes = 0, tile_bits = 0;  // digits of key/1000 (depth-ordered pipeline)
    bool tile16 = false;                      // every tile id fits 16 bits: the depth-ordered instance sort moves u16 sort words
    bool last_keys16 = false;                 // the last frame's sorted keys are u16 tile ids (GS_BUF_KEYS is rebuilt on demand)
    uint32_t* keysG = nullptr;                // ... into this buffer
    bool keysG_valid = false;
    int emit_order = 2;                       // GS_OPT_EMIT_ORDER: 0 depth-bucket order, 1 gaussian-index order (reference), 2 auto
    bool index_order = true;                  // what the frame being enqueued uses
    uint32_t debug_view = 0;                  // GS_OPT_DEBUG_VIEW
    // Frames in flight (GS_OPT_FRAMES_IN_FLIGHT): when gs_render is called while this context's previous frame is still on the
    // device, the frame goes to a SHADOW context (own stream and per-frame arrays, this context's resident splats), created on
    // first need; gs_render then alternates between them, so one frame's blend (instruction-issue bound) overlaps the next
    // frame's projection / binning / sort (memory and latency bound).  gs_wait waits for all of them; read-backs, taps and
    // statistics refer to the context that rendered the LAST frame.
    uint32_t fif = 1;                         // allowed frames in flight (1 = none of the above)
    std::vector<gs_ctx*> shadows;
    bool is_shadow = false;
    gs_ctx* last = nullptr;                   // who rendered the last frame (this or a shadow); nullptr = this
    uint32_t rr = 0;                          // next slot of the ring {this, shadows...}
    uint64_t cap_hint = 0, row_hint = 0;      // largest capacities any member of the ring has grown to
    // gs_render_host / gs_wait_ticket (root ctx): the event of the last frame + copy of every ring member, who rendered which ticket
    hipEvent_t ev_done = nullptr;
    uint64_t next_ticket = 1;
    struct { uint64_t ticket; gs_ctx* member; } tickets[64] = {};
    std::mutex ticket_mu;
    bool tile_cull = true;                    // GS_OPT_TILE_CULL: tight (opacity-aware) binning in gs_render / gs_render_to
    bool last_tight = false;                  // the last frame used it
    uint32_t grid_persist = 0; // workgroups of the persistent (ticket-loop) kernels
    uint32_t blend_ablation = 0; // profiling only (GS_OPT_BLEND_ABLATION)
    uint32_t* blend_prof = nullptr; // profiling only (ablation bit 16): 4 words per blend walker
    uint32_t blend_prof_blocks = 0;
    // scene planes
    void* scene_mem = nullptr;
    size_t scene_bytes = 0;
    GsScene scene{};
    uint32_t n = 0;
    // per-gaussian frame buffers
    uint32_t* counts = nullptr;
    uint32_t* offsets = nullptr;
    void* gdata = nullptr;
    // (key,value) arrays
    uint64_t capacity = 0;
    uint32_t *keysA = nullptr, *valsA = nullptr, *keysB = nullptr, *valsB = nullptr;
    uint32_t *keysU = nullptr, *valsU = nullptr; // debug copies of the unsorted arrays
    uint32_t* chunk_table = nullptr;             // balanced emission: first gaussian of every EMIT_CHUNK output slots
    uint32_t *keysS = nullptr, *valsS = nullptr; // where the sorted result of the last frame lives
    // control block + look-back status words (one allocation, one memset per frame)
    void* ctl_mem = nullptr;
    size_t ctl_bytes = 0;
    GsControl* ctl = nullptr;
    unsigned long long* scan_status = nullptr;  // [2][scan blocks]
    uint32_t* tile_depth = nullptr;             // blend statistic: deepest staged entry per tile (quadrant kernel)
    uint32_t* sort_status = nullptr;            // instance sort (reference binning)
    uint32_t* rows_status = nullptr;            // row sort (tight row pipeline)
    void* grec = nullptr;                       // visible gaussians in (depth bucket, index) order: {id, count word, prefix, arena address} (k_gsort.hip)
    uint32_t tight_nb = 0;                      // GS_OPT_PROJ_CHUNKS: cull chunks per workgroup of the tight projection (0 = automatic)
    void* gsort_scratch = nullptr;              // (bucket, chunk) table of the gaussian-level counting sort
    // tight row pipeline (k_rows.hip): row items in projection order / sorted by tile row, slot addresses in depth order
    uint64_t row_cap = 0;
    uint32_t *arena = nullptr, *rows_sorted = nullptr, *rowptr = nullptr;
    uint32_t *M3 = nullptr, *tileoff = nullptr, *rowtot = nullptr;
    bool tight_ok = false;                      // the canvas has at most 255 tile rows and columns (8-bit digits of the row pipeline)
    bool scene_borrowed = false;                // gs_share_splats: scene_mem belongs to another context
    uint32_t last_passes = 0;
    bool last_by_index = true;
    uint32_t blend_walkers = 1; // workgroups that walk each tile's list independently in the last frame's blend
    // what record_frame notes about the frame it records (where the sorted result lives, which pipeline ran): a REPLAY of the
    // captured frame restores the capture's notes -- a frame recorded in between (gs_render_debug) would otherwise leave its own
    struct FrameNotes { uint32_t* keysS; uint32_t* valsS; uint32_t last_passes, blend_walkers; bool last_by_index, last_keys16, last_tight; } gnotes{};
    GsControl* h_ctl = nullptr; // pinned copy of the control block's counters, fetched on demand (gs_get_stats)
    bool h_ctl_valid = false;
    GsReport* h_rep = nullptr;    // host-mapped: the frame's last binning kernel writes its report here (gs_device.h); no per-frame copy
    uint32_t* sticky = nullptr;   // device: [0] frames that overflowed, [1] fault, [2] largest I, [3] largest arena demand -- NOT in the
                                  // per-frame memset (gs_frame_report)
    size_t ctl_bytes_tight = 0;   // the part of the control block a tight frame polls: what its memset has to zero
    uint64_t max_I_seen = 0;      // largest instance count since GS_OPT_RESET_TIMING
    uint64_t truncated_frames = 0; // frames that overflowed the capacity and were NOT the frame gs_wait could re-render
    // outputs
    uint32_t* ranges = nullptr;
    uint32_t* rgba8 = nullptr;
    float* rgbf = nullptr;
    uint32_t* d_pxb = nullptr; // assemble: pixel boundaries (device copy of pxb_host)
    uint32_t pxb_host[65] = {};
    uint32_t pxb_n = 0;
    hipEvent_t ev[GS_EV_RING][GS_STAGE_COUNT + 1] = {}; // ring of per-frame stage brackets (GS_FLAG_TIMING)
    bool have_events = false;
    uint64_t timed_from = 0; // first frame index included in the stage means
    // frame graph (GS_OPT_FRAME_GRAPH): the frame's launches captured once, replayed with hipGraphLaunch; only the projection's
    // uniforms change from frame to frame (kernel-node parameter update)
    bool use_graph = false, graph_valid = false;
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    hipGraphNode_t pre_node = nullptr;
    GsPreprocessLaunch pre{};
    bool gkey_index = false, gkey_tight = false;
    void* gkey_ext = nullptr;
    uint64_t graph_frames = 0;
    // frame state
    bool have_frame = false, pending = false, last_debug = false;
    void* last_ext = nullptr;
    GsUniforms last_u{};
    uint64_t frames = 0;
};

static uint32_t tiles_f32(uint32_t extent, uint32_t ts) { // ceil(f32(extent)/f32(ts)), process_gaussians.wgsl:79
    return (uint32_t)std::ceil((float)extent / (float)ts);
}
static uint32_t bits_for(uint64_t v) {
    uint32_t b = 0;
    while (v) { ++b; v >>= 1; }
    return b ? b : 1;
}

GS_EXPORT const char* gs_last_error(void) { return g_err; }
GS_EXPORT int32_t gs_abi_version(void) { return GS_ABI_VERSION; }

static void free_kv(gs_ctx* c) {
    hipFree(c->keysA); hipFree(c->valsA); hipFree(c->keysB); hipFree(c->valsB); hipFree(c->keysU); hipFree(c->valsU);
    hipFree(c->ctl_mem); hipFree(c->chunk_table); hipFree(c->keysG);
    hipFree(c->arena); hipFree(c->rows_sorted); hipFree(c->M3);
    c->keysA = c->valsA = c->keysB = c->valsB = c->keysU = c->valsU = c->keysG = nullptr;
    c->arena = c->rows_sorted = c->M3 = nullptr;
    c->keysG_valid = false;
    c->chunk_table = nullptr;
    c->ctl_mem = nullptr;
}

// (key,value) arrays for `capacity` instances, row-item arrays for `row_cap` slots, and the control/status block sized for both.
static int32_t alloc_kv(gs_ctx* c, uint64_t capacity, uint64_t row_cap) {
    if (capacity >= (1ull << 30)) return fail(GS_ERR_CAPACITY, "capacity %llu exceeds 2^30 intersections", (unsigned long long)capacity);
    if (row_cap >= (1ull << 31)) return fail(GS_ERR_CAPACITY, "%llu row items exceed the 2^31 limit", (unsigned long long)row_cap);
    c->graph_valid = false; // (callers have drained the stream: no replay of the captured frame is in flight)
    free_kv(c);
    capacity = std::max<uint64_t>(capacity, 4096);
    row_cap = (std::max<uint64_t>(row_cap, 4096) + 15) & ~(uint64_t)15;
    const size_t kb = (size_t)capacity * 4;
    HIP_TRY(hipMalloc((void**)&c->keysA, kb));
    HIP_TRY(hipMalloc((void**)&c->valsA, kb));
    HIP_TRY(hipMalloc((void**)&c->keysB, kb));
    HIP_TRY(hipMalloc((void**)&c->valsB, kb));
    HIP_TRY(hipMalloc((void**)&c->chunk_table, (size_t)gs_emit_chunks(std::max(capacity, row_cap)) * 4));
    if (c->tight_ok) {
        HIP_TRY(hipMalloc((void**)&c->arena, (size_t)row_cap * 12));
        HIP_TRY(hipMalloc((void**)&c->rows_sorted, (size_t)row_cap * 12));
        HIP_TRY(hipMalloc((void**)&c->M3, (size_t)gs_rows_chunks(row_cap) * 256 * 4));
    }
    const size_t ctl_sz = (sizeof(GsControl) + 255) & ~(size_t)255;
    const size_t scan_sz = (((size_t)gs_scan_blocks(c->n ? c->n : 1) + 1) * 8 + 255) & ~(size_t)255;
    const size_t sort_sz = (size_t)std::max(c->passes, c->tile_passes) * gs_sort_tiles(capacity) * 256 * 4;
    const size_t rows_sz = c->tight_ok ? (size_t)gs_rows_sort_tiles(row_cap) * 256 * 4 : 0;
    const size_t depth_sz = ((size_t)c->T * 4 + 255) & ~(size_t)255;
    // layout: control block | blend depth | row-sort status || scan status | instance-sort status: a tight frame zeroes the first
    // three only (the instance sort's status alone is 27 MB at config B, and a tight frame never touches it)
    c->ctl_bytes = ctl_sz + depth_sz + rows_sz + scan_sz + sort_sz;
    c->ctl_bytes_tight = ctl_sz + depth_sz + rows_sz;
    HIP_TRY(hipMalloc(&c->ctl_mem, c->ctl_bytes));
    c->ctl = (GsControl*)c->ctl_mem;
    c->tile_depth = (uint32_t*)((char*)c->ctl_mem + ctl_sz);
    c->rows_status = (uint32_t*)((char*)c->ctl_mem + ctl_sz + depth_sz);
    c->scan_status = (unsigned long long*)((char*)c->ctl_mem + ctl_sz + depth_sz + rows_sz);
    c->sort_status = (uint32_t*)((char*)c->ctl_mem + ctl_sz + depth_sz + rows_sz + scan_sz);
    c->capacity = capacity;
    c->row_cap = row_cap;
    c->frame.capacity = (uint32_t)capacity;
    c->frame.row_cap = (uint32_t)row_cap;
    return GS_OK;
}

GS_EXPORT int32_t gs_create(const gs_config* cfg, gs_ctx** out) {
    if (!cfg || !out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_create: null argument");
    if (cfg->struct_size != sizeof(gs_config)) return fail(GS_ERR_INVALID_ARGUMENT, "gs_create: struct_size %u != %zu", cfg->struct_size, sizeof(gs_config));
    if (cfg->width == 0 || cfg->height == 0) return fail(GS_ERR_INVALID_ARGUMENT, "gs_create: empty canvas");
    if (cfg->tile_size != 8 && cfg->tile_size != 16 && cfg->tile_size != 32)
        return fail(GS_ERR_INVALID_ARGUMENT, "gs_create: tile_size must be 8, 16 or 32 (got %u)", cfg->tile_size);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GS_ERR_NO_DEVICE, "gs_create: no HIP device");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(GS_ERR_INVALID_ARGUMENT, "gs_create: device %d of %d", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));

    gs_ctx* c = new (std::nothrow) gs_ctx();
    if (!c) return fail(GS_ERR_OUT_OF_MEMORY, "gs_create: host allocation failed");
    c->cfg = *cfg;
    GsFrame& f = c->frame;
    f.width = cfg->width; f.height = cfg->height; f.tile_size = cfg->tile_size;
    f.ntx = tiles_f32(cfg->width, cfg->tile_size);
    f.nty = tiles_f32(cfg->height, cfg->tile_size);
    if (f.ntx >= 32768 || f.nty >= 65536) { delete c; return fail(GS_ERR_INVALID_ARGUMENT, "gs_create: canvas too large"); }
    f.col0 = cfg->col_begin;
    f.col1 = cfg->col_end;
    if (f.col0 == 0 && f.col1 == 0) f.col1 = f.ntx;
    if (f.col1 > f.ntx || f.col0 >= f.col1) { delete c; return fail(GS_ERR_INVALID_ARGUMENT, "gs_create: bad tile-column slab [%u,%u) of %u", f.col0, f.col1, f.ntx); }
    f.full = (f.col0 == 0 && f.col1 == f.ntx) ? 1u : 0u;
    f.px0 = f.col0 * f.tile_size;
    f.slab_w = std::min(f.width, f.col1 * f.tile_size) - f.px0;
    c->T = f.ntx * f.nty;
    // largest key a rect can produce: tile (nty*ntx + ntx) (rows/cols one past the grid), bucket 999
    const uint64_t max_key = ((uint64_t)f.nty * f.ntx + f.ntx) * 1000ull + 999ull;
    if (max_key > 0xFFFFFFFFull) { delete c; return fail(GS_ERR_INVALID_ARGUMENT, "gs_create: tile ids overflow the 32-bit key (write_tile_ids.wgsl:29)"); }
    c->key_bits = bits_for(max_key);
    c->passes = (c->key_bits + 7) / 8;
    const uint32_t tbits = bits_for((uint64_t)f.nty * f.ntx + f.ntx);
    c->tile_passes = (tbits + 7) / 8;
    c->tile_bits = (tbits + c->tile_passes - 1) / c->tile_passes;
    c->tile16 = ((uint64_t)f.nty * f.ntx + f.ntx) < 0xFFFFull;
    if ((uint64_t)(f.ntx + 1) * (f.nty + 1) > GS_COUNT_MASK) { delete c; return fail(GS_ERR_INVALID_ARGUMENT, "gs_create: canvas has too many tiles"); }
    c->tight_ok = f.ntx <= 255u && f.nty <= 255u; // the row pipeline's digits are a tile row / a tile column: 8 bits each (k_rows.hip)

    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    c->grid_persist = (uint32_t)prop.multiProcessorCount * 4;
    if (cfg->stream) c->stream = (hipStream_t)cfg->stream;
    else { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }

    HIP_TRY(hipMalloc((void**)&c->ranges, (size_t)c->T * 4));
    HIP_TRY(hipMemset(c->ranges, 0, (size_t)c->T * 4));
    const size_t px = (size_t)f.slab_w * f.height;
    HIP_TRY(hipMalloc((void**)&c->rgba8, px * 4));
    HIP_TRY(hipMemset(c->rgba8, 0, px * 4));
    if (cfg->flags & GS_FLAG_F32_TAP) HIP_TRY(hipMalloc((void**)&c->rgbf, px * 12));
    HIP_TRY(hipHostMalloc((void**)&c->h_ctl, sizeof(GsControl), hipHostMallocDefault));
    memset(c->h_ctl, 0, sizeof(GsControl));
    HIP_TRY(hipMalloc((void**)&c->d_pxb, 65 * 4));
    HIP_TRY(hipMalloc((void**)&c->tileoff, 256 * 256 * 4));
    HIP_TRY(hipMalloc((void**)&c->rowtot, 256 * 4));
    HIP_TRY(hipMalloc((void**)&c->sticky, 4 * 4));
    HIP_TRY(hipMemset(c->sticky, 0, 4 * 4));
    HIP_TRY(hipHostMalloc((void**)&c->h_rep, sizeof(GsReport), hipHostMallocDefault));
    memset(c->h_rep, 0, sizeof(GsReport));
    if (cfg->flags & GS_FLAG_TIMING) {
        for (auto& row : c->ev)
            for (auto& e : row) HIP_TRY(hipEventCreate(&e));
        c->have_events = true;
    }
    c->fif = (c->own_stream && f.full) ? 3u : 1u; // a caller-supplied stream or a slab orders its work with the caller's: one frame
                                                  // (3: config B 682 / 787 / 804 / 764 frames/s with 1 / 2 / 3 / 4 in flight)
    *out = c;
    return GS_OK;
}

GS_EXPORT int32_t gs_destroy(gs_ctx* c) {
    if (!c) return GS_OK;
    for (gs_ctx* s : c->shadows) gs_destroy(s);
    c->shadows.clear();
    hipSetDevice(c->cfg.device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->gexec) hipGraphExecDestroy(c->gexec);
    if (c->graph) hipGraphDestroy(c->graph);
    free_kv(c);
    if (!c->scene_borrowed) hipFree(c->scene_mem);
    hipFree(c->counts); hipFree(c->offsets); hipFree(c->gdata);
    hipFree(c->grec); hipFree(c->rowptr); hipFree(c->gsort_scratch);
    hipFree(c->ranges); hipFree(c->rgba8); hipFree(c->rgbf); hipFree(c->d_pxb); hipFree(c->sticky); hipFree(c->blend_prof);
    hipFree(c->tileoff); hipFree(c->rowtot);
    if (c->h_ctl) hipHostFree(c->h_ctl);
    if (c->h_rep) hipHostFree(c->h_rep);
    if (c->have_events)
        for (auto& row : c->ev)
            for (auto& e : row) hipEventDestroy(e);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return GS_OK;
}

static int32_t wait_one(gs_ctx* c);
GS_EXPORT int32_t gs_wait(gs_ctx* c);
// Frees the previous scene and per-gaussian work arrays, allocates the work arrays for n gaussians.
static int32_t alloc_per_gaussian(gs_ctx* c, uint64_t n, uint64_t min_capacity = 0, uint64_t min_rows = 0) {
    c->graph_valid = false;
    if (!c->scene_borrowed) hipFree(c->scene_mem);
    hipFree(c->counts); hipFree(c->offsets); hipFree(c->gdata);
    hipFree(c->grec); hipFree(c->rowptr); hipFree(c->gsort_scratch);
    c->grec = nullptr; c->gsort_scratch = nullptr;
    c->scene_mem = nullptr; c->counts = nullptr; c->offsets = nullptr; c->gdata = nullptr;
    c->rowptr = nullptr;
    c->scene_borrowed = false;
    c->n = (uint32_t)n;
    c->frame.n = (uint32_t)n;
    c->have_frame = false;
    const size_t np = ((size_t)n + 63) & ~(size_t)63;
    HIP_TRY(hipMalloc((void**)&c->counts, std::max<size_t>(np * 4, 256)));
    HIP_TRY(hipMalloc((void**)&c->offsets, std::max<size_t>(np * 4, 256)));
    HIP_TRY(hipMalloc(&c->grec, std::max<size_t>(np * 16, 256)));
    HIP_TRY(hipMalloc((void**)&c->rowptr, std::max<size_t>(np * 4, 256)));
    HIP_TRY(hipMalloc(&c->gsort_scratch, gs_gsort_scratch_bytes((uint32_t)n)));
    HIP_TRY(hipMalloc(&c->gdata, std::max<size_t>((size_t)n * 64, 256)));
    HIP_TRY(hipMemsetAsync(c->gdata, 0, std::max<size_t>((size_t)n * 64, 256), c->stream));
    uint64_t cap = c->cfg.max_intersections ? c->cfg.max_intersections : std::max<uint64_t>(4 * n, 1u << 22);
    cap = std::max<uint64_t>(cap, min_capacity);
    cap = std::min<uint64_t>(cap, (1ull << 30) - 1);
    // row-item slots: a visible gaussian takes one per tile row of its ellipse (about three of them at 1080p, about 40 % of the
    // gaussians visible); grown like the (key,value) capacity when a frame needs more
    const uint64_t rows = std::max<uint64_t>(std::max<uint64_t>(2 * n, 1u << 20), min_rows);
    return alloc_kv(c, cap, rows);
}

// Frees the previous scene, allocates the per-gaussian work arrays and the resident scene arrays for n gaussians.
static int32_t scene_alloc(gs_ctx* c, uint64_t n) {
    int32_t rc = alloc_per_gaussian(c, n);
    if (rc != GS_OK) return rc;
    const size_t np = ((size_t)n + 63) & ~(size_t)63; // plane stride keeps every plane and both record arrays 256-byte aligned
    const size_t bytes = np * 4 * 4 + np * 32 + (size_t)n * 192;
    HIP_TRY(hipMalloc(&c->scene_mem, std::max<size_t>(bytes, 256)));
    c->scene_bytes = bytes;
    char* p = (char*)c->scene_mem;
    GsScene& s = c->scene;
    s.px = (float*)p; p += np * 4; s.py = (float*)p; p += np * 4; s.pz = (float*)p; p += np * 4;
    s.smax = (float*)p; p += np * 4;
    s.geo = (float4*)p; p += np * 32;
    s.sh = (float4*)p;
    return GS_OK;
}

static int32_t upload_common(gs_ctx* c, const void* d_aos, uint64_t n) {
    int32_t rc = scene_alloc(c, n);
    if (rc != GS_OK) return rc;
    if (n) gs_launch_repack(d_aos, (uint32_t)n, c->scene, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GS_OK;
}

// a new scene: the shadows of the ring are dropped (they hold arrays sized for the old one and borrow its planes); the next
// frame that finds its predecessor in flight opens a new one
static int32_t drop_shadows(gs_ctx* c) {
    for (gs_ctx* s : c->shadows) gs_destroy(s);
    c->shadows.clear();
    c->last = c;
    c->rr = 0;
    c->cap_hint = 0;
    c->row_hint = 0;
    return GS_OK;
}
GS_EXPORT int32_t gs_share_splats(gs_ctx* c, gs_ctx* owner) {
    if (!c || !owner || c == owner) return fail(GS_ERR_INVALID_ARGUMENT, "gs_share_splats: needs two distinct contexts");
    if (c->cfg.device != owner->cfg.device) return fail(GS_ERR_INVALID_ARGUMENT, "gs_share_splats: contexts are on different devices");
    if (!c->is_shadow) drop_shadows(c);
    if (!owner->scene_mem) return fail(GS_ERR_NO_SCENE, "gs_share_splats: the owner holds no splats");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (c->pending) { int32_t rc = wait_one(c); if (rc != GS_OK) return rc; }
    // start from the capacity the owner has already grown to: a borrower exists to keep several frames in flight, and a
    // frame that overflows while others are queued behind it cannot be re-rendered (GS_ERR_TRUNCATED)
    int32_t rc = alloc_per_gaussian(c, owner->n, c->cfg.max_intersections ? 0 : owner->capacity, owner->row_cap);
    if (rc != GS_OK) return rc;
    c->scene_mem = owner->scene_mem; // read-only during a frame; the owner must outlive this context
    c->scene_bytes = owner->scene_bytes;
    c->scene = owner->scene;
    c->scene_borrowed = true;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GS_OK;
}

GS_EXPORT int32_t gs_upload_splats_device(gs_ctx* c, const void* d_aos, uint64_t n) {
    if (!c || (!d_aos && n)) return fail(GS_ERR_INVALID_ARGUMENT, "gs_upload_splats_device: null argument");
    if (n >= (1ull << 31)) return fail(GS_ERR_INVALID_ARGUMENT, "gs_upload_splats: too many gaussians");
    HIP_TRY(hipSetDevice(c->cfg.device));
    drop_shadows(c);
    return upload_common(c, d_aos, n);
}

GS_EXPORT int32_t gs_upload_splats(gs_ctx* c, const void* aos, uint64_t n) {
    if (!c || (!aos && n)) return fail(GS_ERR_INVALID_ARGUMENT, "gs_upload_splats: null argument");
    if (n >= (1ull << 31)) return fail(GS_ERR_INVALID_ARGUMENT, "gs_upload_splats: too many gaussians");
    HIP_TRY(hipSetDevice(c->cfg.device));
    drop_shadows(c);
    void* d = nullptr;
    const size_t bytes = (size_t)n * GS_SPLAT_RECORD_BYTES;
    HIP_TRY(hipMalloc(&d, std::max<size_t>(bytes, 256)));
    if (bytes) {
        hipError_t e = hipMemcpy(d, aos, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { hipFree(d); return fail(GS_ERR_HIP, "upload memcpy: %s", hipGetErrorString(e)); }
    }
    int32_t rc = upload_common(c, d, n);
    hipFree(d);
    return rc;
}

static inline void mark(gs_ctx* c, int i) { // stage boundary i: closes stage i-1, opens stage i
    if (!c->have_events) return;
    hipEventRecord(c->ev[c->frames % GS_EV_RING][i], c->stream);
    Roctx& r = roctx();
    if (r.push) {
        if (i > 0) r.pop();
        if (i < GS_STAGE_COUNT) r.push(kStageRange[i]);
    }
}

static void mark_cb(void* p, int i) { mark((gs_ctx*)p, i); }

// Every launch of one frame, in order, on the context's stream (directly, or into a stream capture).
static int32_t record_frame(gs_ctx* c, const GsUniforms& u, bool debug, void* ext_rgba8, bool tight) {
    const GsFrame& f = c->frame;
    hipStream_t st = c->stream;
    gs_launch_zero(c->ctl_mem, tight ? c->ctl_bytes_tight : c->ctl_bytes, st); // (every part is a multiple of 256 bytes)
    c->h_ctl_valid = false;
    if (debug) HIP_TRY(hipMemsetAsync(c->gdata, 0, std::max<size_t>((size_t)c->n * 64, 256), st));
    mark(c, 0);
    gs_preprocess_prepare(c->pre, c->scene, u, f, c->gdata, c->counts, tight, c->arena, c->rowptr, c->ctl, c->tight_nb);
    gs_launch_preprocess(c->pre, st);
    mark(c, 1);
    const bool by_index = !tight && (debug || c->index_order);
    bool keys16 = false;
    if (tight) {
        // The tight row pipeline (k_rows.hip).  Stage brackets: "scan" = the gaussian-level sort by depth bucket, "emit" = the
        // row sort (the row items take write_tile_ids' place), "sort" = count + scan + expansion into the final lists,
        // "ranges" = nothing (they fall out of the scan).
        gs_launch_gsort(c->counts, c->rowptr, c->n, c->gsort_scratch, c->grec, c->chunk_table, (uint32_t)gs_emit_chunks(std::max(c->capacity, c->row_cap)),
                        &c->ctl->num_visible, &c->ctl->num_slots, st);
        mark(c, 2);
        gs_launch_rows(c->arena, c->grec, c->chunk_table, c->rows_sorted, c->ctl, c->rows_status, (uint32_t)c->row_cap, c->M3, c->tileoff, c->rowtot, f, c->valsA,
                       c->ranges, c->grid_persist / 4u, c->sticky, c->h_rep, st, mark_cb, c);
        c->keysS = nullptr;
        c->valsS = c->valsA;
        mark(c, 4);
    } else if (by_index) {
        // the reference's order: scan counts in gaussian order, emit in gaussian order, sort by the full key
        gs_launch_scan(c->counts, c->n, c->offsets, c->scan_status, &c->ctl->scan_ticket[0], c->ctl, st);
        mark(c, 2);
        gs_launch_emit(c->gdata, c->counts, c->offsets, nullptr, nullptr, f, c->keysA, c->valsA, c->ctl, st);
    } else {
        // Depth-ordered emission: the key is tile*1000 + bucket, and the required order inside a tile is (bucket,
        // gaussian index).  Sorting the N_vis visible GAUSSIANS by bucket first (stable, 10 bits, ~16x fewer elements
        // than instances: k_gsort.hip) and emitting their instances in that order leaves only the tile id for the stable
        // instance sort: 2 digits of key/1000 instead of 3 of the key.  The sorted (key,value) arrays are identical.
        gs_launch_gsort(c->counts, nullptr, c->n, c->gsort_scratch, c->grec, c->chunk_table, (uint32_t)gs_emit_chunks(std::max(c->capacity, c->row_cap)),
                        &c->ctl->num_visible, &c->ctl->num_intersections, st);
        mark(c, 2);
        gs_launch_emit_balanced(c->gdata, c->grec, c->chunk_table, f, c->keysA, c->valsA, c->ctl, c->grid_persist * 2,
                                c->tile_bits, c->tile_passes, c->tile16, st);
        keys16 = c->tile16;
    }
    if (debug) {
        if (!c->keysU) {
            HIP_TRY(hipMalloc((void**)&c->keysU, (size_t)c->capacity * 4));
            HIP_TRY(hipMalloc((void**)&c->valsU, (size_t)c->capacity * 4));
        }
        HIP_TRY(hipMemcpyAsync(c->keysU, c->keysA, (size_t)c->capacity * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(c->valsU, c->valsA, (size_t)c->capacity * 4, hipMemcpyDeviceToDevice, st));
    }
    if (!tight) {
        mark(c, 3);
        if (by_index)
            gs_launch_sort(c->keysA, c->valsA, c->keysB, c->valsB, c->ctl, c->ctl->sort_ticket, &c->ctl->hist[0][0], &c->ctl->num_intersections,
                           (uint32_t)c->capacity, c->passes, 8, 0, c->sort_status, c->grid_persist, false, nullptr, nullptr, st, &c->keysS, &c->valsS);
        else
            gs_launch_sort(c->keysA, c->valsA, c->keysB, c->valsB, c->ctl, c->ctl->sort_ticket, &c->ctl->hist[0][0], &c->ctl->num_intersections,
                           (uint32_t)c->capacity, c->tile_passes, c->tile_bits, keys16 ? 0 : 1, c->sort_status, c->grid_persist, /*have_hist=*/true,
                           nullptr, nullptr, st, &c->keysS, &c->valsS, keys16);
        mark(c, 4);
        if (keys16) gs_launch_ranges16((const uint16_t*)c->keysS, c->ctl, (uint32_t)c->capacity, c->T, c->ranges, c->grid_persist * 2, c->sticky, c->h_rep, st);
        else gs_launch_ranges(c->keysS, c->ctl, (uint32_t)c->capacity, c->T, c->ranges, c->grid_persist * 2, c->sticky, c->h_rep, st); // streaming: 8 workgroups/CU
    }
    c->last_passes = tight ? 1u : (by_index ? c->passes : c->tile_passes);
    c->last_by_index = by_index;
    c->last_keys16 = keys16;
    c->last_tight = tight;
    mark(c, 5);
    uint32_t* target = ext_rgba8 ? (uint32_t*)ext_rgba8 : c->rgba8;
    if ((c->blend_ablation & 0x10000u) && !c->blend_prof) HIP_TRY(hipMalloc((void**)&c->blend_prof, (size_t)(1u << 20) * 16));
    const int walkers = gs_launch_blend(c->gdata, c->valsS, c->ranges, f, target, c->rgbf, c->ctl, c->tile_depth, (c->cfg.flags & GS_FLAG_EXACT_BLEND) != 0,
                                        c->blend_ablation & 0xFFFFu, tight, st, (c->blend_ablation & 0x10000u) ? c->blend_prof : nullptr, &c->blend_prof_blocks);
    if (walkers < 0) return fail(GS_ERR_INVALID_ARGUMENT, "unsupported tile size %u", f.tile_size);
    c->blend_walkers = (uint32_t)walkers;
    mark(c, 6);
    if (c->debug_view) gs_launch_debug_view(c->ranges, f, c->debug_view, target, st); // developer views, after the timed stages
    return GS_OK; // (no copy back: the frame's report is in host-mapped memory when the stream has drained, gs_device.h GsReport)
}

static void drop_graph(gs_ctx* c) {
    if (c->gexec) hipGraphExecDestroy(c->gexec);
    if (c->graph) hipGraphDestroy(c->graph);
    c->gexec = nullptr; c->graph = nullptr; c->pre_node = nullptr;
    c->graph_valid = false;
}

static int32_t enqueue_frame(gs_ctx* c, const GsUniforms& u, bool debug, void* ext_rgba8) {
    hipStream_t st = c->stream;
    if (c->emit_order == 2) {
        // auto: the depth-ordered pipeline saves (passes - tile_passes) full sweeps of the instance arrays and the histogram
        // pass, and costs the gaussian-level counting sort (three small kernels, k_gsort.hip).  Measured at config B: whole
        // canvas (18.5 M instances) 1.42 vs 1.50 ms, one of 8 slabs (1.9-2.6 M) 418-508 vs 455-566 us, one of 4 slabs 586-699 vs
        // 618-743 us per frame; below ~1 M instances both are launch-bound and the same.
        const uint64_t saved = c->passes > c->tile_passes ? c->passes - c->tile_passes : 0;
        c->index_order = !(c->have_frame && saved * (uint64_t)c->h_rep->num_intersections >= 1000000ull);
    } else {
        c->index_order = (c->emit_order == 1);
    }
    // tight (opacity-aware) binning: product frames only; the sub-block mask shares the value word with the gaussian id
    const bool tight = !debug && c->tile_cull && c->tight_ok && c->n < (1u << GS_ID_BITS);
    // GS_OPT_FRAME_GRAPH: replay the captured frame instead of issuing its ~16 commands one by one (frames without per-stage
    // events or profiler output only).  The capture holds every buffer address and launch geometry of the frame, so anything
    // that moves a buffer or changes an option drops it (graph_valid); the emission order and the output address are part
    // of its identity.
    const bool graphable = c->use_graph && !debug && !c->have_events && !(c->blend_ablation & 0x10000u) && c->n;
    if (graphable) {
        if (!c->gexec || !c->graph_valid || c->gkey_index != c->index_order || c->gkey_tight != tight || c->gkey_ext != ext_rgba8) {
            if (c->gexec) HIP_TRY(hipStreamSynchronize(st)); // a replay of the old capture may still be running: not destroyed under it
            drop_graph(c);
            HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            const int32_t rc = record_frame(c, u, debug, ext_rgba8, tight);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(st, &g);
            if (rc != GS_OK) { if (g) hipGraphDestroy(g); return rc; }
            if (e != hipSuccess || !g) return fail(GS_ERR_HIP, "frame graph capture: %s", hipGetErrorString(e));
            c->graph = g;
            HIP_TRY(hipGraphInstantiate(&c->gexec, g, nullptr, nullptr, 0));
            size_t nn = 0;
            HIP_TRY(hipGraphGetNodes(g, nullptr, &nn));
            std::vector<hipGraphNode_t> nodes(nn);
            HIP_TRY(hipGraphGetNodes(g, nodes.data(), &nn));
            uint32_t found = 0;
            for (hipGraphNode_t nd : nodes) {
                hipGraphNodeType ty;
                if (hipGraphNodeGetType(nd, &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) continue;
                hipKernelNodeParams kp;
                if (hipGraphKernelNodeGetParams(nd, &kp) == hipSuccess && kp.func == c->pre.func) { c->pre_node = nd; ++found; }
            }
            if (found != 1) { // cannot address the projection's node: this frame still runs from the capture, later ones directly
                c->use_graph = false;
                c->pre_node = nullptr;
            }
            c->gkey_index = c->index_order; c->gkey_tight = tight; c->gkey_ext = ext_rgba8;
            c->gnotes = {c->keysS, c->valsS, c->last_passes, c->blend_walkers, c->last_by_index, c->last_keys16, c->last_tight};
            c->graph_valid = true;
        } else {
            c->keysS = c->gnotes.keysS; c->valsS = c->gnotes.valsS; c->last_passes = c->gnotes.last_passes; c->blend_walkers = c->gnotes.blend_walkers;
            c->last_by_index = c->gnotes.last_by_index; c->last_keys16 = c->gnotes.last_keys16; c->last_tight = c->gnotes.last_tight;
            // (the launch descriptor too: the frame in between prepared it for ITS projection -- other kernel, grid and outputs)
            gs_preprocess_prepare(c->pre, c->scene, u, c->frame, c->gdata, c->counts, tight, c->arena, c->rowptr, c->ctl, c->tight_nb);
            hipKernelNodeParams kp{};
            kp.func = const_cast<void*>(c->pre.func);
            kp.gridDim = dim3(c->pre.blocks); kp.blockDim = dim3(256); kp.sharedMemBytes = 0;
            kp.kernelParams = c->pre.args; kp.extra = nullptr;
            HIP_TRY(hipGraphExecKernelNodeSetParams(c->gexec, c->pre_node, &kp));
        }
        HIP_TRY(hipGraphLaunch(c->gexec, st));
        c->graph_frames++;
    } else {
        const int32_t rc = record_frame(c, u, debug, ext_rgba8, tight);
        if (rc != GS_OK) return rc;
    }
    c->keysG_valid = false;
    HIP_TRY(hipGetLastError());
    c->pending = true;
    c->have_frame = true;
    c->last_debug = debug;
    c->last_ext = ext_rgba8;
    c->last_u = u;
    c->frames++;
    return GS_OK;
}

static int32_t render_common(gs_ctx* c, const void* uniforms, bool debug, void* ext) {
    if (!c || !uniforms) return fail(GS_ERR_INVALID_ARGUMENT, "gs_render: null argument");
    if (!c->scene_mem) return fail(GS_ERR_NO_SCENE, "gs_render: no splats uploaded");
    HIP_TRY(hipSetDevice(c->cfg.device));
    GsUniforms u;
    static_assert(sizeof(GsUniforms) == GS_UNIFORM_BYTES, "uniform block must be 160 bytes");
    memcpy(&u, uniforms, sizeof(u));
    return enqueue_frame(c, u, debug, ext);
}

static int32_t wait_one(gs_ctx* c) {
    if (!c) return fail(GS_ERR_INVALID_ARGUMENT, "gs_wait: null ctx");
    HIP_TRY(hipSetDevice(c->cfg.device));
    uint32_t dropped = 0; // frames enqueued before the last one that overflowed: their output was truncated and is gone
    for (int attempt = 0; attempt < 8; ++attempt) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->pending = false;
        if (!c->have_frame) return GS_OK;
        // the sticky words cover EVERY frame enqueued since the last gs_wait (the control block only the last one)
        const uint32_t over_frames = c->h_rep->sticky[0], fault_any = c->h_rep->sticky[1];
        const uint64_t max_I = c->h_rep->sticky[2], max_rows = c->h_rep->sticky[3];
        if (over_frames || fault_any || max_I || max_rows) HIP_TRY(hipMemset(c->sticky, 0, 4 * 4)); // stream is idle
        c->h_rep->sticky[0] = c->h_rep->sticky[1] = c->h_rep->sticky[2] = c->h_rep->sticky[3] = 0;
        if (max_I > c->max_I_seen) c->max_I_seen = max_I;
        if (fault_any || c->h_rep->fault) return fail(GS_ERR_DEVICE_FAULT, "a look-back spin exceeded its bound (fault word set)");
        const uint64_t I = c->h_rep->num_intersections;
        uint64_t rows_last = 0; // arena slots the last frame asked for: 16 shards as large as its fullest one
        if (c->last_tight)
            for (int k = 0; k < 16; ++k) rows_last = std::max<uint64_t>(rows_last, (uint64_t)c->h_rep->row_cursor[k] * 16);
        const bool last_over = I > c->capacity || c->h_rep->overflow;
        if (attempt == 0 && over_frames > (last_over ? 1u : 0u)) dropped = over_frames - (last_over ? 1u : 0u);
        const uint64_t need = std::max<uint64_t>(I, max_I), need_rows = std::max<uint64_t>(rows_last, max_rows);
        if (!last_over && need <= c->capacity && need_rows <= c->row_cap) break;
        // a frame overflowed the (key,value) capacity or the row-item arena: grow geometrically; re-render the last frame if it was one of them
        if (need >= (1ull << 30)) return fail(GS_ERR_CAPACITY, "%llu intersections exceed the 2^30 limit", (unsigned long long)need);
        uint64_t want = c->capacity, want_rows = c->row_cap;
        if (need > c->capacity) want = std::min<uint64_t>(std::max<uint64_t>(need + need / 4, c->capacity * 2), (1ull << 30) - 1);
        if (need_rows > c->row_cap) want_rows = std::max<uint64_t>(need_rows + need_rows / 4, c->row_cap * 2);
        if (want == c->capacity && want_rows == c->row_cap) { // flagged, yet nothing asks for more: the next attempt would be the same
            if (last_over) return fail(GS_ERR_CAPACITY, "the frame reports an overflow that growing cannot fix (capacity %llu, rows %llu)",
                                       (unsigned long long)c->capacity, (unsigned long long)c->row_cap);
            break;
        }
        hipFree(c->keysU); hipFree(c->valsU); c->keysU = c->valsU = nullptr;
        int32_t rc = alloc_kv(c, want, want_rows);
        if (rc != GS_OK) return rc;
        if (!last_over) break;
        if (attempt == 7) return fail(GS_ERR_CAPACITY, "capacity did not converge");
        c->frames--; // the re-render reuses the frame's slot in the event ring
        rc = enqueue_frame(c, c->last_u, c->last_debug, c->last_ext);
        if (rc != GS_OK) return rc;
    }
    if (dropped) {
        c->truncated_frames += dropped;
        return fail(GS_ERR_TRUNCATED, "%u frame(s) enqueued before the last one overflowed the (key,value) capacity and were rendered from "
                    "truncated lists; the capacity has been grown to %llu -- wait after every frame, or pass gs_config.max_intersections",
                    dropped, (unsigned long long)c->capacity);
    }
    return GS_OK;
}

static gs_ctx* last_of(gs_ctx* c) { return (c && c->last) ? c->last : c; }

GS_EXPORT int32_t gs_wait(gs_ctx* c) {
    if (!c) return fail(GS_ERR_INVALID_ARGUMENT, "gs_wait: null ctx");
    int32_t first = wait_one(c);
    char msg[sizeof(g_err)];
    if (first != GS_OK) memcpy(msg, g_err, sizeof(msg));
    for (gs_ctx* s : c->shadows) {
        const int32_t rc = wait_one(s);
        if (rc != GS_OK && first == GS_OK) { first = rc; memcpy(msg, g_err, sizeof(msg)); }
    }
    c->cap_hint = std::max(c->cap_hint, c->capacity);
    c->row_hint = std::max(c->row_hint, c->row_cap);
    for (gs_ctx* s : c->shadows) { c->cap_hint = std::max(c->cap_hint, s->capacity); c->row_hint = std::max(c->row_hint, s->row_cap); }
    if (first != GS_OK) memcpy(g_err, msg, sizeof(msg));
    return first;
}

static int32_t set_option_one(gs_ctx* c, int32_t key, int64_t value);
// One more member of the ring: a context with this one's configuration that borrows its splats.
static int32_t add_shadow(gs_ctx* c) {
    gs_config cfg = c->cfg;
    cfg.stream = nullptr;
    cfg.max_intersections = 0; // starts from the owner's capacity (gs_share_splats)
    gs_ctx* s = nullptr;
    int32_t rc = gs_create(&cfg, &s);
    if (rc != GS_OK) return rc;
    s->fif = 1;
    s->is_shadow = true;
    rc = gs_share_splats(s, c);
    if (rc != GS_OK) { gs_destroy(s); return rc; }
    s->emit_order = c->emit_order; s->tile_cull = c->tile_cull; s->debug_view = c->debug_view;
    s->blend_ablation = c->blend_ablation; s->grid_persist = c->grid_persist; s->timed_from = 0; s->tight_nb = c->tight_nb;
    s->use_graph = c->use_graph;
    c->shadows.push_back(s);
    return GS_OK;
}

GS_EXPORT int32_t gs_render(gs_ctx* c, const void* uniforms) {
    if (!c || !uniforms) return fail(GS_ERR_INVALID_ARGUMENT, "gs_render: null argument");
    gs_ctx* t = c;
    if (c->fif > 1 && c->scene_mem) {
        // a second frame while the first is still in flight: open the next slot of the ring (once), then take turns
        if (c->pending && c->shadows.size() + 1 < c->fif) {
            int32_t rc = add_shadow(c);
            if (rc != GS_OK) return rc;
            c->rr = (uint32_t)c->shadows.size(); // the new slot takes this frame
        }
        if (!c->shadows.empty()) {
            const uint32_t slot = c->rr++ % (uint32_t)(c->shadows.size() + 1);
            t = slot ? c->shadows[slot - 1] : c;
        }
        if (t->capacity < c->cap_hint || t->row_cap < c->row_hint) { // another member has met a bigger frame: grow before, not after, truncating one
            HIP_TRY(hipSetDevice(t->cfg.device));
            if (t->pending) { int32_t rc = wait_one(t); if (rc != GS_OK && rc != GS_ERR_TRUNCATED) return rc; }
            hipFree(t->keysU); hipFree(t->valsU); t->keysU = t->valsU = nullptr;
            int32_t rc = alloc_kv(t, std::max(t->capacity, c->cap_hint), std::max(t->row_cap, c->row_hint));
            if (rc != GS_OK) return rc;
            t->have_frame = false;
        }
    }
    const int32_t rc = render_common(t, uniforms, false, nullptr);
    if (rc == GS_OK) c->last = t;
    return rc;
}
GS_EXPORT int32_t gs_render_host(gs_ctx* c, const void* uniforms, void* host_dst, uint64_t size, uint64_t* ticket) {
    if (!c || !uniforms || !host_dst || !ticket) return fail(GS_ERR_INVALID_ARGUMENT, "gs_render_host: null argument");
    const uint64_t bytes = (uint64_t)c->frame.slab_w * c->frame.height * 4;
    if (size < bytes) return fail(GS_ERR_INVALID_ARGUMENT, "gs_render_host: need %llu bytes, got %llu", (unsigned long long)bytes, (unsigned long long)size);
    int32_t rc = gs_render(c, uniforms);
    if (rc != GS_OK) return rc;
    gs_ctx* t = last_of(c);
    if (!t->ev_done) HIP_TRY(hipEventCreateWithFlags(&t->ev_done, hipEventDisableTiming));
    HIP_TRY(hipMemcpyAsync(host_dst, t->rgba8, bytes, hipMemcpyDeviceToHost, t->stream));
    std::lock_guard<std::mutex> lk(c->ticket_mu);
    HIP_TRY(hipEventRecord(t->ev_done, t->stream));
    const uint64_t k = c->next_ticket++;
    c->tickets[k & 63u].ticket = k;
    c->tickets[k & 63u].member = t;
    *ticket = k;
    return GS_OK;
}

GS_EXPORT int32_t gs_wait_ticket(gs_ctx* c, uint64_t ticket) {
    if (!c) return fail(GS_ERR_INVALID_ARGUMENT, "gs_wait_ticket: null ctx");
    gs_ctx* t = nullptr;
    {
        std::lock_guard<std::mutex> lk(c->ticket_mu);
        if (ticket == 0 || ticket >= c->next_ticket) return fail(GS_ERR_INVALID_ARGUMENT, "gs_wait_ticket: unknown ticket %llu", (unsigned long long)ticket);
        if (c->next_ticket - ticket > 64 || c->tickets[ticket & 63u].ticket != ticket) return GS_OK; // retired long ago: its member has rendered later frames since
        t = c->tickets[ticket & 63u].member;
    }
    // the member's event is re-recorded by every later frame it renders: waiting on it is waiting for at least this ticket's frame
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipEventSynchronize(t->ev_done));
    return GS_OK;
}

GS_EXPORT int32_t gs_host_alloc(uint64_t bytes, void** out) {
    if (!out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_host_alloc: null argument");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, std::max<size_t>((size_t)bytes, 256), hipHostMallocDefault));
    return GS_OK;
}
GS_EXPORT void gs_host_free(void* p) { if (p) hipHostFree(p); }

GS_EXPORT int32_t gs_render_debug(gs_ctx* c, const void* uniforms) {
    const int32_t rc = render_common(c, uniforms, true, nullptr);
    if (rc == GS_OK && c) c->last = c;
    return rc;
}
GS_EXPORT int32_t gs_render_to(gs_ctx* c, const void* uniforms, void* d_rgba8) {
    if (!d_rgba8) return fail(GS_ERR_INVALID_ARGUMENT, "gs_render_to: null output");
    const int32_t rc = render_common(c, uniforms, false, d_rgba8);
    if (rc == GS_OK && c) c->last = c;
    return rc;
}

GS_EXPORT int32_t gs_slab_width(gs_ctx* c, uint32_t* px_begin, uint32_t* px_width) {
    if (!c) return fail(GS_ERR_INVALID_ARGUMENT, "gs_slab_width: null ctx");
    if (px_begin) *px_begin = c->frame.px0;
    if (px_width) *px_width = c->frame.slab_w;
    return GS_OK;
}

static int32_t tap(gs_ctx* c, int32_t which, void** ptr, uint64_t* bytes) {
    const uint64_t I = std::min<uint64_t>(c->h_rep->num_intersections, c->capacity);
    const uint64_t px = (uint64_t)c->frame.slab_w * c->frame.height;
    switch (which) {
    case GS_BUF_TILE_COUNTS: *ptr = c->counts; *bytes = (uint64_t)c->n * 4; return GS_OK;
    case GS_BUF_TILE_OFFSETS:
        if (!c->last_debug) return fail(GS_ERR_NO_FRAME, "the offsets tap needs gs_render_debug (index-order scan)");
        *ptr = c->offsets; *bytes = (uint64_t)c->n * 4; return GS_OK;
    case GS_BUF_GAUSSIAN_DATA: *ptr = c->gdata; *bytes = (uint64_t)c->n * 64; return GS_OK;
    case GS_BUF_KEYS_UNSORTED:
    case GS_BUF_VALUES_UNSORTED:
        if (!c->last_debug || !c->keysU) return fail(GS_ERR_NO_FRAME, "unsorted taps need gs_render_debug");
        *ptr = which == GS_BUF_KEYS_UNSORTED ? c->keysU : c->valsU; *bytes = I * 4; return GS_OK;
    case GS_BUF_KEYS:
        *bytes = I * 4;
        if (!c->last_keys16 && !c->last_tight) { *ptr = c->keysS; return GS_OK; }
        if (!c->keysG_valid) { // the frame never held full keys (16-bit tile ids, or no keys at all on the tight row pipeline): rebuild tile*1000 + bucket once
            if (c->pending) { int32_t rc = wait_one(c); if (rc != GS_OK) return rc; }
            if (!c->keysG) HIP_TRY(hipMalloc((void**)&c->keysG, (size_t)c->capacity * 4));
            if (c->last_tight) gs_launch_rows_rebuild_keys(c->ranges, c->T, c->valsS, c->counts, (uint32_t)I, c->n, c->keysG, c->stream);
            else gs_launch_rebuild_keys((const uint16_t*)c->keysS, c->valsS, c->counts, (uint32_t)I, c->n, 0xFFFFFFFFu, c->keysG, c->stream);
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->keysG_valid = true;
        }
        *ptr = c->keysG;
        return GS_OK;
    case GS_BUF_VALUES:
    case GS_BUF_BLOCK_MASKS: *ptr = c->valsS; *bytes = I * 4; return GS_OK; // tight frames: id | mask << 28 (gs_read_buffer separates them)
    case GS_BUF_RANGES: *ptr = c->ranges; *bytes = (uint64_t)c->T * 4; return GS_OK;
    case GS_BUF_RGBA8: *ptr = c->last_ext ? c->last_ext : (void*)c->rgba8; *bytes = px * 4; return GS_OK;
    case 12: // TESTS ONLY: the resident scene arrays as one block (position planes, largest log-scale, geometry, SH records)
        *ptr = c->scene_mem; *bytes = c->scene_bytes; return GS_OK;
    case 11: // PROFILING ONLY: per-walker stamps of the blend (GS_OPT_BLEND_ABLATION bit 16)
        if (!c->blend_prof) return fail(GS_ERR_INVALID_ARGUMENT, "no blend profile (GS_OPT_BLEND_ABLATION bit 16)");
        *ptr = c->blend_prof; *bytes = (uint64_t)c->blend_prof_blocks * 16; return GS_OK;
    case GS_BUF_RGB_F32:
        if (!c->rgbf) return fail(GS_ERR_INVALID_ARGUMENT, "GS_BUF_RGB_F32 needs GS_FLAG_F32_TAP");
        *ptr = c->rgbf; *bytes = px * 12; return GS_OK;
    default: return fail(GS_ERR_INVALID_ARGUMENT, "unknown buffer id %d", which);
    }
}

GS_EXPORT int32_t gs_read_buffer(gs_ctx* c, int32_t which, void* dst, uint64_t size, uint64_t* written) {
    if (!c) return fail(GS_ERR_INVALID_ARGUMENT, "gs_read_buffer: null ctx");
    c = last_of(c);
    if (!c->have_frame) return fail(GS_ERR_NO_FRAME, "gs_read_buffer: no frame rendered");
    if (c->pending) { int32_t rc = wait_one(c); if (rc != GS_OK) return rc; }
    HIP_TRY(hipSetDevice(c->cfg.device));
    void* p = nullptr;
    uint64_t bytes = 0;
    if (which == GS_BUF_BLOCK_MASKS && !c->last_tight) { // the reference's binning: the blend tests every block of the tile itself
        bytes = std::min<uint64_t>(c->h_rep->num_intersections, c->capacity) * 4;
        if (written) *written = bytes;
        if (!dst) return GS_OK;
        if (size < bytes) return fail(GS_ERR_INVALID_ARGUMENT, "gs_read_buffer: need %llu bytes, got %llu", (unsigned long long)bytes, (unsigned long long)size);
        const uint32_t nb = (c->frame.tile_size / 8) * (c->frame.tile_size / 8);
        const uint32_t all = nb >= 32 ? 0xFFFFFFFFu : (1u << nb) - 1u;
        for (uint64_t i = 0; i < bytes / 4; ++i) ((uint32_t*)dst)[i] = all;
        return GS_OK;
    }
    if (which == GS_BUF_TILE_COUNTS && c->last_tight) {
        // a tight frame's count words hold row-item slots (k_preprocess.hip): the tile counts of the tap are those of its lists
        bytes = (uint64_t)c->n * 4;
        if (written) *written = bytes;
        if (!dst) return GS_OK;
        if (size < bytes) return fail(GS_ERR_INVALID_ARGUMENT, "gs_read_buffer: need %llu bytes, got %llu", (unsigned long long)bytes, (unsigned long long)size);
        const uint64_t I = std::min<uint64_t>(c->h_rep->num_intersections, c->capacity);
        std::vector<uint32_t> v(I);
        if (I) HIP_TRY(hipMemcpy(v.data(), c->valsS, I * 4, hipMemcpyDeviceToHost));
        uint32_t* w = (uint32_t*)dst;
        memset(w, 0, bytes);
        for (uint64_t i = 0; i < I; ++i) { const uint32_t g = v[i] & GS_ID_MASK; if (g < c->n) w[g]++; }
        return GS_OK;
    }
    int32_t rc = tap(c, which, &p, &bytes);
    if (rc != GS_OK) return rc;
    if (written) *written = bytes;
    if (!dst) return GS_OK;
    if (size < bytes) return fail(GS_ERR_INVALID_ARGUMENT, "gs_read_buffer: need %llu bytes, got %llu", (unsigned long long)bytes, (unsigned long long)size);
    if (bytes) HIP_TRY(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
    if (which == GS_BUF_TILE_COUNTS) { // device words also carry the depth bucket in their high 10 bits
        uint32_t* w = (uint32_t*)dst;
        for (uint64_t i = 0; i < bytes / 4; ++i) w[i] &= GS_COUNT_MASK;
    }
    if (which == GS_BUF_VALUES && c->last_tight) { // strip the sub-block mask
        uint32_t* w = (uint32_t*)dst;
        for (uint64_t i = 0; i < bytes / 4; ++i) w[i] &= GS_ID_MASK;
    }
    if (which == GS_BUF_BLOCK_MASKS) { // tight frame: the mask's sub-blocks (tile/2, or the whole 8-pixel tile) as 8x8-block bits
        uint32_t* w = (uint32_t*)dst;
        const uint32_t ts = c->frame.tile_size;
        for (uint64_t i = 0; i < bytes / 4; ++i) {
            const uint32_t m = w[i] >> GS_ID_BITS;
            uint32_t out = m & (ts == 8 ? 1u : 0xFu);
            if (ts == 32) { // quadrant q = (by/2)*2 + bx/2 of block (bx, by), 4 blocks per row
                out = 0;
                for (uint32_t by = 0; by < 4; ++by)
                    for (uint32_t bx = 0; bx < 4; ++bx)
                        if ((m >> ((by / 2) * 2 + bx / 2)) & 1u) out |= 1u << (by * 4 + bx);
            }
            w[i] = out;
        }
    }
    return GS_OK;
}

GS_EXPORT int32_t gs_read_rgba8(gs_ctx* c, void* dst, uint64_t size) {
    if (!dst) return fail(GS_ERR_INVALID_ARGUMENT, "gs_read_rgba8: null destination");
    return gs_read_buffer(c, GS_BUF_RGBA8, dst, size, nullptr);
}

GS_EXPORT int32_t gs_device_ptr(gs_ctx* c, int32_t which, void** d_ptr) {
    if (!c || !d_ptr) return fail(GS_ERR_INVALID_ARGUMENT, "gs_device_ptr: null argument");
    c = last_of(c);
    uint64_t bytes = 0;
    if (which == GS_BUF_RGBA8) { *d_ptr = c->rgba8; return GS_OK; }
    if (!c->have_frame) return fail(GS_ERR_NO_FRAME, "gs_device_ptr: no frame rendered");
    return tap(c, which, d_ptr, &bytes);
}

GS_EXPORT int32_t gs_get_stats(gs_ctx* root, gs_stats* out) {
    if (!root || !out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_get_stats: null argument");
    gs_ctx* c = last_of(root); // everything below describes the context that rendered the last frame ...
    if (c->pending) { int32_t rc = wait_one(c); if (rc != GS_OK) return rc; }
    memset(out, 0, sizeof(*out));
    out->num_gaussians = c->n;
    out->num_tiles = c->T;
    out->sort_passes = c->last_passes ? c->last_passes : c->passes;
    out->frames = root->frames; // ... except the counters that are sums over the ring
    for (gs_ctx* s : root->shadows) out->frames += s->frames;
    out->depth_ordered = (c->have_frame && !c->last_by_index) ? 1u : 0u;
    if (c->have_frame) {
        if (!c->h_ctl_valid) { // the blend's counters live in the control block: fetched when somebody asks
            HIP_TRY(hipSetDevice(c->cfg.device));
            HIP_TRY(hipMemcpy(c->h_ctl, c->ctl, offsetof(GsControl, hist), hipMemcpyDeviceToHost));
            c->h_ctl_valid = true;
        }
        out->num_visible = c->h_rep->num_visible;
        out->num_intersections = c->h_rep->num_intersections;
        out->capacity = c->capacity;
        out->max_intersections_seen = std::max<uint64_t>(c->max_I_seen, c->h_rep->num_intersections);
        out->truncated_frames = root->truncated_frames;
        for (gs_ctx* s : root->shadows) out->truncated_frames += s->truncated_frames;
        out->frames_in_flight = (uint32_t)root->shadows.size() + 1u;
        out->graph_frames = root->graph_frames;
        for (gs_ctx* s : root->shadows) out->graph_frames += s->graph_frames;
        out->tight_binning = c->last_tight ? 1u : 0u;
        out->row_capacity = c->row_cap;
        if (c->last_tight) {
            out->num_row_items = c->h_rep->num_items;
            out->num_row_slots = c->h_rep->num_slots;
        }
        for (int k = 0; k < 64; ++k) out->num_processed += c->h_ctl->num_processed[k];
        if (c->blend_walkers >= 4) { // 8x8-block walkers (4 per 16-tile, 16 per 32-tile): sum over tiles of the deepest walker
            std::vector<uint32_t> depth(c->T);
            if (hipMemcpy(depth.data(), c->tile_depth, (size_t)c->T * 4, hipMemcpyDeviceToHost) == hipSuccess)
                for (uint32_t v : depth) out->num_processed += v;
        }
        for (int k = 0; k < 64; ++k) out->num_evaluated += c->h_ctl->num_evaluated[k];
        if (c->have_events && c->frames > 0) {
            const uint64_t last = c->frames - 1;
            uint64_t first = c->timed_from;
            if (last + 1 > GS_EV_RING && first < last + 1 - GS_EV_RING) first = last + 1 - GS_EV_RING;
            if (first > last) first = last;
            double sum[GS_STAGE_COUNT + 1] = {0};
            uint32_t cnt = 0;
            for (uint64_t fr = first; fr <= last; ++fr) {
                hipEvent_t* e = c->ev[fr % GS_EV_RING];
                float ms = 0, tot = 0;
                bool ok = true;
                float st[GS_STAGE_COUNT];
                for (int i = 0; i < GS_STAGE_COUNT && ok; ++i) {
                    ok = hipEventElapsedTime(&ms, e[i], e[i + 1]) == hipSuccess;
                    st[i] = ms * 1000.0f;
                }
                ok = ok && hipEventElapsedTime(&tot, e[0], e[GS_STAGE_COUNT]) == hipSuccess;
                if (!ok) continue;
                for (int i = 0; i < GS_STAGE_COUNT; ++i) sum[i] += st[i];
                sum[GS_STAGE_COUNT] += tot * 1000.0f;
                ++cnt;
                if (fr == last) {
                    for (int i = 0; i < GS_STAGE_COUNT; ++i) out->stage_us[i] = st[i];
                    out->frame_us = tot * 1000.0f;
                }
            }
            out->frames_timed = cnt;
            if (cnt) {
                for (int i = 0; i < GS_STAGE_COUNT; ++i) out->stage_us_mean[i] = (float)(sum[i] / cnt);
                out->frame_us_mean = (float)(sum[GS_STAGE_COUNT] / cnt);
            }
        }
    }
    return GS_OK;
}

GS_EXPORT int32_t gs_set_option(gs_ctx* c, int32_t key, int64_t value) {
    if (!c) return fail(GS_ERR_INVALID_ARGUMENT, "gs_set_option: null ctx");
    if (key == GS_OPT_FRAMES_IN_FLIGHT) {
        if (value < 1 || value > 4) return fail(GS_ERR_INVALID_ARGUMENT, "gs_set_option: frames in flight must be 1..4");
        if (!c->own_stream || !c->frame.full) value = 1; // a caller-supplied stream / a slab: see gs_create
        int32_t rc = gs_wait(c);
        if (rc != GS_OK && rc != GS_ERR_TRUNCATED) return rc;
        while (c->shadows.size() + 1 > (size_t)value) { // the last frame may live in a shadow that goes away: taps need a new frame
            if (c->last == c->shadows.back()) { c->last = c; }
            gs_destroy(c->shadows.back());
            c->shadows.pop_back();
        }
        c->fif = (uint32_t)value;
        c->rr = 0;
        return GS_OK;
    }
    int32_t rc = set_option_one(c, key, value);
    for (gs_ctx* s : c->shadows) if (rc == GS_OK) rc = set_option_one(s, key, value);
    return rc;
}
static int32_t set_option_one(gs_ctx* c, int32_t key, int64_t value) {
    c->graph_valid = false; // a captured frame holds the options it was recorded with
    switch (key) {
    case GS_OPT_FRAME_GRAPH: c->use_graph = (value != 0); return GS_OK;
    case GS_OPT_BLEND_ABLATION: c->blend_ablation = (uint32_t)value & 0x3FFFFu; return GS_OK;
    case GS_OPT_PERSISTENT_GRID: if (value <= 0) break; c->grid_persist = (uint32_t)value; return GS_OK;
    case GS_OPT_RESET_TIMING: c->timed_from = c->frames; c->max_I_seen = 0; c->truncated_frames = 0; return GS_OK;
    case GS_OPT_EMIT_ORDER: if (value < 0 || value > 2) break; c->emit_order = (int)value; return GS_OK;
    case GS_OPT_UNFUSED: return GS_OK; // (removed in ABI 3: the fused projection+scan+emission launch measured slower; accepted, ignored)
    case GS_OPT_DEBUG_VIEW: if (value < 0 || value > 4) break; c->debug_view = (uint32_t)value; return GS_OK;
    case GS_OPT_TILE_CULL: c->tile_cull = (value != 0); return GS_OK;
    case GS_OPT_PROJ_CHUNKS: if (value != 0 && value != 2 && value != 4 && value != 8) break; c->tight_nb = (uint32_t)value; return GS_OK;
    default: break;
    }
    return fail(GS_ERR_INVALID_ARGUMENT, "gs_set_option: bad key/value %d/%lld", key, (long long)value);
}

GS_EXPORT int32_t gs_assemble_slabs(gs_ctx* c, const void* d_slabs, const uint32_t* col_bounds, uint32_t n_slabs,
                                    uint64_t slab_stride_bytes, void* d_image) {
    if (!c || !d_slabs || !col_bounds || !d_image || n_slabs == 0 || n_slabs > 64)
        return fail(GS_ERR_INVALID_ARGUMENT, "gs_assemble_slabs: bad argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    uint32_t pxb[65];
    for (uint32_t g = 0; g <= n_slabs; ++g) pxb[g] = std::min(c->frame.width, col_bounds[g] * c->frame.tile_size);
    if (pxb[0] != 0 || pxb[n_slabs] != c->frame.width) return fail(GS_ERR_INVALID_ARGUMENT, "gs_assemble_slabs: bounds must cover the canvas");
    if (c->pxb_n != n_slabs || memcmp(c->pxb_host, pxb, (n_slabs + 1) * 4) != 0) {
        // boundaries change rarely: upload them once (synchronously), not every frame
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(c->d_pxb, pxb, (n_slabs + 1) * 4, hipMemcpyHostToDevice));
        memcpy(c->pxb_host, pxb, (n_slabs + 1) * 4);
        c->pxb_n = n_slabs;
    }
    gs_launch_assemble(d_slabs, d_image, c->frame.width, c->frame.height, c->d_pxb, n_slabs, slab_stride_bytes / 4, c->stream);
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

// ---- native PLY loader ---------------------------------------------------------------
// Restates PackedGaussians (reference src/ply.ts:49-228) in C++: header rules of decodeHeader (:49-102: `element
// vertex N`, properties in file order, data right after "end_header\n"), readRawVertex (:104-123: ONLY float and
// uchar properties consume bytes; uchar is value/255), the SH read order f_dc_{0..2} then f_rest_{rgb*K+i}
// (:179-187), degree from the f_rest count (:168-176), and the packed 320-byte record (:190-198).  Degrees < 3 are
// zero-padded to 16 coefficients (the shader hard-codes 16: process_gaussians.wgsl:6).  The reference spends
// "seconds to a couple of minutes" here in JS (index.html:16); this is a single pass over a read() of the file.
#include <string>
#include <thread>
#include <vector>

struct PlyProp { std::string name; int type; /* 0 other (0 bytes), 1 float, 2 uchar */ uint32_t offset; };

// What decodeHeader (ply.ts:49-102) and PackedGaussians' constructor (:162-228) derive from the header: where the vertex data
// starts, the vertex count and stride, and for each of the 11 + 3 (degree + 1)^2 values of a packed record its byte offset and
// type in a vertex and its float slot in the 320-byte record (ply.ts:190-198).
struct PlyLayout {
    uint64_t data_off = 0, vertex_count = 0;
    uint32_t stride = 0;
    int degree = 0, nsrc = 0;
    uint32_t soff[11 + 48], stype[11 + 48], slot[11 + 48];
    bool all_float = true;
};

// header: the first bytes of the file (at least up to and including "end_header\n")
static int32_t ply_parse_header(const unsigned char* buf, size_t len, PlyLayout& L) {
    static const char kEnd[] = "end_header";
    size_t hdr_end = std::string::npos;
    for (size_t i = 0; i + sizeof(kEnd) - 1 <= len; ++i)
        if (memcmp(buf + i, kEnd, sizeof(kEnd) - 1) == 0) { hdr_end = i; break; }
    if (hdr_end == std::string::npos) return fail(GS_ERR_INVALID_ARGUMENT, "gs_ply_load: no end_header");
    const std::string header((const char*)buf, hdr_end + sizeof(kEnd) - 1);
    L.data_off = hdr_end + sizeof(kEnd) - 1 + 1; // the byte after "end_header" (the newline), ply.ts:94
    std::vector<PlyProp> props;
    size_t pos = 0;
    while (pos < header.size()) {
        size_t eol = header.find('\n', pos);
        if (eol == std::string::npos) eol = header.size();
        std::string line = header.substr(pos, eol - pos);
        pos = eol + 1;
        size_t a = line.find_first_not_of(" \t\r"), b = line.find_last_not_of(" \t\r");
        if (a == std::string::npos) continue;
        line = line.substr(a, b - a + 1);
        if (line.rfind("element vertex", 0) == 0) {
            size_t d = line.find_first_of("0123456789");
            if (d != std::string::npos) L.vertex_count = strtoull(line.c_str() + d, nullptr, 10);
        } else if (line.rfind("property", 0) == 0) {
            char w0[64], w1[64], w2[128];
            if (sscanf(line.c_str(), "%63s %63s %127s", w0, w1, w2) == 3) {
                int type = strcmp(w1, "float") == 0 ? 1 : strcmp(w1, "uchar") == 0 ? 2 : 0;
                bool dup = false;
                for (auto& p : props) if (p.name == w2) { p.type = type; dup = true; }
                if (!dup) props.push_back({w2, type, 0});
            }
        } else if (line == "end_header") {
            break;
        }
    }
    uint32_t stride = 0, n_rest = 0;
    for (auto& p : props) {
        p.offset = stride;
        stride += p.type == 1 ? 4u : p.type == 2 ? 1u : 0u;
        if (p.name.rfind("f_rest_", 0) == 0) ++n_rest;
    }
    L.stride = stride;
    const uint32_t per_color = n_rest / 3;
    int degree = -1;
    for (int d = 0; d <= 3; ++d) if ((uint32_t)((d + 1) * (d + 1) - 1) == per_color && n_rest % 3 == 0) degree = d;
    if (degree < 0) return fail(GS_ERR_INVALID_ARGUMENT, "gs_ply_load: Unsupported SH degree (%u f_rest properties)", n_rest); // ply.ts:136
    L.degree = degree;
    auto find = [&](const std::string& name) -> const PlyProp* {
        for (auto& p : props) if (p.name == name) return &p;
        return nullptr;
    };
    const char* base_names[11] = {"x", "y", "z", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3", "opacity"};
    const int base_slot[11] = {0, 1, 2, 4, 5, 6, 8, 9, 10, 11, 12};
    L.nsrc = 0;
    L.all_float = true;
    auto add = [&](const PlyProp* p, int sl) {
        L.soff[L.nsrc] = p->offset; L.stype[L.nsrc] = (uint32_t)p->type; L.slot[L.nsrc] = (uint32_t)sl; ++L.nsrc;
        L.all_float = L.all_float && p->type == 1;
    };
    for (int i = 0; i < 11; ++i) {
        const PlyProp* p = find(base_names[i]);
        if (!p) return fail(GS_ERR_INVALID_ARGUMENT, "gs_ply_load: missing property %s", base_names[i]);
        add(p, base_slot[i]);
    }
    const int nsh = (degree + 1) * (degree + 1);
    for (int k = 0; k < nsh; ++k)
        for (int c = 0; c < 3; ++c) {
            char nm[32];
            if (k == 0) snprintf(nm, sizeof(nm), "f_dc_%d", c);
            else snprintf(nm, sizeof(nm), "f_rest_%u", (unsigned)(c * per_color + (k - 1)));
            const PlyProp* p = find(nm);
            if (!p) return fail(GS_ERR_INVALID_ARGUMENT, "gs_ply_load: missing property %s", nm);
            add(p, 16 + 4 * k + c);
        }
    return GS_OK;
}

GS_EXPORT int32_t gs_ply_load(const char* path, void** records, uint64_t* n_out, int32_t* sh_degree) {
    if (!path || !records || !n_out) return fail(GS_ERR_INVALID_ARGUMENT, "gs_ply_load: null argument");
    *records = nullptr;
    *n_out = 0;
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(GS_ERR_INVALID_ARGUMENT, "gs_ply_load: cannot open %s", path);
    fseek(fp, 0, SEEK_END);
    const long fsize = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    struct FileBuf { // uninitialised storage: a std::vector would spend a pass zero-filling it
        unsigned char* p = nullptr; size_t n = 0;
        ~FileBuf() { free(p); }
        unsigned char* data() const { return p; }
        size_t size() const { return n; }
    } buf;
    buf.n = (size_t)std::max<long>(fsize, 0);
    buf.p = (unsigned char*)malloc(std::max<size_t>(buf.n, 1));
    if (!buf.p) { fclose(fp); return fail(GS_ERR_OUT_OF_MEMORY, "gs_ply_load: out of host memory"); }
    if (fsize > 0 && fread(buf.p, 1, (size_t)fsize, fp) != (size_t)fsize) { fclose(fp); return fail(GS_ERR_INVALID_ARGUMENT, "gs_ply_load: short read"); }
    fclose(fp);
    PlyLayout L;
    int32_t rc = ply_parse_header(buf.data(), buf.size(), L);
    if (rc != GS_OK) return rc;
    const uint64_t vertex_count = L.vertex_count;
    const uint32_t stride = L.stride;
    if (buf.size() < L.data_off + vertex_count * (uint64_t)stride) return fail(GS_ERR_INVALID_ARGUMENT, "gs_ply_load: vertex data truncated");
    float* out = (float*)calloc((size_t)std::max<uint64_t>(vertex_count, 1) * 80, sizeof(float));
    if (!out) return fail(GS_ERR_OUT_OF_MEMORY, "gs_ply_load: out of host memory");
    // vertices are independent, so the pass is split over threads
    const unsigned char* vbase = buf.data() + L.data_off;
    auto work = [&](uint64_t i0, uint64_t i1) {
        for (uint64_t i = i0; i < i1; ++i) {
            const unsigned char* v = vbase + i * stride;
            float* rec = out + i * 80;
            if (L.all_float) {
                for (int s2 = 0; s2 < L.nsrc; ++s2) memcpy(&rec[L.slot[s2]], v + L.soff[s2], 4);
            } else {
                for (int s2 = 0; s2 < L.nsrc; ++s2) {
                    float f = 0.0f; // a property of another type reads as `undefined` in the reference; 0 here
                    if (L.stype[s2] == 1) memcpy(&f, v + L.soff[s2], 4);
                    else if (L.stype[s2] == 2) f = (float)((double)v[L.soff[s2]] / 255.0);
                    rec[L.slot[s2]] = f;
                }
            }
        }
    };
    unsigned nthreads = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (vertex_count < 65536) nthreads = 1;
    std::vector<std::thread> pool;
    for (unsigned th = 1; th < nthreads; ++th)
        pool.emplace_back(work, vertex_count * th / nthreads, vertex_count * (th + 1) / nthreads);
    work(0, vertex_count / nthreads);
    for (auto& th : pool) th.join();
    *records = out;
    *n_out = vertex_count;
    if (sh_degree) *sh_degree = L.degree;
    return GS_OK;
}

GS_EXPORT void gs_ply_free(void* records) { free(records); }

static int32_t scene_alloc(gs_ctx* c, uint64_t n);
// Streams a .ply straight into the resident scene arrays: the file is read in chunks of 64 Ki vertices into one of
// two pinned staging buffers, copied to the device and converted THERE into the final layout (position planes, geometry and SH
// records: gs_ply_chunk_kernel) while the next chunk is being read.  No N x 320-byte array exists on either side: peak host
// memory is the two staging buffers (2 x 64 Ki x vertex stride, ~32 MB), whatever the size of the scene.  The reference builds
// the whole packed buffer in JS first (ply.ts:204-228: "seconds to a couple of minutes", index.html:16).
GS_EXPORT int32_t gs_upload_ply(gs_ctx* c, const char* path, uint64_t* n_out) {
    if (!c || !path) return fail(GS_ERR_INVALID_ARGUMENT, "gs_upload_ply: null argument");
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(GS_ERR_INVALID_ARGUMENT, "gs_upload_ply: cannot open %s", path);
    std::vector<unsigned char> head(1 << 16);
    const size_t got = fread(head.data(), 1, head.size(), fp);
    PlyLayout L;
    int32_t rc = ply_parse_header(head.data(), got, L);
    if (rc != GS_OK) { fclose(fp); return rc; }
    fseek(fp, 0, SEEK_END);
    const uint64_t fsize = (uint64_t)ftell(fp);
    const uint64_t n = L.vertex_count;
    if (n >= (1ull << 31)) { fclose(fp); return fail(GS_ERR_INVALID_ARGUMENT, "gs_upload_ply: too many gaussians"); }
    if (fsize < L.data_off + n * (uint64_t)L.stride) { fclose(fp); return fail(GS_ERR_INVALID_ARGUMENT, "gs_upload_ply: vertex data truncated"); }
    hipError_t he = hipSetDevice(c->cfg.device);
    if (he != hipSuccess) { fclose(fp); return fail(GS_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(he)); }
    drop_shadows(c);
    rc = scene_alloc(c, n);
    if (rc != GS_OK) { fclose(fp); return rc; }
    if (L.degree < 3 && n) { // coefficients the file does not have stay 0 (the shader hard-codes 16: process_gaussians.wgsl:6)
        hipError_t e0 = hipMemsetAsync((void*)c->scene.sh, 0, (size_t)n * 192, c->stream);
        if (e0 != hipSuccess) { fclose(fp); return fail(GS_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e0)); }
    }
    const uint64_t CH = 65536;
    const size_t chunk_bytes = (size_t)CH * L.stride;
    unsigned char* h_stage[2] = {nullptr, nullptr};
    unsigned char* d_stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    auto cleanup = [&]() {
        for (int k = 0; k < 2; ++k) {
            if (h_stage[k]) hipHostFree(h_stage[k]);
            if (d_stage[k]) hipFree(d_stage[k]);
            if (done[k]) hipEventDestroy(done[k]);
        }
        fclose(fp);
    };
#define PLY_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(e_ == hipErrorOutOfMemory ? GS_ERR_OUT_OF_MEMORY : GS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
    for (int k = 0; k < 2; ++k) {
        PLY_TRY(hipHostMalloc((void**)&h_stage[k], std::max<size_t>(chunk_bytes, 256), hipHostMallocDefault));
        PLY_TRY(hipMalloc((void**)&d_stage[k], std::max<size_t>(chunk_bytes, 256)));
        PLY_TRY(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
    }
    GsPlyTable tab;
    tab.stride = L.stride; tab.nsrc = (uint32_t)L.nsrc; tab.all_float = L.all_float ? 1u : 0u;
    for (int k = 0; k < L.nsrc; ++k) { tab.soff[k] = (uint16_t)L.soff[k]; tab.stype[k] = (uint8_t)L.stype[k]; tab.slot[k] = (uint8_t)L.slot[k]; }
    if (L.stride > 0xFFFFu) { cleanup(); return fail(GS_ERR_INVALID_ARGUMENT, "gs_upload_ply: vertex stride %u too large", L.stride); }
    fseek(fp, (long)L.data_off, SEEK_SET);
    uint32_t it = 0;
    for (uint64_t v0 = 0; v0 < n; v0 += CH, ++it) {
        const int k = (int)(it & 1u);
        const uint64_t m = std::min<uint64_t>(CH, n - v0);
        if (it >= 2) PLY_TRY(hipEventSynchronize(done[k])); // the copy out of this staging buffer two chunks ago has finished
        if (fread(h_stage[k], 1, (size_t)(m * L.stride), fp) != (size_t)(m * L.stride)) { cleanup(); return fail(GS_ERR_INVALID_ARGUMENT, "gs_upload_ply: short read"); }
        PLY_TRY(hipMemcpyAsync(d_stage[k], h_stage[k], (size_t)(m * L.stride), hipMemcpyHostToDevice, c->stream));
        gs_launch_ply_chunk(d_stage[k], (uint32_t)m, (uint32_t)v0, tab, c->scene, c->stream);
        PLY_TRY(hipEventRecord(done[k], c->stream));
    }
    PLY_TRY(hipGetLastError());
    PLY_TRY(hipStreamSynchronize(c->stream));
#undef PLY_TRY
    cleanup();
    if (n_out) *n_out = n;
    return GS_OK;
}

// ---- stand-alone stages ---------------------------------------------------------------------------------
GS_EXPORT int32_t gs_sort_pairs_u32(int32_t device, uint32_t* keys, uint32_t* values, uint64_t n, uint32_t key_bits) {
    if (!keys && n) return fail(GS_ERR_INVALID_ARGUMENT, "gs_sort_pairs_u32: null keys");
    if (n >= (1ull << 30)) return fail(GS_ERR_CAPACITY, "gs_sort_pairs_u32: n >= 2^30");
    if (key_bits == 0 || key_bits > 32) key_bits = 32;
    if (n == 0) return GS_OK;
    HIP_TRY(hipSetDevice(device));
    const uint32_t passes = (key_bits + 7) / 8;
    const size_t kb = (size_t)n * 4;
    const size_t ctl_sz = (sizeof(GsControl) + 255) & ~(size_t)255;
    const size_t st_sz = (size_t)passes * gs_sort_tiles(n) * 256 * 4;
    uint32_t *kA = nullptr, *vA = nullptr, *kB = nullptr, *vB = nullptr;
    void* ctl_mem = nullptr;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    int32_t rc = GS_OK;
    auto cleanup = [&]() { hipFree(kA); hipFree(vA); hipFree(kB); hipFree(vB); hipFree(ctl_mem); };
#define TRY2(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(GS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); cleanup(); return rc; } } while (0)
    TRY2(hipMalloc((void**)&kA, kb)); TRY2(hipMalloc((void**)&vA, kb)); TRY2(hipMalloc((void**)&kB, kb)); TRY2(hipMalloc((void**)&vB, kb));
    TRY2(hipMalloc(&ctl_mem, ctl_sz + st_sz));
    TRY2(hipMemset(ctl_mem, 0, ctl_sz + st_sz));
    TRY2(hipMemcpy(kA, keys, kb, hipMemcpyHostToDevice));
    if (values) TRY2(hipMemcpy(vA, values, kb, hipMemcpyHostToDevice));
    else TRY2(hipMemset(vA, 0, kb));
    GsControl* ctl = (GsControl*)ctl_mem;
    const uint32_t n32 = (uint32_t)n;
    TRY2(hipMemcpy(&ctl->num_intersections, &n32, 4, hipMemcpyHostToDevice));
    uint32_t *ok = nullptr, *ov = nullptr;
    gs_launch_sort(kA, vA, kB, vB, ctl, ctl->sort_ticket, &ctl->hist[0][0], &ctl->num_intersections, n32, passes, 8, 0,
                   (uint32_t*)((char*)ctl_mem + ctl_sz), (uint32_t)prop.multiProcessorCount * 4, false, nullptr, nullptr, nullptr, &ok, &ov);
    TRY2(hipGetLastError());
    TRY2(hipDeviceSynchronize());
    uint32_t fault = 0;
    TRY2(hipMemcpy(&fault, &ctl->fault, 4, hipMemcpyDeviceToHost));
    TRY2(hipMemcpy(keys, ok, kb, hipMemcpyDeviceToHost));
    if (values) TRY2(hipMemcpy(values, ov, kb, hipMemcpyDeviceToHost));
    cleanup();
    if (fault) return fail(GS_ERR_DEVICE_FAULT, "gs_sort_pairs_u32: look-back spin bound exceeded");
    return GS_OK;
}

GS_EXPORT int32_t gs_exclusive_scan_u32(int32_t device, uint32_t* data, uint64_t n, uint64_t* total) {
    if ((!data && n) || !total) return fail(GS_ERR_INVALID_ARGUMENT, "gs_exclusive_scan_u32: null argument");
    if (n >= (1ull << 31)) return fail(GS_ERR_INVALID_ARGUMENT, "gs_exclusive_scan_u32: n too large");
    *total = 0;
    if (n == 0) return GS_OK;
    HIP_TRY(hipSetDevice(device));
    const size_t kb = (size_t)n * 4;
    const size_t ctl_sz = (sizeof(GsControl) + 255) & ~(size_t)255;
    const size_t st_sz = ((size_t)gs_scan_blocks((uint32_t)n) + 1) * 8;
    uint32_t *in = nullptr, *out = nullptr;
    void* ctl_mem = nullptr;
    int32_t rc = GS_OK;
    auto cleanup = [&]() { hipFree(in); hipFree(out); hipFree(ctl_mem); };
    TRY2(hipMalloc((void**)&in, kb)); TRY2(hipMalloc((void**)&out, kb)); TRY2(hipMalloc(&ctl_mem, ctl_sz + st_sz));
    TRY2(hipMemset(ctl_mem, 0, ctl_sz + st_sz));
    TRY2(hipMemcpy(in, data, kb, hipMemcpyHostToDevice));
    GsControl* ctl = (GsControl*)ctl_mem;
    gs_launch_scan(in, (uint32_t)n, out, (unsigned long long*)((char*)ctl_mem + ctl_sz), &ctl->scan_ticket[0], ctl, nullptr);
    TRY2(hipGetLastError());
    TRY2(hipDeviceSynchronize());
    GsControl h;
    TRY2(hipMemcpy(&h, ctl, offsetof(GsControl, hist), hipMemcpyDeviceToHost));
    TRY2(hipMemcpy(data, out, kb, hipMemcpyDeviceToHost));
    cleanup();
    if (h.fault) return fail(GS_ERR_DEVICE_FAULT, "gs_exclusive_scan_u32: look-back spin bound exceeded");
    *total = h.num_intersections;
    return GS_OK;
}
