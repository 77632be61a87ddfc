// gs_device.h -- shared device-side definitions for the gfx950 kernels.
//
// Canonical float semantics (DESIGN.md "Bit-exactness"): the kernels that feed integer outputs
// (rects, tile counts, sort keys) evaluate every WGSL expression as written, left to right, one
// IEEE binary32 rounding per operation.  The whole library is compiled with -ffp-contract=off,
// so a*b+c is two roundings unless it is written as __builtin_fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GS_WAVE 64

// ---- frame constants, passed by value to every kernel -------------------------------------------
struct GsFrame {
    uint32_t n;          // gaussians
    uint32_t width, height, tile_size;
    uint32_t ntx, nty;   // ceil(f32(W)/f32(ts)) (process_gaussians.wgsl:79)
    uint32_t col0, col1; // tile-column slab [col0,col1)
    uint32_t px0, slab_w;// first pixel column and pixel width of the slab
    uint32_t capacity;   // entries the (key,value) arrays can hold
    uint32_t full;       // col0==0 && col1==ntx
    uint32_t row_cap;    // row-item slots the arena / the row-sorted array can hold (a multiple of 16: tight row pipeline)
};

// ---- device-resident control block (zeroed by one memset per frame) -----------------------------
// Every word another workgroup polls lives here or in the status arrays that follow it.
// tile_counts[] words: tile count in the low 22 bits, depth bucket u32(min(50*depth,999)) in the high 10
#define GS_COUNT_BITS 22
#define GS_COUNT_MASK 0x3FFFFFu

struct GsControl {
    uint32_t scan_ticket[2];  // dynamic block ids of the tile-count scan (index-order pipeline)
    uint32_t sort_ticket[4];  // dynamic tile ids, one per radix pass of the instance sort
    uint32_t rows_ticket;     // ... of the row sort (k_rows.hip)
    uint32_t num_slots;       // row-item slots of the frame in depth order (written by the gaussian-level sort)
    uint32_t num_items;       // row items of the frame (slots that are not holes; written by the expansion)
    uint32_t fault;           // set when a bounded spin gives up
    uint32_t overflow;        // set when I exceeds capacity, or the row items the arena
    uint32_t num_intersections; // I
    uint32_t num_visible;
    uint32_t pad0;
    uint32_t row_cursor[16];  // tight projection: slots handed out of each of the arena's 16 shards
    unsigned long long num_processed[64]; // blend: staged list entries (64 partial sums)
    unsigned long long num_evaluated[64]; // blend: (wave, entry) pairs that survived the 8x8 cull
    uint32_t hist[4][256];    // instance sort: digit histograms -> exclusive digit bases
    uint32_t rowhist[8][256]; // row sort: items per tile row, accumulated by the tight projection in 8 copies (workgroup % 8:
                              // one word would take every workgroup's atomic); readers add the copies up (gs_rowhist)
};
__device__ __forceinline__ uint32_t gs_rowhist(const GsControl* ctl, uint32_t r) {
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += ctl->rowhist[k][r];
    return s;
}

// What gs_wait needs to know about a frame, written by ONE thread of the frame's last binning kernel straight into host-mapped
// (page-locked) memory: no device-to-host copy is enqueued per frame (round 2: two copy kernels of ~4.7 us each behind every
// blend).  sticky: words that survive the per-frame memset -- every frame folds its flags into them, so the report of the last
// frame of a batch also tells about the frames enqueued before it ([0] frames that overflowed a capacity, [1] a bounded spin
// gave up, [2] largest instance count, [3] largest arena demand in slots).
struct GsReport {
    uint32_t fault, overflow, num_intersections, num_visible, num_slots, num_items, pad0, pad1;
    uint32_t row_cursor[16];
    uint32_t sticky[4];
};
__device__ __forceinline__ void gs_frame_report(const GsControl* ctl, uint32_t I, uint32_t num_items, uint32_t capacity, uint32_t* sticky,
                                                GsReport* rep) {
    uint32_t need = 0; // arena slots this frame would have needed: 16 shards as large as the fullest one
    for (int k = 0; k < 16; ++k) need = ctl->row_cursor[k] > need ? ctl->row_cursor[k] : need;
    need = need > 0x07FFFFFFu ? 0x7FFFFFFFu : need * 16u;
    const uint32_t over = (I > capacity || ctl->overflow) ? 1u : 0u;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (sticky) {
        s0 = atomicAdd(&sticky[0], over) + over;
        s1 = atomicOr(&sticky[1], ctl->fault ? 1u : 0u) | (ctl->fault ? 1u : 0u);
        s2 = atomicMax(&sticky[2], I); s2 = s2 > I ? s2 : I;
        s3 = atomicMax(&sticky[3], need); s3 = s3 > need ? s3 : need;
    }
    if (rep) {
        rep->fault = ctl->fault; rep->overflow = over; rep->num_intersections = I; rep->num_visible = ctl->num_visible;
        rep->num_slots = ctl->num_slots; rep->num_items = num_items;
        for (int k = 0; k < 16; ++k) rep->row_cursor[k] = ctl->row_cursor[k];
        rep->sticky[0] = s0; rep->sticky[1] = s1; rep->sticky[2] = s2; rep->sticky[3] = s3;
    }
}

struct GsTightOut { // product-path outputs of the tight projection beside GaussianData and the count words (k_preprocess.hip)
    uint32_t* arena; uint32_t* rowptr; GsControl* ctl;
};

struct GsScene {
    const float* px; const float* py; const float* pz; // f32[N] each: all the cull reads
    const float* smax; // f32[N]: largest log-scale of the gaussian (only read by tile-column slabs: conservative radius)
    const float4* geo; // 32 bytes per gaussian: [0] log-scale xyz, opacity logit   [1] rot r,x,y,z
    const float4* sh;  // 192 bytes per gaussian (three 64-byte sectors of its own): 48 SH floats, coefficient-major RGB
                       // Two arrays, not one 256-byte record: the covariance phase reads 32 bytes, the colour phase -- long
                       // after it, behind the tight row counting -- 192; in one record the first 128-byte line was fetched by
                       // both (round 2: 1.19 GB of traffic for 0.78 GB of algorithmic bytes, profiles/README.md)
};

struct GsPlyTable { // where the 11 + 48 values of a packed record live in a raw .ply vertex (gs_upload_ply)
    uint32_t stride, nsrc, all_float;
    uint16_t soff[11 + 48];
    uint8_t stype[11 + 48], slot[11 + 48];
};

struct GsUniforms { // 160 B, renderer.ts:15-24
    float view[16];
    float proj[16];
    float cam[3];
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
};

// ---- canonical scalar helpers (same definitions as the oracle, written independently) -----------
__device__ __forceinline__ float wg_max(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float wg_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ int wg_maxi(int a, int b) { return (a < b) ? b : a; }
__device__ __forceinline__ int wg_mini(int a, int b) { return (b < a) ? b : a; }

// f32 -> i32: truncate, saturate, NaN -> 0
__device__ __forceinline__ int f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int)x;
}
__device__ __forceinline__ uint32_t f2u_sat(float x) {
    if (x != x) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    if (x <= 0.0f) return 0u;
    return (uint32_t)x;
}

// Canonical exp: Cody-Waite reduction by ln2 (hi/lo), degree-5 polynomial, two-step power-of-two
// scaling.  Only IEEE fma/mul/add/rint, so it is bit-identical to the CPU oracle's exp.
__device__ __forceinline__ float gs_exp(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return __builtin_inff();
    if (x < -103.97208404541015625f) return 0.0f;
    const float nf = __builtin_rintf(x * 1.44269502162933349609375f);
    float r = __builtin_fmaf(-nf, 0.693145751953125f, x);
    r = __builtin_fmaf(-nf, 1.42860677465796470642e-06f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float z = r * r;
    float y = __builtin_fmaf(p, z, r);
    y = y + 1.0f;
    const int n = (int)nf;
    const int a = n >> 1;
    const int b = n - a;
    const float sa = __uint_as_float((uint32_t)(a + 127) << 23);
    const float sb = __uint_as_float((uint32_t)(b + 127) << 23);
    return (y * sa) * sb;
}

// ---- wave64 helpers -------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Scans run on DPP row shifts and row broadcasts (VALU latency, no trip through the LDS crossbar as
// ds_bpermute would make): shr 1,2,4,8 inside each row of 16, then lane 15 -> row 1 and 3, lane 31 -> rows 2,3.
// A lane whose source is out of range or masked off reads `old` = the identity.
#define GS_DPP(v, ctrl, rm) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rm), 0xf, false)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t) {
    v += GS_DPP(v, 0x111, 0xf);
    v += GS_DPP(v, 0x112, 0xf);
    v += GS_DPP(v, 0x114, 0xf);
    v += GS_DPP(v, 0x118, 0xf);
    v += GS_DPP(v, 0x142, 0xa);
    v += GS_DPP(v, 0x143, 0xc);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v) { // unsigned: the identity is 0
    uint32_t t;
    t = GS_DPP(v, 0x111, 0xf); v = (t > v) ? t : v;
    t = GS_DPP(v, 0x112, 0xf); v = (t > v) ? t : v;
    t = GS_DPP(v, 0x114, 0xf); v = (t > v) ? t : v;
    t = GS_DPP(v, 0x118, 0xf); v = (t > v) ? t : v;
    t = GS_DPP(v, 0x142, 0xa); v = (t > v) ? t : v;
    t = GS_DPP(v, 0x143, 0xc); v = (t > v) ? t : v;
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0u), 63);
}

// ---- inter-workgroup words (cdna_hip_programming.md Guideline 16, form R2: the data is the flag) --
// A status word is one naturally aligned 4- or 8-byte granule written by ONE relaxed agent-scope
// atomic store (sc1, bypasses L1/keeps no stale copy) and polled with relaxed agent-scope loads.
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent64(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define GS_SPIN_LIMIT (1u << 22) // bounded spins: give up, raise GsControl::fault, never hang the GPU
