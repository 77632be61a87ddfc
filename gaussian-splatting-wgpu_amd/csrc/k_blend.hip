// k_blend.hip -- per-tile front-to-back alpha blend and the rgba8unorm store.
//
// Replaces compute_tiles.wgsl::main (reference src/compute_tiles.wgsl:30-75) and the blit
// render.wgsl::vertex_main/fragment_main (src/render.wgsl:14-31), which is the identity at equal
// size: the blended image IS the presented image, so the extra full-frame pass disappears.
//
// The reference re-gathers the 64-byte GaussianData record from global memory for every pixel of
// the tile (256x redundant, compute_tiles.wgsl:49-50; README TODO "Load Gaussian data workgroup
// wide").  Here a workgroup stages a batch of its tile's sorted list ONCE into LDS (40 B/entry:
// centre in pixels, conic, opacity, colour, cull limit), each wave skips the entries that cannot
// touch its 8x8 pixel block, pixels read the surviving entries by LDS broadcast, and the tile stops
// as soon as every pixel is finished under the exact criterion below -- which changes no output
// bit:
//     a later entry can only be kept if alpha >= c (c = f32(1/255)) and T*(1-alpha) >= 1e-4;
//     fl(T*fl(1-alpha)) <= fl(T*fl(1-c)) for every alpha >= c, so once fl(T*fl(1-c)) < 1e-4 the
//     pixel's colour is final.
//
// Two arithmetic modes (gs_abi.h GS_FLAG_EXACT_BLEND):
//   EXACT : the expression tree of the WGSL, one rounding per operation, canonical exp -> bit-equal
//           to the CPU oracle.
//   fast  : same f32 algorithm with fma contraction and the hardware exp2 (v_exp_f32); differs from
//           EXACT by rounding only (<= 1e-4 per channel away from keep/skip thresholds).
// Bound: f32 VALU + transcendental issue (about 22 VALU slots per pixel x entry), not HBM:
// algorithmic bytes are 40 B per staged entry + 4 B per pixel.
#include "gs_device.h"
#include "gs_tight.h"
#include <type_traits>

// Minimum over the pixel block [dxlo,dxhi] x [dylo,dyhi] (offsets g - p) of the quadratic
// q(d) = 0.5*(cx*dx^2 + cz*dy^2) + cy*dx*dy (power = -q).  For a positive-definite conic whose centre
// is outside the block the minimiser lies on an edge facing the centre: at most two 1-D problems.
// (v_rcp_f32 instead of an IEEE division: q is evaluated AT the clamped point, so a 1-ulp error in the
// minimiser only moves q by a second-order amount, far inside the caller's margin.)
__device__ __forceinline__ float block_qmin(float cx, float cy, float cz, float dxlo, float dxhi, float dylo, float dyhi,
                                            float& mag) {
    const float X = __builtin_fminf(__builtin_fmaxf(0.0f, dxlo), dxhi); // clamp(0, lo, hi)
    const float Y = __builtin_fminf(__builtin_fmaxf(0.0f, dylo), dyhi);
    float q = 3.0e38f;
    mag = 0.0f;
    if (X == 0.0f && Y == 0.0f) return 0.0f; // centre inside the block
    if (X != 0.0f) {
        const float dy = __builtin_fminf(__builtin_fmaxf(-cy * X * __builtin_amdgcn_rcpf(cz), dylo), dyhi);
        const float a = 0.5f * cx * X * X, b = 0.5f * cz * dy * dy, c = cy * X * dy;
        q = a + b + c;
        mag = __builtin_fabsf(a) + __builtin_fabsf(b) + __builtin_fabsf(c);
    }
    if (Y != 0.0f) {
        const float dx = __builtin_fminf(__builtin_fmaxf(-cy * Y * __builtin_amdgcn_rcpf(cx), dxlo), dxhi);
        const float a = 0.5f * cx * dx * dx, b = 0.5f * cz * Y * Y, c = cy * dx * Y;
        const float q2 = a + b + c;
        if (q2 < q) { q = q2; mag = __builtin_fabsf(a) + __builtin_fabsf(b) + __builtin_fabsf(c); }
    }
    return q;
}

// The transmittance cull of the blend walkers (see gs_blend_quad_kernel, "... and the transmittance that is left"): true if
// Tmax (1 - alpha_lo) < 1e-4 with margins, alpha_lo a lower bound of the entry's alpha over the box [dxlo,dxhi] x [dylo,dyhi]
// (offsets centre - pixel; the quadratic's maximum over a rectangle is at a corner).  Positive-definite conics only.
__device__ __forceinline__ bool blend_tmax_cull(float cx, float cy, float cz, float op, float dxlo, float dxhi, float dylo, float dyhi, float Tmax) {
    const float ax0 = (0.5f * cx) * dxlo * dxlo, ax1 = (0.5f * cx) * dxhi * dxhi, by0 = (0.5f * cz) * dylo * dylo, by1 = (0.5f * cz) * dyhi * dyhi;
    const float c00 = cy * dxlo * dylo, c01 = cy * dxlo * dyhi, c10 = cy * dxhi * dylo, c11 = cy * dxhi * dyhi;
    const float qmax = __builtin_fmaxf(__builtin_fmaxf(ax0 + by0 + c00, ax0 + by1 + c01), __builtin_fmaxf(ax1 + by0 + c10, ax1 + by1 + c11));
    const float qmag = __builtin_fmaxf(ax0, ax1) + __builtin_fmaxf(by0, by1) +
                       __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(c00), __builtin_fabsf(c01)), __builtin_fmaxf(__builtin_fabsf(c10), __builtin_fabsf(c11)));
    const float alo = 0.99f * __builtin_fminf(0.99f, op * __builtin_amdgcn_exp2f(-1.44269502162933349609375f * (qmax + 1.0e-5f * qmag)));
    return Tmax * (1.0f - alo) < 0.0000999f; // (NaN anywhere: false, the entry stays)
}

// One workgroup per tile, TS*TS threads, one pixel per thread; wave w owns the 8x8 pixel block
// (w % (TS/8), w / (TS/8)) of the tile.  The tile's sorted list is consumed in batches of TS*TS
// entries through a DOUBLE-BUFFERED LDS stage: while a batch is being blended, the next batch's
// (value -> GaussianData) gathers are already in flight in registers, so there is one barrier per
// batch and the gather latency is covered.  Per batch every wave first builds, 64 entries at a time
// (one per lane), the mask of entries that can reach alpha >= 1/255 somewhere in ITS 8x8 block
// (conservative: an entry is dropped only if op*exp(-qmin) < c255 with a margin far above f32
// rounding, so no output bit changes), then walks only the set bits with a branch-free body.
// A finished pixel needs no flag inside the loop: by the exit criterion no later entry can pass
// `T*(1-alpha) >= 1e-4` for it, so `done` is only evaluated between batches to skip whole waves/tiles.
template <int TS, bool EXACT>
__global__ __launch_bounds__(TS* TS) void gs_blend_kernel(const uint4* __restrict__ gdata, const uint32_t* __restrict__ values,
                                                           const uint32_t* __restrict__ ranges, GsFrame f,
                                                           uint32_t* __restrict__ rgba8, float* __restrict__ rgbf, GsControl* ctl,
                                                           uint32_t dbg, uint32_t id_mask) {
    constexpr int NT = TS * TS;
    constexpr int ROUNDS = NT / 64; // 64-entry groups per batch
    constexpr int WPR = TS / 8;     // waves per tile row
    // staged entry e of a batch = three 16-byte pieces, one per 16-byte piece of the GaussianData record
    __shared__ float4 sP0[2][NT]; // gx, gy (pixels), cull limit ln(255*opacity)+margin, -
    __shared__ float4 sP1[2][NT]; // conic.x, conic.y, conic.z (fused mode: pre-scaled), depth
    __shared__ float4 sP2[2][NT]; // r, g, b, opacity

    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t tx = f.col0 + blockIdx.x, ty = blockIdx.y;
    const uint32_t tile = tx + ty * f.ntx;
    const uint32_t start = tile > 0 ? ranges[tile - 1] : 0u;
    uint32_t end = ranges[tile];
    if (end > f.capacity) end = f.capacity;

    const uint32_t bx0 = tx * TS + (w % WPR) * 8, by0 = ty * TS + (w / WPR) * 8;
    const uint32_t gx = bx0 + (lane & 7), gy = by0 + (lane >> 3);
    const float pxf = (float)gx, pyf = (float)gy;
    const float bx0f = (float)bx0, by0f = (float)by0;
    float T = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
    const bool outside = !(gx < f.width && gy < f.height);
    bool done = outside;
    const float c255 = (float)(1.0 / 255.0);
    const float Wf = (float)f.width, Hf = (float)f.height;
    uint32_t staged = 0, evaluated = 0;

    // Gather: a wave fetches the 64 records of its 64 entries with FOUR lanes per 64-byte record
    // (lane = 4*rec + piece), 16 records per load instruction, so the texture-address unit sees 64
    // full lines per batch and wave instead of 192 partial ones.  Each lane converts the piece it
    // holds and writes it straight to its slot of the stage: no transpose, no extra buffer.
    const uint32_t piece = lane & 3u, qrec = lane >> 2;
    uint4 rq[4];
    uint32_t gnext = 0; // value (gaussian id) of entry b + 2*NT + tid, fetched two batches ahead
    auto fetch_ids = [&](uint32_t b) { gnext = (b + tid < end) ? values[b + tid] : 0u; };
    auto fetch = [&](uint32_t b, uint32_t gval) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t rec = qrec + 16u * i; // entry (w*64 + rec) of the batch
            uint32_t g = __shfl(gval, rec, 64) & id_mask; // tight frames: the sub-block mask rides in the high bits (gs_tight.h)
#ifdef GS_PROFILING
            if (dbg & 2u) g &= 1023u; // gather from a cache-resident window
#endif
            if (b + w * 64u + rec < end && piece < 3u) rq[i] = gdata[(uint64_t)g * 4 + piece];
        }
    };
    auto stage = [&](uint32_t b, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t e = w * 64u + qrec + 16u * i;
            if (b + e < end) {
                const float x = __uint_as_float(rq[i].x), y = __uint_as_float(rq[i].y), z = __uint_as_float(rq[i].z),
                            ww = __uint_as_float(rq[i].w);
                if (piece == 0u) {
                    sP0[buf][e].x = x * Wf; // compute_tiles.wgsl:52
                    sP0[buf][e].y = y * Hf;
                } else if (piece == 1u) {
                    // fused mode: fold -0.5 and log2(e) into the conic once per entry, so that
                    // power*log2(e) = hx*dx^2 + hy*dx*dy + hz*dy^2
                    const float L = 1.44269502162933349609375f;
                    sP1[buf][e] = EXACT ? make_float4(x, y, z, ww) : make_float4((-0.5f * L) * x, (-L) * y, (-0.5f * L) * z, ww);
                } else if (piece == 2u) {
                    sP2[buf][e] = make_float4(x, y, z, ww);
                    // alpha >= c255  <=>  q <= ln(255*op); +0.01 keeps the cull conservative (rounding is ~1e-6)
                    sP0[buf][e].z = __builtin_amdgcn_logf(ww * 255.0f) * 0.693147182464599609375f + 0.01f; // v_log_f32 (log2) * ln 2
                }
            }
        }
    };

    uint32_t gcur = 0;
    if (start < end) {
        fetch_ids(start);
        gcur = gnext;
        fetch(start, gcur);
        fetch_ids(start + NT);
        stage(start, 0);
    }
    __syncthreads();
    int buf = 0;
    for (uint32_t b = start; b < end; b += NT, buf ^= 1) {
        const uint32_t nb = b + NT;
        if (nb < end) {
            gcur = gnext;
            fetch(nb, gcur);          // in flight while this batch is blended
            fetch_ids(nb + NT);       // ids two batches ahead: the gather above never waits for them
        }
        const uint32_t cnt = (end - b < (uint32_t)NT) ? end - b : (uint32_t)NT;
        staged += cnt;
        const unsigned long long lv = __ballot(!done);
        if (lv != 0ull) { // otherwise this wave's 8x8 block is final (uniform per wave)
            // the live box and the transmittance left in it (as in gs_blend_quad_kernel: an entry that cannot change a live pixel is skipped)
            uint32_t lcm = (uint32_t)lv | (uint32_t)(lv >> 32);
            lcm |= lcm >> 16; lcm |= lcm >> 8; lcm &= 0xFFu;
            const float lc0 = (float)__builtin_ctz(lcm | 0x100u), lc1 = (float)(31 - __builtin_clz(lcm | 1u));
            const float lr0 = (float)(__builtin_ctzll(lv) >> 3), lr1 = (float)((63 - __builtin_clzll(lv)) >> 3);
            const float Tmax = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)wave_incl_max(done ? 0u : __float_as_uint(T)), 63));
#pragma unroll 1
            for (int r = 0; r < ROUNDS; ++r) {
                const uint32_t e0 = (uint32_t)r * 64u;
                if (e0 >= cnt) break;
                bool rel = false;
                if (e0 + lane < cnt) {
                    const float4 p0 = sP0[buf][e0 + lane];
                    float4 p1 = sP1[buf][e0 + lane];
                    if (!EXACT) { // undo the staging scale (the margin absorbs the extra rounding)
                        const float iL = 0.693147182464599609375f;
                        p1.x *= -2.0f * iL;
                        p1.y *= -iL;
                        p1.z *= -2.0f * iL;
                    }
                    const float dxh = p0.x - bx0f, dyh = p0.y - by0f;
                    const float dxlo = dxh - lc1, dxhi = dxh - lc0, dylo = dyh - lr1, dyhi = dyh - lr0; // the LIVE pixels' box
                    const bool pd = (p1.x > 0.0f) && (p1.z > 0.0f) && (p1.x * p1.z - p1.y * p1.y > 0.0f);
                    float mag;
                    const float q = block_qmin(p1.x, p1.y, p1.z, dxlo, dxhi, dylo, dyhi, mag);
                    rel = !pd || !(q > p0.z + 1.0e-5f * mag); // NaNs compare false -> relevant
                    if (rel && pd && blend_tmax_cull(p1.x, p1.y, p1.z, sP2[buf][e0 + lane].w, dxlo, dxhi, dylo, dyhi, Tmax)) rel = false;
#ifdef GS_PROFILING
                    if (dbg & 1u) rel = false; // staging + cull cost without the pixel loop
#endif
                }
                unsigned long long m = __ballot(rel);
                evaluated += (uint32_t)__popcll(m);
                while (m) {
                    const uint32_t e = e0 + (uint32_t)__builtin_ctzll(m);
                    m &= m - 1ull;
                    const float4 p0 = sP0[buf][e];
                    const float4 p1 = sP1[buf][e];
                    const float4 p2v = sP2[buf][e];
                    const float dx = p0.x - pxf, dy = p0.y - pyf;
                    if (EXACT) {
                        const float t1 = p1.x * dx * dx, t2 = p1.z * dy * dy, t3 = p1.y * dx * dy;
                        const float power = -0.5f * (t1 + t2) - t3;
                        const float alpha = wg_min(0.99f, p2v.w * gs_exp(power));
                        const float test = T * (1.0f - alpha);
                        const float cond = (power <= 0.0f && alpha >= c255 && test >= 0.0001f) ? 1.0f : 0.0f;
                        cr += cond * p2v.x * alpha * T;
                        cg += cond * p2v.y * alpha * T;
                        cb += cond * p2v.z * alpha * T;
                        T = cond * test + (1.0f - cond) * T;
                    } else {
                        const float u = __builtin_fmaf(p1.x, dx, p1.y * dy);
                        const float v = (p1.z * dy) * dy;
                        const float pw = __builtin_fmaf(dx, u, v); // power * log2(e)
                        const float alpha = __builtin_fminf(0.99f, p2v.w * __builtin_amdgcn_exp2f(pw));
                        const float test = __builtin_fmaf(-T, alpha, T);
                        const bool keep = (pw <= 0.0f) && (alpha >= c255) && (test >= 0.0001f);
                        const float ak = keep ? alpha : 0.0f;
                        const float wgt = ak * T;
                        cr = __builtin_fmaf(p2v.x, wgt, cr);
                        cg = __builtin_fmaf(p2v.y, wgt, cg);
                        cb = __builtin_fmaf(p2v.z, wgt, cb);
                        T = keep ? test : T;
                    }
                }
            }
            // exit criterion: no later entry can be kept once fl(T*fl(1-c255)) < 1e-4
            if (EXACT) done = outside || (T * (1.0f - c255) < 0.0001f);
            else done = outside || (__builtin_fmaf(-T, c255, T) < 0.0001f);
        }
        if (nb < end) stage(nb, buf ^ 1); // every wave finished reading buf^1 before the previous barrier
        // one barrier per batch: publishes the next stage; stop when every pixel of the tile is final
        if (__syncthreads_and(done ? 1 : 0)) break;
    }
    // statistics: spread over 64 words so that 8 160+ tiles do not serialise on one atomic
    if (tid == 0 && staged) atomicAdd(&ctl->num_processed[(blockIdx.x + blockIdx.y * gridDim.x) & 63u], (unsigned long long)staged);
    if (lane == 0 && evaluated) atomicAdd(&ctl->num_evaluated[(blockIdx.x + blockIdx.y * gridDim.x + w) & 63u], (unsigned long long)evaluated);

    // textureStore(render_target, xy, vec4(C, 1)) to rgba8unorm (compute_tiles.wgsl:71): clamp, *255, round
    if (!outside) {
        const float c[3] = {cr, cg, cb};
        uint32_t q = 0xFF000000u;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float v = c[ch];
            v = (v != v) ? 0.0f : wg_min(wg_max(v, 0.0f), 1.0f);
            q |= (uint32_t)__builtin_floorf(v * 255.0f + 0.5f) << (8 * ch);
        }
        const uint64_t o = (uint64_t)gy * f.slab_w + (gx - f.px0);
        rgba8[o] = q;
        if (rgbf) {
            rgbf[o * 3 + 0] = cr;
            rgbf[o * 3 + 1] = cg;
            rgbf[o * 3 + 2] = cb;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Quadrant variant (the default): one single-wave workgroup per 8x8 pixel block of a 16x16 tile, each walking
// the tile's list on its own: no workgroup barrier anywhere, the eight waves resident on a SIMD are eight
// independent gather -> park -> blend pipelines that hide each other's memory latency; 32 640 short
// pipelines keep the dispatcher refilling the SIMDs until the end of the launch (round 1's one wave per
// whole tile let the occupancy decay as its 8 160 waves retired), and a block stops as soon as ITS 64
// pixels are final.  The four blocks of a tile are workgroups b, b+8, b+16, b+24 (the same XCD under
// round-robin placement) so the three extra gathers of every record are normally L2 hits; placement
// only affects speed.
// ------------------------------------------------------------------------------------------------
// MASKED (tight frames, gs_tight.h): values[] holds gaussian id | sub-block mask << 28.  A walker looks at ITS bit before
// it gathers: entries that cannot touch its pixels cost one id read and nothing else (no record gather, no cull
// arithmetic).  At tile 16 the bit is per 8x8 block, i.e. exactly this walker's pixels, so the closed-form cull below is
// skipped altogether; at tile 32 the bit is per 16x16 quadrant and the cull still runs on what the bit lets through.
#ifdef GS_PROFILING
// PROFILING BUILD ONLY: footprint of the evaluations -- [0] evaluations, [1] lanes with alpha >= 1/255, [2] 4x4 pixel quads (of the
// block's four) holding such a lane, [3] evaluations with none, [4] live lanes (pixel not final), [5] kept lanes (both tests), [6] evaluations
// whose alpha >= 1/255 lanes are all final, [7] evaluations with <= 16 live lanes, [8] entries parked while <= 16 pixels are live, [9] blocks of the tile their masks name, [10] evaluations that keep no lane (out: 11 words); 256 copies 128 bytes apart
__device__ unsigned long long gs_blend_foot[256][16];
extern "C" __attribute__((visibility("default"))) int gs_prof_blend_footprint(unsigned long long* out4, int reset) {
    static unsigned long long h[256][16];
    if (out4) {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(gs_blend_foot), sizeof(h)) != hipSuccess) return -1;
        for (int k = 0; k < 11; ++k) { out4[k] = 0; for (int c = 0; c < 256; ++c) out4[k] += h[c][k]; }
    }
    if (reset) { for (auto& r : h) for (auto& v : r) v = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(gs_blend_foot), h, sizeof(h)) != hipSuccess) return -1; }
    return 0;
}
#endif
typedef uint32_t gs_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t gs_u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t gs_u32x4 __attribute__((ext_vector_type(4)));
template <bool EXACT, int TS = 16, bool MASKED = false>
__global__ __launch_bounds__(64) void gs_blend_quad_kernel(const uint4* __restrict__ gdata, const uint32_t* __restrict__ values,
                                                            const uint32_t* __restrict__ ranges, GsFrame f,
                                                            uint32_t* __restrict__ rgba8, float* __restrict__ rgbf, GsControl* ctl,
                                                            uint32_t* __restrict__ tile_depth, uint32_t dbg, uint32_t* __restrict__ prof) {
    constexpr uint32_t BPR = TS / 8, NB = BPR * BPR;
#ifdef GS_PROFILING // (build.py --profiling; GS_OPT_BLEND_ABLATION bit 16): start / end stamp (100 MHz), evaluated and staged entries per walker
    const uint32_t t_start = prof ? (uint32_t)__builtin_amdgcn_s_memrealtime() : 0u;
#endif
    // EXACT: sP0 gx gy - - | sP1 conic | sP2 r g b opacity.   fused: sP0 conic.x', 0.99 r, 0.99 g, 0.99 b | sL log2(opacity / 0.99)
    // (only the checked walk reads it).  sQ: ring of the value words whose mask bit is this walker's, in list order.
    // fused: the exponent of a pixel as a polynomial in its COLUMN inside the block, xl = 0..7:
    //     log2(alpha / 0.99) = C xl^2 + B_r xl + A_r,   C = conic.x' per entry, (B_r, A_r) per entry and pixel ROW r of the block,
    // tabulated by the lane that parks the entry (4 instructions per row).  The loop evaluates it with two fmas and no
    // subtraction; round 3's first table (conic.y' dy, conic.z' dy^2 + log2 op per row) still needed dx = gx - px and a third
    // read per entry.  Rows are 65 slots apart: the 8 rows of a slot then sit in 8 different bank pairs (ds_read_b64).
#ifndef GS_L_OFF_T
#define GS_L_OFF_T 0
#define GS_L_OFF_Q 4160
#define GS_L_OFF_P 5184
#define GS_L_OFF_L 6208
#define GS_L_TOTAL 6464
#endif
    __shared__ __attribute__((aligned(16))) unsigned char lds_x[EXACT ? 4096 : 16];         // EXACT: sP0 | sP1 | sP2 | sQ
    __shared__ __attribute__((aligned(16))) unsigned char lds_f[EXACT ? 16 : GS_L_TOTAL];   // fused: placed by hand (the offsets matter: see gs_launch_blend)
    float4* const sP0 = reinterpret_cast<float4*>(EXACT ? lds_x : lds_f + GS_L_OFF_P);
    float4* const sP1 = reinterpret_cast<float4*>(lds_x + (EXACT ? 1024 : 0));
    float4* const sP2 = reinterpret_cast<float4*>(lds_x + (EXACT ? 2048 : 0));
    uint32_t* const sQ = reinterpret_cast<uint32_t*>(EXACT ? lds_x + 3072 : lds_f + GS_L_OFF_Q);
    float* const sL = reinterpret_cast<float*>(lds_f + (EXACT ? 0 : GS_L_OFF_L));
    float2* const sT = reinterpret_cast<float2*>(lds_f + (EXACT ? 0 : GS_L_OFF_T));
    const uint32_t lane = threadIdx.x;
    const uint32_t slab_tx = f.col1 - f.col0;
    // Workgroups are dealt round-robin to the 8 XCDs (b % 8), each with its own L2.  XCD x owns the column strips
    // x, x+8, ... (a strip = SW tile columns) and walks its tiles row-major, the 4 quadrants of a tile next to each
    // other: the tiles one splat covers are then mostly blended on ONE XCD and close in time, so more of the gathers
    // of its record hit in that L2 (a placement hint only - correctness does not depend on it).  SW = 3 measured best
    // (config B: 1 -> 1002 us, 2 -> 993, 3 -> 932, 5 -> 945, 8 -> 970, 15 -> 983).
    const uint32_t b = blockIdx.x;
    const uint32_t x = b & 7u, j = b >> 3, q = j % NB, t = j / NB;
    const uint32_t SW = (dbg >> 8) & 0xffu;
    const uint32_t ns = (slab_tx + SW - 1) / SW; // strips in this slab
    if (x >= ns) return;
    const uint32_t n_x = (ns - 1 - x) / 8 + 1;
    const uint32_t lastw = slab_tx - (ns - 1) * SW;
    const uint32_t Wx = n_x * SW - ((((ns - 1) & 7u) == x) ? SW - lastw : 0u); // tile columns owned by XCD x
    if (t >= Wx * f.nty) return;
    const uint32_t ty = t / Wx, cc = t % Wx;
    const uint32_t tx = f.col0 + (x + 8u * (cc / SW)) * SW + cc % SW;
    const uint32_t lin = ty * slab_tx + (tx - f.col0);
    const uint32_t tile = tx + ty * f.ntx;
    const uint32_t start = tile > 0 ? ranges[tile - 1] : 0u;
    uint32_t end = ranges[tile];
    if (end > f.capacity) end = f.capacity;
    const float c255 = (float)(1.0 / 255.0);
    const float Wf = (float)f.width, Hf = (float)f.height;
    const uint32_t bx0 = tx * TS + (q % BPR) * 8u, by0 = ty * TS + (q / BPR) * 8u;
    const uint32_t gx = bx0 + (lane & 7), gy = by0 + (lane >> 3);
    const float pxf = (float)gx, pyf = (float)gy, bx0f = (float)bx0, by0f = (float)by0;
    const bool outside = !(gx < f.width && gy < f.height);
    bool done = outside;
    float T = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
    uint32_t staged = 0, evaluated = 0;
#ifdef GS_PROFILING
    uint32_t fp_lanes = 0, fp_quads = 0, fp_none = 0, fp_live = 0, fp_kept = 0, fp_dead = 0, fp_q1 = 0, fp_tailn = 0, fp_tailbits = 0, fp_nokeep = 0;
#endif

    // the three pieces of a record a lane fetches (uv | conic | colour, opacity), as native vectors: each is ONE register tuple
    // from the load to the asm pin below, so nothing has to be copied (and waited for) in between
    gs_u32x2 r0 = {0u, 0u};
    gs_u32x3 r1 = {0u, 0u, 0u};
    gs_u32x4 r2 = {0u, 0u, 0u, 0u};
    // this walker's bit of the mask: its 8x8 block at tile 16, the 16x16 quadrant holding it at tile 32
    const uint32_t mybit = GS_ID_BITS + (TS == 32 ? ((q / BPR) / 2u) * 2u + ((q % BPR) / 2u) : q);
    // The ID STREAM.  A walker's bit is set in about three of eight entries of its tile's list (tile 16).  Round 2 parked the
    // survivors of every 64 list entries -- 24 on average -- so the per-batch work (the row tables: 4 instructions and an LDS
    // store per pixel row, by the survivor's lane) ran on 37 % of the lanes and was 18 % of the kernel's instructions.  Now
    // the list is read 192 entries ahead, the value words whose bit is set are appended to a ring in LDS (order kept), and a
    // batch is the next 64 SURVIVORS: full lanes in the parking, 2.7x fewer batches, the same evaluations in the same order.
    uint32_t qhead = 0, qn = 0;  // ring state (wave-uniform)
    uint32_t pos = start;        // next list entry of the id stream
    uint32_t g0 = 0, g1 = 0, g2 = 0; // value words of list entries pos + {0, 64, 128} + lane, loaded one batch ahead
    // All loads are UNCONDITIONAL: an exec-masked load leaves the compiler merging old and new registers right behind the load,
    // i.e. waiting for it at once, and the prefetch would be a prefetch in name only (it was, until round 2).  The id words come
    // through a buffer descriptor of THIS tile's list: a read past its end returns 0 (no mask bit set) by the hardware's bounds
    // check, and the address is lane * 4 + a scalar offset + an immediate: no vector instruction per load.
    const __amdgpu_buffer_rsrc_t list_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(values + start), 0, (int)((end - start) * 4u), 0x00020000);
    const uint32_t lane4 = lane * 4u;
    auto load_ids = [&]() {
        const uint32_t so = (pos - start) * 4u;
        g0 = __builtin_amdgcn_raw_buffer_load_b32(list_rsrc, lane4, so, 0);
        g1 = __builtin_amdgcn_raw_buffer_load_b32(list_rsrc, lane4 + 256u, so, 0);
        g2 = __builtin_amdgcn_raw_buffer_load_b32(list_rsrc, lane4 + 512u, so, 0);
    };
    const uint32_t mybitmask = 1u << mybit;
    auto push = [&](uint32_t v, uint32_t first) {
        bool want;
        if (MASKED) want = (v & mybitmask) != 0u;  // (0 past the end of the list)
        else want = first + lane < end;            // a plain id: 0 is one
        const unsigned long long m = __ballot(want);
        if (want) sQ[__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, qhead + qn)) & 255u] = v;
        qn += (uint32_t)__popcll(m);
    };
    auto fill = [&]() { // until a batch is queued or the list is exhausted (one trip, except where the bit is rare: then the loads of the later trips are exposed)
        while (qn < 64u && pos < end) {
            push(g0, pos); push(g1, pos + 64u); push(g2, pos + 128u);
            const uint32_t adv = end - pos < 192u ? end - pos : 192u;
            staged += adv;
            pos += adv;
            if (pos < end) load_ids();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto take = [&](uint32_t& v) -> uint32_t { // the next batch: its value word for lane < the returned count
        const uint32_t cnt = qn < 64u ? qn : 64u;
        v = sQ[(qhead + lane) & 255u];
        qhead = (qhead + cnt) & 255u;
        qn -= cnt;
        return cnt;
    };
    auto fetch = [&](uint32_t v, uint32_t cnt) {
        uint32_t g = MASKED ? (v & GS_ID_MASK) : v;
#ifdef GS_PROFILING
        if (dbg & 2u) g &= 1023u; // gather from a cache-resident window
#endif
        g = lane < cnt ? g : 0u;
        const uint32_t* rec = reinterpret_cast<const uint32_t*>(gdata + (uint64_t)g * 4);
        r0 = *reinterpret_cast<const gs_u32x2*>(rec);
        r1 = *reinterpret_cast<const gs_u32x3*>(rec + 4);
        r2 = *reinterpret_cast<const gs_u32x4*>(rec + 8);
    };
    uint32_t cnt = 0, vnext = 0;
    if (start < end) {
        load_ids();
        fill();
        cnt = take(vnext);
        fetch(vnext, cnt);
    }
    while (cnt) {
        bool rel = false, npd = false;
        // The records fetched one batch ago are first needed HERE.  The empty asm pins that: without it the compiler copies the
        // loaded registers into the operand tuples of the LDS stores / packed multiplies right behind the loads, and the
        // s_waitcnt that copy needs turns the prefetch into a blocking gather (seen in the ISA of every build before this one).
        asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2));
        const float gxp = __uint_as_float(r0.x) * Wf, gyp = __uint_as_float(r0.y) * Hf; // compute_tiles.wgsl:52
        const float cx = __uint_as_float(r1.x), cy = __uint_as_float(r1.y), cz = __uint_as_float(r1.z);
        const float op = __uint_as_float(r2.w);
        // THE LIVE BOX.  Whatever decided that an entry can touch this block (the binning's mask bit at tile 16, nothing in a
        // reference-binning frame) decided it for the block's 64 pixels -- but most of them are FINAL long before the block is:
        // at an average evaluation 8 of the 64 pixels were still live, and a third of the evaluations touched final pixels only
        // (tools/blend_footprint.py, config B: the block lives as long as its last pixel).  A final pixel ignores every later
        // entry, so an entry is parked only if its alpha >= 1/255 ellipse reaches the BOUNDING BOX OF THE PIXELS THAT
        // ARE STILL LIVE (conservative closed form, block_qmin): no output bit changes, 27 % fewer evaluations at config B
        // (blend 495 -> 418 us).  The box is taken by the whole wave, outside the branch of the lanes that hold an entry.
        const unsigned long long lv = __ballot(!done);
        uint32_t lcm = (uint32_t)lv | (uint32_t)(lv >> 32);
        lcm |= lcm >> 16; lcm |= lcm >> 8; lcm &= 0xFFu; // columns of the block that hold a live pixel
        const float lc0 = (float)__builtin_ctz(lcm | 0x100u), lc1 = (float)(31 - __builtin_clz(lcm | 1u));
        const float lr0 = (float)(__builtin_ctzll(lv | (1ull << 63)) >> 3), lr1 = (float)((63 - __builtin_clzll(lv | 1ull)) >> 3);
        // ... AND THE TRANSMITTANCE THAT IS LEFT.  A live pixel whose T is just above the final threshold (1.0039e-4) accepts only
        // entries with alpha <= 1 - 1e-4 / T -- a few per cent -- and stays live, keeping its block alive, until one comes: 84 % of
        // the evaluations that the live box leaves keep NO lane (tools/blend_footprint.py).  With Tmax = the largest T among the live
        // pixels and a lower bound of the entry's alpha over the live box (the quadratic's maximum is at a corner), an entry whose
        // Tmax (1 - alpha_lo) is below 1e-4 with margins fails `T (1 - alpha) >= 1e-4` on every live pixel: not parked, no bit changes
        // (config B: blend 412 -> 278 us).
        const float Tmax = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)wave_incl_max(done ? 0u : __float_as_uint(T)), 63)); // (T >= 0; a NaN is the largest)
        if (lane < cnt) {
            const float lim = __builtin_amdgcn_logf(op * 255.0f) * 0.693147182464599609375f + 0.01f; // alpha >= c255 <=> q <= ln(255*op)
            const bool pd = (cx > 0.0f) && (cz > 0.0f) && (cx * cz - cy * cy > 0.0f);
            {   // (at tile 16 the binning's mask bit has already said "this block": the box is what is left to test)
                const float dxh = gxp - bx0f, dyh = gyp - by0f;
                float mag;
                const float qm = block_qmin(cx, cy, cz, dxh - lc1, dxh - lc0, dyh - lr1, dyh - lr0, mag);
                const bool nocull = (dbg & 4u) != 0u; // GS_OPT_BLEND_ABLATION bit 2: both culls off (tests: they must not change a bit)
                rel = lv != 0ull && (nocull || !pd || !(qm > lim + 1.0e-5f * mag)); // (no live pixel: a block outside the canvas)
#ifndef GS_NO_TMAX // (A/B: tools/build_variant.py notmax -DGS_NO_TMAX)
                if (rel && pd && !nocull) {
                    const float x0 = dxh - lc1, x1 = dxh - lc0, y0 = dyh - lr1, y1 = dyh - lr0;
                    const float ax0 = (0.5f * cx) * x0 * x0, ax1 = (0.5f * cx) * x1 * x1, by0 = (0.5f * cz) * y0 * y0, by1 = (0.5f * cz) * y1 * y1;
                    const float c00 = cy * x0 * y0, c01 = cy * x0 * y1, c10 = cy * x1 * y0, c11 = cy * x1 * y1;
                    const float qmax = __builtin_fmaxf(__builtin_fmaxf(ax0 + by0 + c00, ax0 + by1 + c01), __builtin_fmaxf(ax1 + by0 + c10, ax1 + by1 + c11));
                    const float qmag = __builtin_fmaxf(ax0, ax1) + __builtin_fmaxf(by0, by1) +
                                       __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(c00), __builtin_fabsf(c01)), __builtin_fmaxf(__builtin_fabsf(c10), __builtin_fabsf(c11)));
                    // alpha >= min(0.99, op exp(-qmax)) on every live pixel; 1 % off for the roundings of q, exp and the loop's own alpha
                    const float alo = 0.99f * __builtin_fminf(0.99f, op * __builtin_amdgcn_exp2f(-1.44269502162933349609375f * (qmax + 1.0e-5f * qmag)));
                    if (Tmax * (1.0f - alo) < 0.0000999f) rel = false; // (NaN anywhere: the comparison is false, the entry stays)
                }
#endif
#ifdef GS_BLEND_PIXTEST
                // few live pixels: the same two tests per PIXEL instead of per box (alpha within its rounding bounds at the pixel, against
                // 1/255 and against what the pixel's own T still accepts)
                if (rel && pd && !nocull && (uint32_t)__popcll(lv) <= GS_BLEND_PIXTEST) {
                    bool any = false;
                    for (unsigned long long mm = lv; mm; mm &= mm - 1ull) {
                        const int pl = __builtin_ctzll(mm);
                        const float Tp = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(T), pl));
                        const float dx = dxh - (float)(pl & 7), dy = dyh - (float)(pl >> 3);
                        const float a = (0.5f * cx) * dx * dx, b = (0.5f * cz) * dy * dy, c = cy * dx * dy;
                        const float q = a + b + c, mg = 1.0e-5f * (a + b + __builtin_fabsf(c));
                        const float ahi = 1.01f * (op * __builtin_amdgcn_exp2f(-1.44269502162933349609375f * (q - mg)));
                        const float alo = 0.99f * __builtin_fminf(0.99f, op * __builtin_amdgcn_exp2f(-1.44269502162933349609375f * (q + mg)));
                        any = any || !(ahi < c255 || Tp * (1.0f - alo) < 0.0000999f); // (NaN: kept)
                    }
                    rel = any;
                }
#endif
            }
#ifdef GS_PROFILING
            if (dbg & 1u) rel = false; // staging cost without the pixel loop
#endif
            npd = rel && !pd;
        }
        // only surviving entries are parked for the broadcast, DENSELY (slot = rank among the survivors, list order kept): the
        // evaluation loop then walks slots 0, 1, 2 ... with immediate LDS offsets, four per trip, instead of deriving an address
        // from a bit scan for every entry (one VALU move per evaluation in a loop that runs at the VALU issue rate)
#ifdef GS_PROFILING
        if ((dbg & 32u) && MASKED && __popcll(lv) <= 16) { // parked entries while <= 16 pixels are live: how many of the tile's 4 blocks does the entry's mask name?
            const unsigned long long rm = __ballot(rel);
            const uint32_t bits = __popc((vnext >> GS_ID_BITS) & 15u);
            fp_tailn += (uint32_t)__popcll(rm);
            fp_tailbits += (uint32_t)__popcll(__ballot(rel && bits >= 1u)) + (uint32_t)__popcll(__ballot(rel && bits >= 2u)) + (uint32_t)__popcll(__ballot(rel && bits >= 3u)) + (uint32_t)__popcll(__ballot(rel && bits >= 4u));
        }
#endif
        const unsigned long long m = __ballot(rel);
        if (rel) {
            const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const float L = 1.44269502162933349609375f;
            // LDS reads are priced by width (ds_read_b64 2 cycles, b128 4, b96 8 - the loop is LDS-array bound as much as
            // VALU bound), so the fused layout is two full float4 and one float: (x, y, r, g) (conic', log2 op) (b).
            // The opacity rides in the exponent: alpha = exp2(power*log2(e) + log2(op)).
            if (EXACT) {
                sP0[slot] = make_float4(gxp, gyp, 0.0f, 0.0f);
                sP1[slot] = make_float4(cx, cy, cz, 0.0f);
                sP2[slot] = make_float4(__uint_as_float(r2.x), __uint_as_float(r2.y), __uint_as_float(r2.z), op);
            } else {
                // with d = gx - bx0 (block-relative centre), dy = gy - (by0 + r):
                //   hx (d - xl)^2 + hy dy (d - xl) + hz dy^2 + lop'  =  hx xl^2 + (-2 hx d - hy dy) xl + (hx d^2 + hy d dy + hz dy^2 + lop')
                // lop' = log2(op / 0.99): the loop computes e = min(1, alpha / 0.99) with the exp's clamp modifier, and 0.99 rides in
                // the colours and the constants (alpha = min(0.99, op exp(power)), compute_tiles.wgsl:61)
                const float hx = (-0.5f * L) * cx, hy = (-L) * cy, hz = (-0.5f * L) * cz;
                const float lop = __builtin_amdgcn_logf(op) + 0.014499569695115089f; // + log2(1 / 0.99)
                const float d = gxp - bx0f;
                const float b0 = (-2.0f * hx) * d, a1 = hy * d, a0 = __builtin_fmaf(hx * d, d, lop);
                sP0[slot] = make_float4(hx, 0.99f * __uint_as_float(r2.x), 0.99f * __uint_as_float(r2.y), 0.99f * __uint_as_float(r2.z));
                sL[slot] = lop;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float dy = gyp - (by0f + (float)r);
                    sT[r * 65 + slot] = make_float2(__builtin_fmaf(-hy, dy, b0), __builtin_fmaf(dy, __builtin_fmaf(dy, hz, a1), a0));
                }
            }
        }
        // Order of issue: the id loads of the batch after the next one (inside fill), THEN the next batch's gathers.  The compiler
        // waits for everything outstanding at the top of the loop (the ring's trip count is not known to it): with the ids issued
        // last they were waited for a few instructions after their issue
        fill();
        const uint32_t cnt_next = take(vnext);
        if (cnt_next) fetch(vnext, cnt_next);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t nrel = (uint32_t)__popcll(m);
        evaluated += nrel;
        // The reference skips an entry whose power is > 0; for a positive-definite conic the power cannot be (beyond
        // rounding, which the oracle's ill-conditioning margin covers), so the fused loop only pays for that compare in
        // a batch that holds a survivor with a non-positive-definite conic (never produced by the projection's +0.3
        // low-pass; NaN records land here too).
        const uint32_t trow = (lane >> 3) * 65u;
        const float xlf = (float)(lane & 7u);
        auto walk = [&](auto checked_tag) {
            constexpr bool CHECKED = decltype(checked_tag)::value;
            auto one = [&](uint32_t e, const float4 p0, const float2 tt) { // p0 / tt: the entry's sP0 word / (fused) its row-table pair
                if (EXACT) {
                    const float dx = p0.x - pxf;
                    const float4 p1 = sP1[e];
                    const float dy = p0.y - pyf;
                    const float4 p2v = sP2[e];
                    const float t1 = p1.x * dx * dx, t2 = p1.z * dy * dy, t3 = p1.y * dx * dy;
                    const float power = -0.5f * (t1 + t2) - t3;
                    const float alpha = wg_min(0.99f, p2v.w * gs_exp(power));
                    const float test = T * (1.0f - alpha);
                    const float cond = (power <= 0.0f && alpha >= c255 && test >= 0.0001f) ? 1.0f : 0.0f;
                    cr += cond * p2v.x * alpha * T;
                    cg += cond * p2v.y * alpha * T;
                    cb += cond * p2v.z * alpha * T;
                    T = cond * test + (1.0f - cond) * T;
                } else {
                    const float pw = __builtin_fmaf(xlf, __builtin_fmaf(xlf, p0.x, tt.x), tt.y); // log2(alpha / 0.99) before the clamp
                    float ea; // min(1, alpha / 0.99): v_exp_f32 with the clamp modifier (one instruction; a v_min on alpha was a second)
                    asm("v_exp_f32_e64 %0, %1 clamp" : "=v"(ea) : "v"(pw));
                    const float k1 = (float)(1.0 / 255.0 / 0.99); // alpha >= 1/255  <=>  ea >= 1/(255 * 0.99)
#ifdef GS_PROFILING
                    if (dbg & 32u) {
                        const unsigned long long km = __ballot(ea >= k1);
                        const unsigned long long q0 = 0x000000000F0F0F0Full; // pixel (x, y) = (lane & 7, lane >> 3): quad (x >> 2, y >> 2)
                        fp_lanes += (uint32_t)__popcll(km);
                        fp_quads += (uint32_t)((km & q0) != 0) + (uint32_t)((km & (q0 << 4)) != 0) + (uint32_t)((km & (q0 << 32)) != 0) + (uint32_t)((km & (q0 << 36)) != 0);
                        fp_none += (uint32_t)(km == 0);
                        // lanes whose pixel is still live (an entry can still pass its transmittance test) and what the entry does to them
                        const unsigned long long lv = __ballot(__builtin_fmaf(-T, c255, T) >= 0.0001f);
                        const unsigned long long kp = __ballot(ea >= k1 && __builtin_fmaf(T * ea, -0.99f, T) >= 0.0001f);
                        fp_live += (uint32_t)__popcll(lv);
                        fp_kept += (uint32_t)__popcll(kp);
                        fp_dead += (uint32_t)((km & lv) == 0);
                        fp_q1 += (uint32_t)(__popcll(lv) <= 16);
                        fp_nokeep += (uint32_t)(kp == 0ull);
                    }
#endif
                    if (CHECKED) {
                        const float wgt = T * ea;                   // T alpha / 0.99 (the colours carry the 0.99)
                        const float test = __builtin_fmaf(wgt, -0.99f, T);
                        bool keep = (ea >= k1) && (test >= 0.0001f);
                        keep = keep && (pw <= sL[e]); // power <= 0
                        const float wk = keep ? wgt : 0.0f;
                        cr = __builtin_fmaf(p0.y, wk, cr);
                        cg = __builtin_fmaf(p0.z, wk, cg);
                        cb = __builtin_fmaf(p0.w, wk, cb);
                        T = keep ? test : T;
                    } else {
                        // The keep/skip decision as an exec mask instead of two selects: the two compares narrow exec, the three
                        // accumulations and the new T are written under it, exec is restored.  11 VALU per evaluation in all (2 fma,
                        // exp, mul, fma, 2 cmpx, mov, 3 fmac) -- and this loop runs at the VALU issue rate, two thirds of the frame's
                        // vector instructions are issued here.  Same operations on the kept lanes, nothing on the others.
                        float wgt, test;
                        unsigned long long save;
                        asm(
                            "v_mul_f32_e32 %[wgt], %[T], %[ea]\n\t"
                            "v_fma_f32 %[test], %[wgt], %[m99], %[T]\n\t"
                            "s_mov_b64 %[save], exec\n\t"
                            "v_cmpx_le_f32_e32 vcc, %[k1], %[ea]\n\t"
                            "v_cmpx_le_f32_e32 vcc, %[thr], %[test]\n\t"
                            "v_mov_b32_e32 %[T], %[test]\n\t"
                            "v_fmac_f32_e32 %[cr], %[c0], %[wgt]\n\t"
                            "v_fmac_f32_e32 %[cg], %[c1], %[wgt]\n\t"
                            "v_fmac_f32_e32 %[cb], %[c2], %[wgt]\n\t"
                            "s_mov_b64 exec, %[save]"
                            : [save] "=&s"(save), [wgt] "=&v"(wgt), [test] "=&v"(test), [T] "+v"(T), [cr] "+v"(cr), [cg] "+v"(cg), [cb] "+v"(cb)
                            : [k1] "s"(k1), [thr] "s"(0.0001f), [m99] "s"(-0.99f), [ea] "v"(ea), [c0] "v"(p0.y), [c1] "v"(p0.z), [c2] "v"(p0.w)
                            : "vcc");
                    }
                }
            };
            // A group's LDS reads are ALL issued before its first evaluation, and a scheduling barrier keeps them there: left to
            // itself the machine scheduler sometimes interleaves them with the arithmetic (read, wait, evaluate, read, wait ...),
            // which costs 6 % of the kernel (config B 521 vs 488 us) -- and which of the two it picked changed with edits as far
            // away as the kernel's first line (profiles/r03_notes.md)
            auto group = [&](uint32_t e, auto n_tag) {
                constexpr int G = decltype(n_tag)::value;
                float4 p[G];
                float2 t[G];
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    p[k] = sP0[e + k];
                    t[k] = EXACT ? make_float2(0.0f, 0.0f) : sT[trow + e + k]; // this pixel row's (B_r, A_r)
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < G; ++k) one(e + k, p[k], t[k]);
                __builtin_amdgcn_sched_barrier(0);
            };
            uint32_t e = 0;
            if (!EXACT) // eight per group: measured (config B) 4: 524 us, 6: 500, 8: 492, 12: 517 (the reads of a group share one wait)
                for (; e + 8u <= nrel; e += 8u) group(e, std::integral_constant<int, 8>{});
            for (; e + 4u <= nrel; e += 4u) group(e, std::integral_constant<int, 4>{});
            for (; e < nrel; ++e) group(e, std::integral_constant<int, 1>{});
        };
        if (EXACT || __ballot(npd) != 0ull) walk(std::true_type{});
        else walk(std::false_type{});
        if (EXACT) done = outside || (T * (1.0f - c255) < 0.0001f);
        else done = outside || (__builtin_fmaf(-T, c255, T) < 0.0001f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (__ballot(!done) == 0ull) break; // this block's 64 pixels are final (exact criterion)
        cnt = cnt_next;
    }
    // statistics: a tile's "staged before early exit" depth is the deepest any of its quadrants went
    // (tile_depth[] is zeroed with the control block; the host sums it)
    if (lane == 0 && staged) atomicMax(&tile_depth[lin], staged);
    if (lane == 0 && evaluated) atomicAdd(&ctl->num_evaluated[(b + 1u) & 63u], (unsigned long long)evaluated);
#ifdef GS_PROFILING
    if ((dbg & 32u) && lane == 0) {
        unsigned long long* fp = gs_blend_foot[b & 255u];
        atomicAdd(&fp[0], (unsigned long long)evaluated); atomicAdd(&fp[1], (unsigned long long)fp_lanes);
        atomicAdd(&fp[2], (unsigned long long)fp_quads); atomicAdd(&fp[3], (unsigned long long)fp_none);
        atomicAdd(&fp[4], (unsigned long long)fp_live); atomicAdd(&fp[5], (unsigned long long)fp_kept);
        atomicAdd(&fp[6], (unsigned long long)fp_dead); atomicAdd(&fp[7], (unsigned long long)fp_q1);
        atomicAdd(&fp[8], (unsigned long long)fp_tailn); atomicAdd(&fp[9], (unsigned long long)fp_tailbits); atomicAdd(&fp[10], (unsigned long long)fp_nokeep);
    }
    if (prof && lane == 0) {
        prof[b * 4u + 0u] = t_start;
        prof[b * 4u + 1u] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        prof[b * 4u + 2u] = evaluated;
        prof[b * 4u + 3u] = staged;
    }
#endif
    if (!outside) {
        const float c[3] = {cr, cg, cb};
        uint32_t px = 0xFF000000u;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float v = c[ch];
            v = (v != v) ? 0.0f : wg_min(wg_max(v, 0.0f), 1.0f);
            px |= (uint32_t)__builtin_floorf(v * 255.0f + 0.5f) << (8 * ch);
        }
        const uint64_t o = (uint64_t)gy * f.slab_w + (gx - f.px0);
        rgba8[o] = px;
        if (rgbf) {
            rgbf[o * 3 + 0] = cr;
            rgbf[o * 3 + 1] = cg;
            rgbf[o * 3 + 2] = cb;
        }
    }
}

// ---- multi-GPU presentation: slabs (rank-major, each u32[H][w_g]) -> one row-major u32[H][W] image ---
__global__ __launch_bounds__(256) void gs_assemble_kernel(const uint32_t* __restrict__ slabs, uint32_t* __restrict__ image,
                                                           uint32_t width, uint32_t height, const uint32_t* __restrict__ px_bounds,
                                                           uint32_t n_slabs, uint64_t slab_stride_px) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)width * height) return;
    const uint32_t y = (uint32_t)(t / width), x = (uint32_t)(t % width);
    uint32_t g = 0;
    while (g + 1 < n_slabs && x >= px_bounds[g + 1]) ++g;
    const uint32_t x0 = px_bounds[g], w = px_bounds[g + 1] - x0;
    // tightly packed: slabs before g hold x0*height pixels in total; padded (all-gather): fixed stride per rank
    const uint64_t slab_off = slab_stride_px ? (uint64_t)g * slab_stride_px : (uint64_t)x0 * height;
    image[t] = slabs[slab_off + (uint64_t)y * w + (x - x0)];
}

// ---- the developer views left commented out in compute_tiles.wgsl:35-38,67-70 (GS_OPT_DEBUG_VIEW) ------------------
// 1: last row/column of every tile in red; 2: list length / 1000 as grey; 3: pixel position gradient;
// 4: list length / 100 in red and green.  Drawn over the finished frame; never part of the hot path.
__global__ __launch_bounds__(256) void gs_debug_view_kernel(const uint32_t* __restrict__ ranges, GsFrame f, uint32_t view,
                                                             uint32_t* __restrict__ rgba8) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)f.slab_w * f.height) return;
    const uint32_t y = (uint32_t)(t / f.slab_w), x = f.px0 + (uint32_t)(t % f.slab_w);
    const uint32_t tile = x / f.tile_size + (y / f.tile_size) * f.ntx;
    const uint32_t start = tile > 0 ? ranges[tile - 1] : 0u, end = ranges[tile];
    const float len = (float)(end - start);
    float c[3];
    if (view == 1u) {
        if (x % f.tile_size != f.tile_size - 1u && y % f.tile_size != f.tile_size - 1u) return;
        c[0] = 1.0f; c[1] = 0.0f; c[2] = 0.0f;
    } else if (view == 2u) {
        c[0] = c[1] = c[2] = len / 1000.0f;
    } else if (view == 3u) {
        c[0] = (float)x / (float)f.width; c[1] = (float)y / (float)f.height; c[2] = 0.0f;
    } else {
        c[0] = c[1] = len / 100.0f; c[2] = 0.0f;
    }
    uint32_t px = 0xFF000000u;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float v = wg_min(wg_max(c[ch], 0.0f), 1.0f);
        px |= (uint32_t)__builtin_floorf(v * 255.0f + 0.5f) << (8 * ch);
    }
    rgba8[t] = px;
}
void gs_launch_debug_view(const uint32_t* ranges, const GsFrame& f, uint32_t view, uint32_t* rgba8, hipStream_t st) {
    const uint64_t total = (uint64_t)f.slab_w * f.height;
    if (!total || !view) return;
    hipLaunchKernelGGL(gs_debug_view_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, ranges, f, view, rgba8);
}

// ---- host launchers --------------------------------------------------------------------------------
template <int TS>
static void launch_blend_t(bool exact, dim3 grid, hipStream_t st, const uint4* gdata, const uint32_t* values, const uint32_t* ranges,
                           const GsFrame& f, uint32_t* rgba8, float* rgbf, GsControl* ctl, uint32_t dbg, uint32_t id_mask) {
    if (exact)
        hipLaunchKernelGGL((gs_blend_kernel<TS, true>), grid, dim3(TS * TS), 0, st, gdata, values, ranges, f, rgba8, rgbf, ctl, dbg, id_mask);
    else
        hipLaunchKernelGGL((gs_blend_kernel<TS, false>), grid, dim3(TS * TS), 0, st, gdata, values, ranges, f, rgba8, rgbf, ctl, dbg, id_mask);
}
template <int TS>
static void launch_quad_t(bool exact, bool masked, uint32_t nblk, uint32_t pad, hipStream_t st, const uint4* g, const uint32_t* values,
                          const uint32_t* ranges, const GsFrame& f, uint32_t* rgba8, float* rgbf, GsControl* ctl, uint32_t* tile_depth,
                          uint32_t dbg, uint32_t* prof) {
#define GS_QUAD(E, M) hipLaunchKernelGGL((gs_blend_quad_kernel<E, TS, M>), dim3(nblk), dim3(64), pad, st, g, values, ranges, f, rgba8, rgbf, ctl, tile_depth, dbg, prof)
    if (exact) { if (masked) GS_QUAD(true, true); else GS_QUAD(true, false); }
    else { if (masked) GS_QUAD(false, true); else GS_QUAD(false, false); }
#undef GS_QUAD
}
// Returns -1 for an unsupported tile size, 4 when the quadrant kernel ran (gs_stats.num_processed is then the sum of
// tile_depth[], the per-tile maximum over its four independent walkers), 1 otherwise (ctl->num_processed).
int gs_launch_blend(const void* gdata, const uint32_t* values, const uint32_t* ranges, const GsFrame& f, uint32_t* rgba8, float* rgbf,
                    GsControl* ctl, uint32_t* tile_depth, bool exact, uint32_t ablation, bool masked, hipStream_t st, uint32_t* prof,
                    uint32_t* prof_blocks) {
    uint32_t dbg = ablation; // GS_OPT_BLEND_ABLATION: 0 = product path
    const uint32_t id_mask = masked ? GS_ID_MASK : 0xFFFFFFFFu; // kernels without mask support only strip the bits
    const dim3 grid(f.col1 - f.col0, f.nty);
    if (grid.x == 0 || grid.y == 0) return 1;
    const uint4* g = (const uint4*)gdata;
    switch (f.tile_size) {
    case 8: launch_blend_t<8>(exact, grid, st, g, values, ranges, f, rgba8, rgbf, ctl, dbg, id_mask); return 1;
    case 16:
    case 32: {
        // default: one single-wave workgroup per 8x8 pixel block; GS_OPT_BLEND_ABLATION bit 3 picks the workgroup-per-tile kernel
        // (identical results: the tile-8 kernel instantiated for tile 16 / 32)
        const bool t32 = f.tile_size == 32;
        if (!(dbg & 8u)) {
            // strip width: the widest of 3, 2, 1 tile columns that still spreads the slab's columns evenly over the 8 XCDs
            // (120 or 240 columns -> 3; a 60-column slab -> 2; 30- and 15-column slabs -> 1); GS_OPT_BLEND_ABLATION
            // bits 8..15 override it
            const uint32_t slab_tx = grid.x;
            auto widest = [&](uint32_t sw) {
                const uint32_t ns = (slab_tx + sw - 1) / sw, lastw = slab_tx - (ns - 1) * sw;
                uint32_t wmax = 0;
                for (uint32_t x = 0; x < 8 && x < ns; ++x) {
                    const uint32_t n_x = (ns - 1 - x) / 8 + 1, wx = n_x * sw - ((((ns - 1) & 7u) == x) ? sw - lastw : 0u);
                    wmax = wx > wmax ? wx : wmax;
                }
                return wmax;
            };
            uint32_t SW = (dbg >> 8) & 0xffu;
            if (!SW) {
                SW = 1;
                for (uint32_t sw = 3; sw > 1; --sw)
                    if (widest(sw) * 80u <= slab_tx * 11u) { SW = sw; break; } // within 10 % of slab_tx / 8
            }
            dbg = (dbg & 0xffu) | (SW << 8);
            const uint32_t wmax = widest(SW);
            const uint32_t nblk = (t32 ? 128u : 32u) * wmax * grid.y;
            // PROFILING ONLY: bits 6/7 reserve dynamic LDS so that only 2 / 4 waves fit a SIMD (occupancy sensitivity:
            // config B 8 waves -> 982 us, 4 -> 1214, 2 -> 1890)
#ifdef GS_PROFILING
            const uint32_t pad = (dbg & 64u) ? 20480u - 3072u : (dbg & 128u) ? 10240u - 3072u : 0u;
#else
            const uint32_t pad = 0u;
#endif
            if (prof_blocks) *prof_blocks = nblk;
            if (t32) { launch_quad_t<32>(exact, masked, nblk, pad, st, g, values, ranges, f, rgba8, rgbf, ctl, tile_depth, dbg, prof); return 16; }
            launch_quad_t<16>(exact, masked, nblk, pad, st, g, values, ranges, f, rgba8, rgbf, ctl, tile_depth, dbg, prof);
            return 4;
        }
        if (t32) { launch_blend_t<32>(exact, grid, st, g, values, ranges, f, rgba8, rgbf, ctl, dbg, id_mask); return 1; } // ablation bit 3: 1024-thread workgroup per tile
        launch_blend_t<16>(exact, grid, st, g, values, ranges, f, rgba8, rgbf, ctl, dbg, id_mask);
        return 1;
    }
    default: return -1;
    }
}
void gs_launch_assemble(const void* slabs, void* image, uint32_t width, uint32_t height, const uint32_t* d_px_bounds, uint32_t n_slabs,
                        uint64_t slab_stride_px, hipStream_t st) {
    const uint64_t total = (uint64_t)width * height;
    hipLaunchKernelGGL(gs_assemble_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, (const uint32_t*)slabs, (uint32_t*)image,
                       width, height, d_px_bounds, n_slabs, slab_stride_px);
}
