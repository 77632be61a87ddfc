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

// Minimum over the pixel block [dxlo,dxhi] x [dylo,dyhi] (offsets g - p) of the quadratic
// q(d) = 0.5*(cx*dx^2 + cz*dy^2) + cy*dx*dy (power = -q).  For a positive-definite conic whose centre
// is outside the block the minimiser lies on an edge facing the centre: at most two 1-D problems.
__device__ __forceinline__ float block_qmin(float cx, float cy, float cz, float dxlo, float dxhi, float dylo, float dyhi,
                                            float& mag) {
    const float X = __builtin_fminf(__builtin_fmaxf(0.0f, dxlo), dxhi); // clamp(0, lo, hi)
    const float Y = __builtin_fminf(__builtin_fmaxf(0.0f, dylo), dyhi);
    float q = 3.0e38f;
    mag = 0.0f;
    if (X == 0.0f && Y == 0.0f) return 0.0f; // centre inside the block
    if (X != 0.0f) {
        const float dy = __builtin_fminf(__builtin_fmaxf(-cy * X / cz, dylo), dyhi);
        const float a = 0.5f * cx * X * X, b = 0.5f * cz * dy * dy, c = cy * X * dy;
        q = a + b + c;
        mag = __builtin_fabsf(a) + __builtin_fabsf(b) + __builtin_fabsf(c);
    }
    if (Y != 0.0f) {
        const float dx = __builtin_fminf(__builtin_fmaxf(-cy * Y / cx, dxlo), dxhi);
        const float a = 0.5f * cx * dx * dx, b = 0.5f * cz * Y * Y, c = cy * dx * Y;
        const float q2 = a + b + c;
        if (q2 < q) { q = q2; mag = __builtin_fabsf(a) + __builtin_fabsf(b) + __builtin_fabsf(c); }
    }
    return q;
}

// One workgroup per tile, TS*TS threads, one pixel per thread; wave w owns the 8x8 pixel block
// (w % (TS/8), w / (TS/8)) of the tile.  Per batch of NT staged entries every wave first builds,
// 64 entries at a time (one per lane), the mask of entries that can reach alpha >= 1/255 somewhere
// in ITS 8x8 block (conservative: an entry is dropped only if op*exp(-qmin) < c255 with a margin far
// above f32 rounding, so no output bit changes), then walks only the set bits.
template <int TS, bool EXACT>
__global__ __launch_bounds__(TS* TS) void gs_blend_kernel(const uint4* __restrict__ gdata, const uint32_t* __restrict__ values,
                                                           const uint32_t* __restrict__ ranges, GsFrame f,
                                                           uint32_t* __restrict__ rgba8, float* __restrict__ rgbf, GsControl* ctl, uint32_t dbg) {
    constexpr int NT = TS * TS;
    constexpr int ROUNDS = NT / 64; // 64-entry groups per batch
    constexpr int WPR = TS / 8;     // waves per tile row
    __shared__ float4 sA[NT]; // gx, gy, conic.x, conic.y        (fused mode: conic pre-scaled)
    __shared__ float4 sB[NT]; // conic.z, opacity, r, g
    __shared__ float2 sC[NT]; // b, ln(255*opacity) + margin

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
    bool done = !(gx < f.width && gy < f.height);
    const float c255 = (float)(1.0 / 255.0);
    const float Wf = (float)f.width, Hf = (float)f.height;
    uint32_t staged = 0, evaluated = 0;

    for (uint32_t b = start; b < end; b += NT) {
        // barrier: the previous batch is no longer being read; stop when every pixel of the tile is final
        if (__syncthreads_and(done ? 1 : 0)) break;
        const uint32_t idx = b + tid;
        if (idx < end) {
            uint32_t g = values[idx];
            if (dbg & 2u) g = (g & 1023u); // TIMING EXPERIMENT ONLY: gather from a cache-resident window
            const uint4 r0 = gdata[(uint64_t)g * 4 + 0];
            const uint4 r1 = gdata[(uint64_t)g * 4 + 1];
            const uint4 r2 = gdata[(uint64_t)g * 4 + 2];
            const float gxp = __uint_as_float(r0.x) * Wf, gyp = __uint_as_float(r0.y) * Hf; // compute_tiles.wgsl:52
            const float op = __uint_as_float(r2.w);
            float cx = __uint_as_float(r1.x), cy = __uint_as_float(r1.y), cz = __uint_as_float(r1.z);
            if (!EXACT) {
                // fused mode: fold -0.5 and log2(e) into the conic once per entry, so that
                // power*log2(e) = hx*dx^2 + hy*dx*dy + hz*dy^2
                const float L = 1.44269502162933349609375f;
                cx = (-0.5f * L) * cx;
                cy = (-L) * cy;
                cz = (-0.5f * L) * cz;
            }
            sA[tid] = make_float4(gxp, gyp, cx, cy);
            sB[tid] = make_float4(cz, op, __uint_as_float(r2.x), __uint_as_float(r2.y));
            // alpha >= c255  <=>  q <= ln(255*op); +0.01 keeps the cull conservative (rounding is ~1e-6)
            sC[tid] = make_float2(__uint_as_float(r2.z), __builtin_logf(op * 255.0f) + 0.01f);
        }
        __syncthreads();
        const uint32_t cnt = (end - b < (uint32_t)NT) ? end - b : (uint32_t)NT;
        staged += cnt;
        if (__ballot(!done) == 0ull) continue; // this wave's block is final (uniform per wave)

#pragma unroll 1
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e0 = (uint32_t)r * 64u;
            if (e0 >= cnt) break;
            bool rel = false;
            if (e0 + lane < cnt) {
                float4 a4 = sA[e0 + lane];
                float cz = sB[e0 + lane].x;
                const float lim = sC[e0 + lane].y;
                if (!EXACT) { // undo the staging scale (the margin absorbs the extra rounding)
                    const float iL = 0.693147182464599609375f;
                    a4.z *= -2.0f * iL;
                    a4.w *= -iL;
                    cz *= -2.0f * iL;
                }
                const float dxhi = a4.x - bx0f, dxlo = dxhi - 7.0f, dyhi = a4.y - by0f, dylo = dyhi - 7.0f;
                const bool pd = (a4.z > 0.0f) && (cz > 0.0f) && (a4.z * cz - a4.w * a4.w > 0.0f);
                float mag;
                const float q = block_qmin(a4.z, a4.w, cz, dxlo, dxhi, dylo, dyhi, mag);
                rel = !pd || !(q > lim + 1.0e-5f * mag); // NaNs compare false -> relevant
                if (dbg & 1u) rel = false; // TIMING EXPERIMENT ONLY: staging + cull cost without the pixel loop
            }
            unsigned long long m = __ballot(rel);
            evaluated += (uint32_t)__popcll(m);
            while (m) {
                const uint32_t e = e0 + (uint32_t)__builtin_ctzll(m);
                m &= m - 1ull;
                const float4 a4 = sA[e];
                const float4 b4 = sB[e];
                const float colb = sC[e].x;
                if (done) continue;
                const float dx = a4.x - pxf, dy = a4.y - pyf;
                if (EXACT) {
                    const float t1 = a4.z * dx * dx, t2 = b4.x * dy * dy, t3 = a4.w * dx * dy;
                    const float power = -0.5f * (t1 + t2) - t3;
                    const float alpha = wg_min(0.99f, b4.y * gs_exp(power));
                    const float test = T * (1.0f - alpha);
                    const float cond = (power <= 0.0f && alpha >= c255 && test >= 0.0001f) ? 1.0f : 0.0f;
                    cr += cond * b4.z * alpha * T;
                    cg += cond * b4.w * alpha * T;
                    cb += cond * colb * alpha * T;
                    T = cond * test + (1.0f - cond) * T;
                    if (T * (1.0f - c255) < 0.0001f) done = true;
                } else {
                    const float u = __builtin_fmaf(a4.z, dx, a4.w * dy);
                    const float v = (b4.x * dy) * dy;
                    const float p2 = __builtin_fmaf(dx, u, v);
                    const float alpha = __builtin_fminf(0.99f, b4.y * __builtin_amdgcn_exp2f(p2));
                    const float test = __builtin_fmaf(-T, alpha, T);
                    if (p2 <= 0.0f && alpha >= c255 && test >= 0.0001f) {
                        const float wgt = alpha * T;
                        cr = __builtin_fmaf(b4.z, wgt, cr);
                        cg = __builtin_fmaf(b4.w, wgt, cg);
                        cb = __builtin_fmaf(colb, wgt, cb);
                        T = test;
                        if (__builtin_fmaf(-test, c255, test) < 0.0001f) done = true;
                    }
                }
            }
        }
    }
    // statistic: spread over 64 words so that 8 160+ tiles do not serialise on one atomic
    if (tid == 0 && staged) atomicAdd(&ctl->num_processed[(blockIdx.x + blockIdx.y * gridDim.x) & 63u], (unsigned long long)staged);
    if (lane == 0 && evaluated) atomicAdd(&ctl->num_evaluated[(blockIdx.x + blockIdx.y * gridDim.x + w) & 63u], (unsigned long long)evaluated);

    // textureStore(render_target, xy, vec4(C, 1)) to rgba8unorm (compute_tiles.wgsl:71): clamp, *255, round
    if (gx < f.width && gy < f.height) {
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

// ---- host launchers --------------------------------------------------------------------------------
template <int TS>
static void launch_blend_t(bool exact, dim3 grid, hipStream_t st, const uint4* gdata, const uint32_t* values, const uint32_t* ranges,
                           const GsFrame& f, uint32_t* rgba8, float* rgbf, GsControl* ctl, uint32_t dbg) {
    if (exact)
        hipLaunchKernelGGL((gs_blend_kernel<TS, true>), grid, dim3(TS * TS), 0, st, gdata, values, ranges, f, rgba8, rgbf, ctl, dbg);
    else
        hipLaunchKernelGGL((gs_blend_kernel<TS, false>), grid, dim3(TS * TS), 0, st, gdata, values, ranges, f, rgba8, rgbf, ctl, dbg);
}
int gs_launch_blend(const void* gdata, const uint32_t* values, const uint32_t* ranges, const GsFrame& f, uint32_t* rgba8, float* rgbf,
                    GsControl* ctl, bool exact, uint32_t ablation, hipStream_t st) {
    const uint32_t dbg = ablation; // GS_OPT_BLEND_ABLATION: 0 = product path
    const dim3 grid(f.col1 - f.col0, f.nty);
    if (grid.x == 0 || grid.y == 0) return 0;
    const uint4* g = (const uint4*)gdata;
    switch (f.tile_size) {
    case 8: launch_blend_t<8>(exact, grid, st, g, values, ranges, f, rgba8, rgbf, ctl, dbg); return 0;
    case 16: launch_blend_t<16>(exact, grid, st, g, values, ranges, f, rgba8, rgbf, ctl, dbg); return 0;
    case 32: launch_blend_t<32>(exact, grid, st, g, values, ranges, f, rgba8, rgbf, ctl, dbg); return 0;
    default: return -1;
    }
}
void gs_launch_assemble(const void* slabs, void* image, uint32_t width, uint32_t height, const uint32_t* d_px_bounds, uint32_t n_slabs,
                        uint64_t slab_stride_px, hipStream_t st) {
    const uint64_t total = (uint64_t)width * height;
    hipLaunchKernelGGL(gs_assemble_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, (const uint32_t*)slabs, (uint32_t*)image,
                       width, height, d_px_bounds, n_slabs, slab_stride_px);
}
