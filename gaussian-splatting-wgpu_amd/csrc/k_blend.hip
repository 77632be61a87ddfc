// k_blend.hip -- per-tile front-to-back alpha blend and the rgba8unorm store.
//
// Replaces compute_tiles.wgsl::main (reference src/compute_tiles.wgsl:30-75) and the blit
// render.wgsl::vertex_main/fragment_main (src/render.wgsl:14-31), which is the identity at equal
// size: the blended image IS the presented image, so the extra full-frame pass disappears.
//
// The reference re-gathers the 64-byte GaussianData record from global memory for every pixel of
// the tile (256x redundant, compute_tiles.wgsl:49-50; README TODO "Load Gaussian data workgroup
// wide").  Here a workgroup stages a batch of its tile's sorted list ONCE into LDS (36 B/entry:
// centre in pixels, conic, opacity, colour), all pixels read the batch by LDS broadcast, and the
// tile stops as soon as every pixel is finished under the exact criterion below -- which changes
// no output bit:
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

template <int TS, int NT, bool EXACT>
__global__ __launch_bounds__(NT) void gs_blend_kernel(const uint4* __restrict__ gdata, const uint32_t* __restrict__ values,
                                                       const uint32_t* __restrict__ ranges, GsFrame f,
                                                       uint32_t* __restrict__ rgba8, float* __restrict__ rgbf, GsControl* ctl) {
    constexpr int PPT = (TS * TS) / NT; // pixels per thread
    static_assert(PPT * NT == TS * TS, "tile must divide evenly over the threads");
    __shared__ float4 sA[NT]; // gx, gy, conic.x, conic.y   (fast mode: conic pre-scaled, see below)
    __shared__ float4 sB[NT]; // conic.z, opacity, r, g
    __shared__ float sC[NT];  // b

    const uint32_t tid = threadIdx.x;
    const uint32_t tx = f.col0 + blockIdx.x, ty = blockIdx.y;
    const uint32_t tile = tx + ty * f.ntx;
    const uint32_t start = tile > 0 ? ranges[tile - 1] : 0u;
    uint32_t end = ranges[tile];
    if (end > f.capacity) end = f.capacity;

    float pxf[PPT], pyf[PPT], T[PPT], cr[PPT], cg[PPT], cb[PPT];
    bool done[PPT];
    bool all_done = true;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const uint32_t p = tid + k * NT;
        const uint32_t gx = tx * TS + (p % TS), gy = ty * TS + (p / TS);
        pxf[k] = (float)gx;
        pyf[k] = (float)gy;
        T[k] = 1.0f;
        cr[k] = cg[k] = cb[k] = 0.0f;
        done[k] = !(gx < f.width && gy < f.height);
        all_done = all_done && done[k];
    }
    const float c255 = (float)(1.0 / 255.0);
    const float Wf = (float)f.width, Hf = (float)f.height;
    uint32_t staged = 0;

    for (uint32_t b = start; b < end; b += NT) {
        // barrier: the previous batch is no longer being read; stop when every pixel is final
        if (__syncthreads_and(all_done ? 1 : 0)) break;
        const uint32_t idx = b + tid;
        if (idx < end) {
            const uint32_t g = values[idx];
            const uint4 r0 = gdata[(uint64_t)g * 4 + 0];
            const uint4 r1 = gdata[(uint64_t)g * 4 + 1];
            const uint4 r2 = gdata[(uint64_t)g * 4 + 2];
            const float gxp = __uint_as_float(r0.x) * Wf, gyp = __uint_as_float(r0.y) * Hf; // compute_tiles.wgsl:52
            float cx = __uint_as_float(r1.x), cy = __uint_as_float(r1.y), cz = __uint_as_float(r1.z);
            if (!EXACT) {
                // fold -0.5 and log2(e) into the conic once per entry: power*log2e = hx*dx*dx + hz*dy*dy + hy*dx*dy
                const float L = 1.44269502162933349609375f;
                cx = (-0.5f * L) * cx;
                cz = (-0.5f * L) * cz;
                cy = (-L) * cy;
            }
            sA[tid] = make_float4(gxp, gyp, cx, cy);
            sB[tid] = make_float4(cz, __uint_as_float(r2.w), __uint_as_float(r2.x), __uint_as_float(r2.y));
            sC[tid] = __uint_as_float(r2.z);
        }
        __syncthreads();
        const uint32_t cnt = (end - b < (uint32_t)NT) ? end - b : (uint32_t)NT;
        staged += cnt;
        if (!all_done) {
            for (uint32_t e = 0; e < cnt; ++e) {
                const float4 a4 = sA[e];
                const float4 b4 = sB[e];
                const float colb = sC[e];
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    if (done[k]) continue;
                    const float dx = a4.x - pxf[k], dy = a4.y - pyf[k];
                    if (EXACT) {
                        const float t1 = a4.z * dx * dx, t2 = b4.x * dy * dy, t3 = a4.w * dx * dy;
                        const float power = -0.5f * (t1 + t2) - t3;
                        const float alpha = wg_min(0.99f, b4.y * gs_exp(power));
                        const float test = T[k] * (1.0f - alpha);
                        const float cond = (power <= 0.0f && alpha >= c255 && test >= 0.0001f) ? 1.0f : 0.0f;
                        cr[k] += cond * b4.z * alpha * T[k];
                        cg[k] += cond * b4.w * alpha * T[k];
                        cb[k] += cond * colb * alpha * T[k];
                        T[k] = cond * test + (1.0f - cond) * T[k];
                        if (T[k] * (1.0f - c255) < 0.0001f) done[k] = true;
                    } else {
                        const float u = __builtin_fmaf(a4.z, dx, a4.w * dy);
                        const float v = (b4.x * dy) * dy;
                        const float p2 = __builtin_fmaf(dx, u, v); // power * log2(e)
                        const float alpha = __builtin_fminf(0.99f, b4.y * __builtin_amdgcn_exp2f(p2));
                        const float test = __builtin_fmaf(-T[k], alpha, T[k]);
                        if (p2 <= 0.0f && alpha >= c255 && test >= 0.0001f) {
                            const float wgt = alpha * T[k];
                            cr[k] = __builtin_fmaf(b4.z, wgt, cr[k]);
                            cg[k] = __builtin_fmaf(b4.w, wgt, cg[k]);
                            cb[k] = __builtin_fmaf(colb, wgt, cb[k]);
                            T[k] = test;
                            if (__builtin_fmaf(-test, c255, test) < 0.0001f) done[k] = true;
                        }
                    }
                }
            }
            all_done = true;
#pragma unroll
            for (int k = 0; k < PPT; ++k) all_done = all_done && done[k];
        }
    }
    if (tid == 0 && staged) atomicAdd(&ctl->num_processed, (unsigned long long)staged);

    // textureStore(render_target, xy, vec4(C, 1)) to rgba8unorm (compute_tiles.wgsl:71): clamp, *255, round
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const uint32_t p = tid + k * NT;
        const uint32_t gx = tx * TS + (p % TS), gy = ty * TS + (p / TS);
        if (gx < f.width && gy < f.height) {
            const float c[3] = {cr[k], cg[k], cb[k]};
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
                rgbf[o * 3 + 0] = cr[k];
                rgbf[o * 3 + 1] = cg[k];
                rgbf[o * 3 + 2] = cb[k];
            }
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
template <int TS, int NT>
static void launch_blend_t(bool exact, dim3 grid, hipStream_t st, const uint4* gdata, const uint32_t* values, const uint32_t* ranges,
                           const GsFrame& f, uint32_t* rgba8, float* rgbf, GsControl* ctl) {
    if (exact)
        hipLaunchKernelGGL((gs_blend_kernel<TS, NT, true>), grid, dim3(NT), 0, st, gdata, values, ranges, f, rgba8, rgbf, ctl);
    else
        hipLaunchKernelGGL((gs_blend_kernel<TS, NT, false>), grid, dim3(NT), 0, st, gdata, values, ranges, f, rgba8, rgbf, ctl);
}
// threads_per_tile: 0 = default for the tile size.
int gs_launch_blend(const void* gdata, const uint32_t* values, const uint32_t* ranges, const GsFrame& f, uint32_t* rgba8, float* rgbf,
                    GsControl* ctl, bool exact, uint32_t threads_per_tile, hipStream_t st) {
    const dim3 grid(f.col1 - f.col0, f.nty);
    if (grid.x == 0 || grid.y == 0) return 0;
    const uint4* g = (const uint4*)gdata;
    switch (f.tile_size) {
    case 8: launch_blend_t<8, 64>(exact, grid, st, g, values, ranges, f, rgba8, rgbf, ctl); return 0;
    case 16:
        if (threads_per_tile == 64) launch_blend_t<16, 64>(exact, grid, st, g, values, ranges, f, rgba8, rgbf, ctl);
        else if (threads_per_tile == 128) launch_blend_t<16, 128>(exact, grid, st, g, values, ranges, f, rgba8, rgbf, ctl);
        else launch_blend_t<16, 256>(exact, grid, st, g, values, ranges, f, rgba8, rgbf, ctl);
        return 0;
    case 32:
        if (threads_per_tile == 1024) launch_blend_t<32, 1024>(exact, grid, st, g, values, ranges, f, rgba8, rgbf, ctl);
        else launch_blend_t<32, 256>(exact, grid, st, g, values, ranges, f, rgba8, rgbf, ctl);
        return 0;
    default: return -1;
    }
}
void gs_launch_assemble(const void* slabs, void* image, uint32_t width, uint32_t height, const uint32_t* d_px_bounds, uint32_t n_slabs,
                        uint64_t slab_stride_px, hipStream_t st) {
    const uint64_t total = (uint64_t)width * height;
    hipLaunchKernelGGL(gs_assemble_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, (const uint32_t*)slabs, (uint32_t*)image,
                       width, height, d_px_bounds, n_slabs, slab_stride_px);
}
