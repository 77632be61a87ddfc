// k_preprocess.hip -- scene re-layout and the per-gaussian projection stage.
//
// Replaces process_gaussians.wgsl::main (reference src/process_gaussians.wgsl:35-106) and its
// helpers in_frustum (:108-125), compute_cov3d (:127-162), compute_cov2d (:165-218),
// compute_color_from_sh (:240-280), sigmoid (:282-294), getRect (:297-319).
//
// HBM layout (the reference keeps 320-byte AoS records, 84 B of which are padding, and reads the whole
// record of every gaussian, culled or not).  The scene is re-laid-out once at upload into
//   px,py,pz  f32[N] planes      12 B read by EVERY gaussian: the frustum cull needs nothing else
//   geo       32 B per gaussian   {log-scale, opacity | rot}
//   sh        192 B per gaussian  48 SH floats (three 64-byte sectors of its own)
// and the kernel runs in three phases per workgroup of 512 (4096 in the tight path) gaussians:
//   1. cull on the position planes (coalesced 4-byte loads), survivors compacted through LDS;
//   2. DENSE lanes (one survivor each) read their 32 B of geometry: covariance, conic, radius, rect, tile count
//      -- and, for a tile-column slab, drop out if no instance lands in it; the tight path then computes the
//      gaussian's ROW ITEMS (gs_tight.h) with the whole workgroup, one (survivor, tile row) per thread;
//   3. the 192 B of SH: colour, sigmoid, the 64-byte GaussianData store.
// A culled gaussian costs 12 B, a visible one 12 + 32 + 192 B (round 2 kept geometry and SH in one 256-byte record
// whose first 128-byte line was fetched by phase 2 AND, long after, by phase 3: 1.53x the algorithmic traffic).
// Algorithmic bytes per gaussian: 12 (culled) or 236 (visible) read; 4 (count) + 56 (visible) written.
// Bound: HBM.  No MFMA (no contraction on this path).
#include "gs_device.h"
#include "gs_kernels.h"
#include "gs_tight.h"

// ---- upload: 320-byte AoS (ply.ts:190-198) -> position planes + geometry / SH records ---------------
// One thread per (gaussian, 16-byte column of the source record); runs once per scene.
__global__ __launch_bounds__(256) void gs_repack_kernel(const float4* __restrict__ aos, uint32_t n, float* px, float* py,
                                                         float* pz, float* smax, float* geo, float* sh) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t total = (uint64_t)n * 20; // 20 float4 per source record
    if (t >= total) return;
    const uint32_t g = (uint32_t)(t / 20), c = (uint32_t)(t % 20);
    const float4 v = aos[t];
    float* r = geo + (uint64_t)g * 8;
    if (c == 0) { px[g] = v.x; py[g] = v.y; pz[g] = v.z; }
    else if (c == 1) { r[0] = v.x; r[1] = v.y; r[2] = v.z; smax[g] = __builtin_fmaxf(v.x, __builtin_fmaxf(v.y, v.z)); }
    else if (c == 2) { r[4] = v.x; r[5] = v.y; r[6] = v.z; r[7] = v.w; }
    else if (c == 3) { r[3] = v.x; }
    else {
        const uint32_t k = c - 4; // SH coefficient k: rgb -> packed floats 3k..3k+2 of the gaussian's 48
        float* q = sh + (uint64_t)g * 48 + 3 * k;
        q[0] = v.x;
        q[1] = v.y;
        q[2] = v.z;
    }
}

// ---- upload from a .ply, chunk by chunk: raw vertices (any property order, float / uchar) -> the same arrays ----------
// One thread per (vertex, value): the 11 + 48 values PackedGaussians reads from a vertex (ply.ts:166-198).  uchar values are
// value / 255 evaluated in double and rounded to f32, as the reference's Number arithmetic does (ply.ts:113-119).
__global__ __launch_bounds__(256) void gs_ply_chunk_kernel(const unsigned char* __restrict__ raw, uint32_t m, uint32_t first, GsPlyTable t,
                                                            float* px, float* py, float* pz, float* smax, float* geo, float* sh) {
    const uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (uint64_t)m * t.nsrc) return;
    const uint32_t v = (uint32_t)(id / t.nsrc), k = (uint32_t)(id % t.nsrc);
    const unsigned char* src = raw + (uint64_t)v * t.stride;
    auto rd = [&](uint32_t kk) {
        const unsigned char* q = src + t.soff[kk];
        if (t.stype[kk] == 1u) {
            uint32_t w;
            if (t.all_float) w = *reinterpret_cast<const uint32_t*>(q); // every property is 4 bytes: aligned
            else w = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
            return __uint_as_float(w);
        }
        if (t.stype[kk] == 2u) return (float)((double)q[0] / 255.0);
        return 0.0f;
    };
    const uint64_t g = (uint64_t)first + v;
    const uint32_t slot = t.slot[k];
    const float val = rd(k);
    if (slot == 0u) px[g] = val;
    else if (slot == 1u) py[g] = val;
    else if (slot == 2u) pz[g] = val;
    else if (slot >= 4u && slot <= 6u) {
        geo[g * 8 + (slot - 4u)] = val;
        if (slot == 4u) smax[g] = __builtin_fmaxf(val, __builtin_fmaxf(rd(4), rd(5))); // table entries 3, 4, 5 are scale_0..2
    } else if (slot >= 8u && slot <= 11u) geo[g * 8 + 4u + (slot - 8u)] = val;
    else if (slot == 12u) geo[g * 8 + 3u] = val;
    else { // SH coefficient kk, channel c at record float 16 + 4 kk + c -> packed float 3 kk + c
        const uint32_t kk = (slot - 16u) >> 2, ch = (slot - 16u) & 3u;
        sh[g * 48 + 3u * kk + ch] = val;
    }
}
// a degree below 3 leaves the higher coefficients of the (pre-zeroed) SH array at 0: the shader hard-codes 16 (process_gaussians.wgsl:6)
void gs_launch_ply_chunk(const void* d_raw, uint32_t m, uint32_t first, const GsPlyTable& t, const GsScene& s, hipStream_t st) {
    const uint64_t total = (uint64_t)m * t.nsrc;
    if (!total) return;
    hipLaunchKernelGGL(gs_ply_chunk_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, (const unsigned char*)d_raw, m, first, t,
                       (float*)s.px, (float*)s.py, (float*)s.pz, (float*)s.smax, (float*)s.geo, (float*)s.sh);
}

// ---- canonical 3x3 helpers: column-major m[c][r]; (A*B)[c][r] = sum_k A[k][r]*B[c][k], k ascending
struct M3 { float m[3][3]; };
__device__ __forceinline__ M3 m3_mul(const M3& A, const M3& B) {
    M3 o;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) o.m[c][r] = (A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1]) + A.m[2][r] * B.m[c][2];
    return o;
}
__device__ __forceinline__ M3 m3_t(const M3& A) {
    M3 o;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) o.m[c][r] = A.m[r][c];
    return o;
}
__device__ __forceinline__ void m4_mulv(const float* m, float x, float y, float z, float* o) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = ((m[0 + r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * 1.0f;
}

// Columns of a rect that fall in the slab; column ntx aliases to column 0 of the next tile row
// (write_tile_ids.wgsl:26-31).  Returns main-run [xa,xb) and whether the alias column is owned.
__device__ __forceinline__ void slab_cols(uint32_t rx0, uint32_t rx1, const GsFrame& f, uint32_t& xa, uint32_t& wmain,
                                          uint32_t& alias) {
    const uint32_t hi = rx1 < f.ntx ? rx1 : f.ntx; // real columns end at ntx
    xa = rx0 > f.col0 ? rx0 : f.col0;
    const uint32_t xb = hi < f.col1 ? hi : f.col1;
    wmain = xb > xa ? xb - xa : 0u;
    alias = (rx1 == f.ntx + 1u && f.col0 == 0u) ? 1u : 0u;
}

// sigmoid (:282-294): both branches evaluated, blended by a 0/1 float
__device__ __forceinline__ float sigmoid_ref(float o) {
    const float ez = gs_exp(o);
    const float cond = (o >= 0.0f) ? 1.0f : 0.0f;
    return (cond * (1.0f / (1.0f + gs_exp(-o)))) + ((1.0f - cond) * (ez / (1.0f + ez)));
}

// compute_color_from_sh (:240-280): the view direction's SH basis times the 16 RGB coefficients of the record, + 0.5, clamped.
__device__ __forceinline__ void sh_colour(const float4* __restrict__ sh, float x, float y, float z, const GsUniforms& u, float col[3]) {
    const float dx = x - u.cam[0], dy = y - u.cam[1], dz = z - u.cam[2];
    const float dl = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);
    const float X = dx / dl, Y = dy / dl, Z = dz / dl;
    const float xx = X * X, yy = Y * Y, zz = Z * Z, xy = X * Y, xz = X * Z, yz = Y * Z;
    float k[16];
    k[4] = 1.0925484305920792f * xy;
    k[5] = -1.0925484305920792f * yz;
    k[6] = 0.31539156525252005f * ((2.f * zz - xx) - yy);
    k[7] = -1.0925484305920792f * xz;
    k[8] = 0.5462742152960396f * (xx - yy);
    k[9] = (-0.5900435899266435f * Y) * (3.f * xx - yy);
    k[10] = (2.890611442640554f * xy) * Z;
    k[11] = (-0.4570457994644658f * Y) * ((4.f * zz - xx) - yy);
    k[12] = (0.3731763325901154f * Z) * ((2.f * zz - 3.f * xx) - 3.f * yy);
    k[13] = (-0.4570457994644658f * X) * ((4.f * zz - xx) - yy);
    k[14] = (1.445305721320277f * Z) * (xx - yy);
    k[15] = (-0.5900435899266435f * X) * (xx - 3.f * yy);
    float shv[48];
#pragma unroll
    for (int p = 0; p < 12; ++p) {
#ifdef PRE_ABLATE_SH
        const float4 vv = make_float4(X, Y, Z, 0.5f); // PROFILING BUILD ONLY: no SH reads
#else
        const float4 vv = sh[p];
#endif
        shv[4 * p + 0] = vv.x; shv[4 * p + 1] = vv.y; shv[4 * p + 2] = vv.z; shv[4 * p + 3] = vv.w;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float res = 0.28209479177387814f * shv[c];
        res = res + 0.4886025119029199f * ((((-Y) * shv[3 + c]) + Z * shv[6 + c]) - X * shv[9 + c]);
#pragma unroll
        for (int j = 4; j < 16; ++j) res = res + k[j] * shv[3 * j + c];
        res = res + 0.5f;
        col[c] = wg_max(res, 0.0f);
    }
}

// in_frustum (process_gaussians.wgsl:108-125) and, for a tile-column slab, a conservative reach test on the position alone.
__device__ __forceinline__ bool in_frustum_slab(const GsUniforms& u, const GsFrame& f, float x, float y, float z, float smax_log) {
    float ph[4], pv[4];
    m4_mulv(u.proj, x, y, z, ph);
    const float pw = 1.0f / (ph[3] + 0.0000001f);
    const float ndx = ph[0] * pw, ndy = ph[1] * pw;
    m4_mulv(u.view, x, y, z, pv);
    if ((pv[2] <= 0.2f) || (ndx <= -1.1f || ndx >= 1.1f || ndy <= -1.1f || ndy >= 1.1f)) return false;
    if (f.full) return true;
    // Tile-column slab: drop the gaussian here, before its record is touched, if even a conservative
    // bound of its screen radius cannot reach this rank's columns.  lambda_max(cov2d) <= |J|_F^2 * s_max^2
    // (W is orthonormal), |J|_F^2 <= (fx/z)^2 (1+limx^2) + (fy/z)^2 (1+limy^2) after the clamp of :180-186,
    // and lambda_1 <= (a+c) + sqrt(0.1) <= 2*lambda_max(cov) + 0.92; margins cover the f32 rounding.
    const float sm = __expf(smax_log) * u.scale_modifier * 1.001f;
    const float limx = 1.3f * u.tan_fovx, limy = 1.3f * u.tan_fovy;
    const float iz = 1.0f / pv[2];
    const float jf2 = (u.focal_x * iz) * (u.focal_x * iz) * (1.0f + limx * limx) + (u.focal_y * iz) * (u.focal_y * iz) * (1.0f + limy * limy);
    const float rb = 3.0f * __builtin_sqrtf(2.0f * jf2 * sm * sm + 0.92f) * 1.001f + 2.0f;
    const float pxs = ((ndx * 0.5f) + 0.5f) * (float)f.width;
    const float ts = (float)f.tile_size;
    // instance columns lie in [floor(lo/ts), floor(hi/ts)] CLAMPED to [0, ntx] (getRect, :305-313): a splat
    // entirely left of the screen still lands in column 0, one entirely right of it in column ntx (alias)
    const float lo = pxs - rb, hi = pxs + rb;
    const bool reach = (lo < (float)f.col1 * ts) && (f.col0 == 0u || hi >= (float)f.col0 * ts);
    const bool alias = (f.col0 == 0u) && (hi >= (float)f.ntx * ts); // column ntx aliases to column 0
    return (reach || alias) || !(rb == rb);
}

#ifdef GS_PROFILING
// PROFILING BUILD ONLY (tools/pre_profile.py): clock cycles the waves of the tight projection spend in its phases, summed over the
// waves of a launch ([0] cull, [1] projection arithmetic, [2] slot scan + cursor bump, [3] row-item loop, [4] colour + record,
// [5] waves).  256 copies, 128 bytes apart, picked by workgroup: 24 000 waves adding to one line would take longer than the kernel.
__device__ unsigned long long gs_pre_prof[256][16];
extern "C" __attribute__((visibility("default"))) int gs_prof_preprocess(unsigned long long* out8, int reset) {
    static unsigned long long h[256][16];
    if (out8) {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(gs_pre_prof), sizeof(h)) != hipSuccess) return -1;
        for (int k = 0; k < 8; ++k) { out8[k] = 0; for (int c = 0; c < 256; ++c) out8[k] += h[c][k]; }
    }
    if (reset) { for (auto& r : h) for (auto& v : r) v = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(gs_pre_prof), h, sizeof(h)) != hipSuccess) return -1; }
    return 0;
}
// (wave-uniform by construction: readfirstlane keeps the accumulators in scalar registers, the kernel's vector budget is untouched)
#define PRE_STAMP(k) do { const uint32_t t_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__builtin_amdgcn_s_memtime()); \
                          pacc[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pacc[k] + (t_ - tprev))); tprev = t_; } while (0)
#else
#define PRE_STAMP(k) do { } while (0)
#endif
#define PRE_G 512 // gaussians per cull chunk
#ifndef PRE_WAVES
#define PRE_WAVES 4 // waves per SIMD the register allocator must leave room for
#endif
typedef uint32_t gs_row_u32x3 __attribute__((ext_vector_type(3), aligned(4)));

// Product-path outputs of the tight projection (TIGHT = true), beside GaussianData and the count words:
//   arena   : 12-byte row items (gs_tight.h), handed out to workgroups by 16 bump cursors (GsControl::row_cursor; shard =
//             workgroup % 16 owns arena slots [shard * f.row_cap / 16, ...)): a gaussian's slots are consecutive, in row order
//   rowptr  : first arena slot of every visible gaussian
//   counts  : SLOTS of the gaussian (rows, twice that with an aliased column) | depth bucket << 22 -- what the gaussian-level
//             sort (k_gsort.hip) scans; 0 when no tile survives (the tile-count tap of a tight frame is derived from the lists)
//   GsControl::rowhist[r] += items of tile row r (the digit histogram of the row sort, k_rows.hip)

// TIGHT = true : the product path's opacity-aware binning (gs_tight.h): an instance (gaussian, tile) exists only if the tile
//                intersects the gaussian's alpha >= 1/255 ellipse; the kernel writes the gaussian's row items.
//                gs_render_debug and GS_OPT_TILE_CULL 0 use TIGHT = false: the reference's rect count, nothing else.
// NB           : 512-gaussian cull chunks per workgroup.  NB = 8 (tight path, and narrow slabs): a workgroup culls 4 096
//                gaussians (positions of 8 per thread in flight at a time), appends the survivors to ONE list in LDS and runs
//                the dense phases on full waves.  A tile-column slab keeps ~1/G of the frustum survivors: with one chunk per
//                workgroup its dense phases would run on a few dozen lanes each (131-145 us per rank at 8 slabs).  The tight
//                path always uses 8: 1 490 workgroups at 6.1 M gaussians keep the bump cursors and the row histogram cold
//                (one atomic per workgroup and trip / per touched tile row instead of eight times as many).
template <bool TIGHT, int NB>
__global__ __launch_bounds__(256, PRE_WAVES) void gs_preprocess_kernel(GsScene s, GsUniforms u, GsFrame f, uint4* __restrict__ gdata,
                                                             uint32_t* __restrict__ tile_counts, GsTightOut to) {
    __shared__ uint32_t s_ids[PRE_G * NB];
    constexpr bool KEEP_POS = NB == 2; // (whole canvas; at NB = 8 the 48 KB would halve the slab workgroups per CU: 65-100 -> 128-184 us per rank)
    __shared__ float s_pos[3][KEEP_POS ? PRE_G * NB : 1];
    __shared__ uint32_t s_cnt[2][4];
    __shared__ uint32_t s_misc[8];
    // TIGHT: per-survivor records of the row-item loop (written and read by the survivor's own wave)
    __shared__ float4 s_tA[TIGHT ? 256 : 1], s_tB[TIGHT ? 256 : 1], s_tC[TIGHT ? 256 : 1]; // gx gy kc qa | qb rcx xmax dyR | eR mode rows cols
    __shared__ uint32_t s_mark[TIGHT ? 256 : 1], s_tcnt[TIGHT ? 256 : 1], s_tgid[TIGHT ? 256 : 1];
    __shared__ uint32_t s_tw[4];
    __shared__ uint32_t s_rowhist[TIGHT ? 256 : 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t bid = blockIdx.x;
    const uint32_t base = bid * (PRE_G * NB);
    if (TIGHT) s_rowhist[tid] = 0u;
#ifdef GS_PROFILING
    uint32_t pacc[5] = {0, 0, 0, 0, 0};
    uint32_t tprev = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__builtin_amdgcn_s_memtime());
#endif

    // ---- phase 1: in_frustum (:108-125) on the position planes, survivors compacted ----
    uint32_t nvis = 0;
    if (NB > 1) {
        // survivors are appended wave by wave through one LDS counter: their order in the list is free, every output is
        // indexed by the gaussian
        if (tid == 0) s_misc[1] = 0u;
        __syncthreads();
        constexpr int GRP = 2 * NB < 8 ? 2 * NB : 8; // positions a thread keeps in flight
#pragma unroll 1
        for (int h = 0; h < 2 * NB / GRP; ++h) {
            float X[GRP], Y[GRP], Z[GRP], S[GRP];
            // every position load of the group is issued before the first test: the loads are UNCONDITIONAL (index clamped; a
            // conditional load is waited for right behind its issue) and a scheduling barrier keeps the tests behind them --
            // the compiler had made four load / wait / test rounds of this
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const uint32_t i = base + (uint32_t)(h * GRP + k) * 256u + tid;
                const uint32_t ic = i < f.n ? i : f.n - 1u;
                X[k] = s.px[ic]; Y[k] = s.py[ic]; Z[k] = s.pz[ic];
                S[k] = 0.0f;
            }
            if (!f.full) { // (one branch around the group's four loads, not one per load with its wait behind it)
#pragma unroll
                for (int k = 0; k < GRP; ++k) {
                    const uint32_t i = base + (uint32_t)(h * GRP + k) * 256u + tid;
                    S[k] = s.smax[i < f.n ? i : f.n - 1u];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const uint32_t off = (uint32_t)(h * GRP + k) * 256u + tid;
                const uint32_t i = base + off;
                bool v = false;
                if (i < f.n) {
                    v = in_frustum_slab(u, f, X[k], Y[k], Z[k], S[k]);
                    if (!v) tile_counts[i] = 0u;
                }
                const unsigned long long b = __ballot(v);
                if (b) {
                    uint32_t at = 0;
                    if (lane == (uint32_t)__builtin_ctzll(b)) at = atomicAdd(&s_misc[1], (uint32_t)__popcll(b));
                    at = (uint32_t)__builtin_amdgcn_readlane((int)at, __builtin_ctzll(b));
                    if (v) {
                        const uint32_t slot = at + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
                        s_ids[slot] = off;
                        if (KEEP_POS) { s_pos[0][slot] = X[k]; s_pos[1][slot] = Y[k]; s_pos[2][slot] = Z[k]; } // the survivor's position rides along
                    }
                }
            }
        }
        __syncthreads();
        nvis = s_misc[1];
    } else {
        bool vis[2];
        unsigned long long bal[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t i = base + k * 256 + tid;
            vis[k] = false;
            if (i < f.n) {
                vis[k] = in_frustum_slab(u, f, s.px[i], s.py[i], s.pz[i], f.full ? 0.0f : s.smax[i]);
                if (!vis[k]) tile_counts[i] = 0u;
            }
            bal[k] = __ballot(vis[k]);
            if (lane == 0) s_cnt[k][w] = (uint32_t)__popcll(bal[k]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            uint32_t before = 0;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const uint32_t c = s_cnt[k][ww];
                if (ww < (int)w) before += c;
                nvis += c;
            }
            if (k == 1) before += s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
            if (vis[k]) s_ids[before + (uint32_t)__popcll(bal[k] & ((1ull << lane) - 1ull))] = k * 256 + tid;
        }
        __syncthreads();
    }

    PRE_STAMP(0);
    // ---- phases 2 and 3: one survivor per lane ----
    const uint32_t trips = (nvis + 255u) / 256u;
    const uint32_t ts_ = f.tile_size, sub = ts_ >= 16u ? ts_ / 2u : ts_, ns = ts_ / sub;
    const float inv_ts = 1.0f / (float)ts_, inv_sub = 1.0f / (float)sub;
    const uint32_t shard = bid & 15u, shard_cap = f.row_cap >> 4;
    for (uint32_t v = tid, trip = 0; TIGHT ? trip < trips : v < nvis; v += 256, ++trip) {
        // TIGHT: every thread takes every trip (workgroup barriers inside); a thread without a survivor recomputes the
        // last one and writes nothing
        const bool active = v < nvis;
        const uint32_t i = base + s_ids[active ? v : nvis - 1u];
        const float4* geo = s.geo + (uint64_t)i * 2;
        // (NB = 2: the position the cull loaded rides through LDS: three gathers per survivor less in a kernel bound by its memory pipeline)
        const float x = KEEP_POS ? s_pos[0][v < nvis ? v : nvis - 1u] : s.px[i], y = KEEP_POS ? s_pos[1][v < nvis ? v : nvis - 1u] : s.py[i],
                    z = KEEP_POS ? s_pos[2][v < nvis ? v : nvis - 1u] : s.pz[i];
        float ph[4], pv[4];
        m4_mulv(u.proj, x, y, z, ph);
        const float pw = 1.0f / (ph[3] + 0.0000001f);
        const float ndx = ph[0] * pw, ndy = ph[1] * pw;
        m4_mulv(u.view, x, y, z, pv);
        const float uvx = (ndx * 0.5f) + 0.5f, uvy = (ndy * 0.5f) + 0.5f; // :54
        // compute_cov3d (:127-162)
        const float4 so = geo[0]; // log-scale xyz, opacity logit
        const float4 q = geo[1];
        const float mod = u.scale_modifier;
        const float sc[3] = {gs_exp(so.x) * mod, gs_exp(so.y) * mod, gs_exp(so.z) * mod};
        const float len = __builtin_sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
        const float qr = q.x / len, qx = q.y / len, qy = q.z / len, qz = q.w / len;
        M3 R;
        R.m[0][0] = 1.f - 2.f * (qy * qy + qz * qz); R.m[0][1] = 2.f * (qx * qy - qr * qz); R.m[0][2] = 2.f * (qx * qz + qr * qy);
        R.m[1][0] = 2.f * (qx * qy + qr * qz); R.m[1][1] = 1.f - 2.f * (qx * qx + qz * qz); R.m[1][2] = 2.f * (qy * qz - qr * qx);
        R.m[2][0] = 2.f * (qx * qz - qr * qy); R.m[2][1] = 2.f * (qy * qz + qr * qx); R.m[2][2] = 1.f - 2.f * (qx * qx + qy * qy);
        M3 M;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) M.m[c][r] = sc[r] * R.m[c][r];
        const M3 Sig = m3_mul(m3_t(M), M);
        // compute_cov2d (:165-218)
        float t0 = pv[0], t1 = pv[1];
        const float t2 = pv[2];
        const float limx = 1.3f * u.tan_fovx, limy = 1.3f * u.tan_fovy;
        t0 = wg_min(limx, wg_max(-limx, t0 / t2)) * t2;
        t1 = wg_min(limy, wg_max(-limy, t1 / t2)) * t2;
        M3 J;
        J.m[0][0] = u.focal_x / t2; J.m[0][1] = 0.f; J.m[0][2] = -(u.focal_x * t0) / (t2 * t2);
        J.m[1][0] = 0.f; J.m[1][1] = u.focal_y / t2; J.m[1][2] = -(u.focal_y * t1) / (t2 * t2);
        J.m[2][0] = 0.f; J.m[2][1] = 0.f; J.m[2][2] = 0.f;
        M3 Wm; // W[c][r] = V[r][c]
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) Wm.m[c][r] = u.view[r * 4 + c];
        const M3 Tm = m3_mul(Wm, J);
        M3 Vrk;
        Vrk.m[0][0] = Sig.m[0][0]; Vrk.m[0][1] = Sig.m[0][1]; Vrk.m[0][2] = Sig.m[0][2];
        Vrk.m[1][0] = Sig.m[0][1]; Vrk.m[1][1] = Sig.m[1][1]; Vrk.m[1][2] = Sig.m[1][2];
        Vrk.m[2][0] = Sig.m[0][2]; Vrk.m[2][1] = Sig.m[1][2]; Vrk.m[2][2] = Sig.m[2][2];
        const M3 cov = m3_mul(m3_mul(m3_t(Tm), m3_t(Vrk)), Tm);
        const float ca = cov.m[0][0] + 0.3f, cb = cov.m[0][1], cc = cov.m[1][1] + 0.3f;
        const float det = ca * cc - cb * cb;
        uint32_t count = 0;
        uint32_t rminx = 0, rminy = 0, rmaxx = 0, rmaxy = 0;
        uint32_t t_xa = 0, t_wmain = 0, t_alias = 0;
        float conx = 0.f, cony = 0.f, conz = 0.f, opacity = 0.f;
        if (det != 0.0f) { // :60 (det == 0 -> count 0, nothing written)
            const float det_inv = 1.0f / det;
            conx = cc * det_inv; cony = (-cb) * det_inv; conz = ca * det_inv;
            const float mid = 0.5f * (ca + cc);
            const float sq = __builtin_sqrtf(wg_max(0.1f, mid * mid - det));
            const float l1 = mid + sq, l2 = mid - sq;
            const float radius = __builtin_ceilf(3.f * __builtin_sqrtf(wg_max(l1, l2)));
            // getRect (:297-319)
            const float pxs = uvx * (float)f.width, pys = uvy * (float)f.height;
            const int ts = (int)f.tile_size, ntx = (int)f.ntx, nty = (int)f.nty;
            rminx = (uint32_t)wg_mini(ntx, wg_maxi(0, f2i_sat(pxs - radius) / ts));
            rminy = (uint32_t)wg_mini(nty, wg_maxi(0, f2i_sat(pys - radius) / ts));
            rmaxx = (uint32_t)(wg_mini(ntx, wg_maxi(0, f2i_sat(pxs + radius) / ts)) + 1);
            rmaxy = (uint32_t)(wg_mini(nty, wg_maxi(0, f2i_sat(pys + radius) / ts)) + 1);
            uint32_t xa, wmain, alias; // columns of the rect inside this ctx's slab (the whole rect when f.full)
            slab_cols(rminx, rmaxx, f, xa, wmain, alias);
            count = f.full ? (rmaxy - rminy) * (rmaxx - rminx) /* :86 */ : (rmaxy - rminy) * (wmain + alias);
            if (TIGHT) { t_xa = xa; t_wmain = wmain; t_alias = alias; }
        }
        uint32_t nslots = 0, rowptr = 0;
        if (TIGHT) {
            // The slots of a wave's 64 survivors are computed by THAT wave, one (survivor, slot) per lane and trip of the loop
            // below: the per-survivor records in LDS are written and read by one wave only (no workgroup barrier inside the
            // loop), and a lane looping over its own rect's rows would wait for the tallest rect of its wave.
            uint32_t ra = 0, nrows = 0;
            if (active && count) {
                opacity = sigmoid_ref(so.w);
                const TightG tg = tight_setup(uvx, uvy, conx, cony, conz, opacity, (float)f.width, (float)f.height);
                if (tg.mode == 0u) count = 0u; // opacity below 1/255: no pixel can pass
                else {
                    uint32_t rb;
                    tight_rows(tg, rminy, rmaxy, f.tile_size, inv_ts, f.nty, t_alias, ra, rb);
                    nrows = rb - ra;
                    nslots = nrows * (1u + t_alias);
                    if (!nslots) count = 0u;
                    s_tA[tid] = make_float4(tg.gx, tg.gy, tg.kc, tg.qa);
                    s_tB[tid] = make_float4(tg.qb, tg.rcx, tg.xmax, tg.dyR);
                    s_tC[tid] = make_float4(tg.eR, __uint_as_float(tg.mode), __uint_as_float(ra | (nrows << 16)),
                                            __uint_as_float(t_xa | (t_wmain << 16) | (t_alias << 31)));
                    s_tgid[tid] = i;
                }
            } else {
                count = 0u;
            }
            s_tcnt[tid] = 0u;
            PRE_STAMP(1);
            const uint32_t rincl = wave_incl_scan(nslots, lane);
            const uint32_t Rw = (uint32_t)__builtin_amdgcn_readlane((int)rincl, 63); // slots of this wave's survivors
            const uint32_t myrp = rincl - nslots;
            if (lane == 63) s_tw[w] = rincl;
            __syncthreads();
            uint32_t wbase = 0, R = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { if (k < (int)w) wbase += s_tw[k]; R += s_tw[k]; }
            if (tid == 0) { // this trip's slots: one bump of the shard's cursor
                uint32_t at = 0, ok = 1u;
                if (R) {
                    at = atomicAdd(&to.ctl->row_cursor[shard], R);
                    if (at > shard_cap || R > shard_cap - at) { ok = 0u; to.ctl->overflow = 1u; } // the frame does not fit: gs_wait grows the arena and re-renders
                }
                s_misc[2] = at; s_misc[3] = ok;
            }
            __syncthreads();
            const uint32_t abase = shard * shard_cap + s_misc[2] + wbase;
            const bool ok = s_misc[3] != 0u;
            PRE_STAMP(2);
            uint32_t jcarry = 0u; // owner (+1) of the slot just before the batch
            for (uint32_t rb0 = 0; rb0 < Rw; rb0 += 64u) {
                // owner of a slot = the survivor whose slots [myrp, myrp + nslots) hold it: every survivor with slots marks the batch
                // position of its first one, a running maximum (DPP) spreads the marks
                s_mark[tid] = 0u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (nslots && myrp >= rb0 && myrp < rb0 + 64u) s_mark[(w << 6) + (myrp - rb0)] = lane + 1u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                uint32_t m = wave_incl_max(s_mark[tid]);
                m = m > jcarry ? m : jcarry;
                jcarry = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
                const uint32_t ri = rb0 + lane;
                const uint32_t rpj = (uint32_t)__shfl((int)myrp, (int)(m ? m - 1u : 0u), 64); // (every lane takes part: the loop is wave-uniform)
                if (ri < Rw) {
                    const uint32_t j = (w << 6) + (m - 1u);
                    const float4 a = s_tA[j], b = s_tB[j], c4 = s_tC[j];
                    TightG g;
                    g.gx = a.x; g.gy = a.y; g.kc = a.z; g.qa = a.w; g.qb = b.x; g.rcx = b.y; g.xmax = b.z; g.dyR = b.w;
                    g.eR = c4.x; g.mode = __float_as_uint(c4.y);
                    g.cx = g.cy = g.cz = g.cxz = g.lim2 = g.ymax = 0.0f; // (setup only)
                    const uint32_t rw = __float_as_uint(c4.z), cw = __float_as_uint(c4.w);
                    uint32_t w1, w2;
                    const uint32_t ilen = tight_slot_item(g, ri - rpj, rw & 0xFFFFu, rw >> 16, f.tile_size, inv_ts, sub, inv_sub, ns, f.nty, cw & 0xFFFFu,
                                                          (cw >> 16) & 0x7FFFu, cw >> 31, w1, w2);
                    if (ilen) {
                        atomicAdd(&s_tcnt[j], ilen);
                        atomicAdd(&s_rowhist[w1 & 0xFFu], 1u);
                    }
                    if (ok) {
                        gs_row_u32x3 it;
                        it.x = ilen ? s_tgid[j] : GS_ROW_HOLE; it.y = w1; it.z = w2;
                        *reinterpret_cast<gs_row_u32x3*>(to.arena + (uint64_t)(abase + ri) * 3u) = it;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (count) count = ok ? s_tcnt[tid] : 0u;
            rowptr = abase + myrp;
            PRE_STAMP(3);
            if (!active) continue;
        }
        // low 22 bits: tile count (TIGHT: row-item slots); high 10: the key's depth bucket, u32(min(50*depth, 999)) (write_tile_ids.wgsl:31)
        const uint32_t bucket = f2u_sat(wg_min(50.0f * pv[2], 999.0f));
        tile_counts[i] = count ? ((TIGHT ? nslots : count) | (bucket << GS_COUNT_BITS)) : 0u;
        if (count == 0) continue; // det == 0, no tile survives, or (slab mode) no instance in this rank's tile columns
        if (TIGHT) to.rowptr[i] = rowptr;

        // ---- phase 3: colour (:240-280) and opacity (:282-294) ----
        float col[3];
        sh_colour(s.sh + (uint64_t)i * 12, x, y, z, u, col);
        if (!TIGHT) opacity = sigmoid_ref(so.w);
        // GaussianData record (:97-104), 64 B as four 16-byte stores
        uint4* o4 = gdata + (uint64_t)i * 4;
        o4[0] = make_uint4(__float_as_uint(uvx), __float_as_uint(uvy), 0u, 0u);
        o4[1] = make_uint4(__float_as_uint(conx), __float_as_uint(cony), __float_as_uint(conz), __float_as_uint(pv[2]));
        o4[2] = make_uint4(__float_as_uint(col[0]), __float_as_uint(col[1]), __float_as_uint(col[2]), __float_as_uint(opacity));
        o4[3] = make_uint4(rminx, rminy, rmaxx, rmaxy);
    }
    PRE_STAMP(4);
#ifdef GS_PROFILING
    if (TIGHT && lane == 0) {
        unsigned long long* pp = gs_pre_prof[(bid * 4u + w) & 255u];
        for (int k = 0; k < 5; ++k) atomicAdd(&pp[k], (unsigned long long)pacc[k]);
        atomicAdd(&pp[5], 1ull);
    }
#endif
    if (TIGHT) { // the workgroup's items per tile row -> the row sort's digit histogram
        __syncthreads();
        const uint32_t c = s_rowhist[tid];
        if (c) atomicAdd(&to.ctl->rowhist[bid & 7u][tid], c);
    }
}

// ---- host launchers --------------------------------------------------------------------------------
void gs_launch_repack(const void* d_aos, uint32_t n, const GsScene& s, hipStream_t st) {
    const uint64_t total = (uint64_t)n * 20;
    const uint32_t blocks = (uint32_t)((total + 255) / 256);
    if (!blocks) return;
    hipLaunchKernelGGL(gs_repack_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)d_aos, n, (float*)s.px, (float*)s.py,
                       (float*)s.pz, (float*)s.smax, (float*)s.geo, (float*)s.sh);
}
// The projection's launch as data: the one kernel of a frame whose arguments change from frame to frame (the uniforms, by
// value), so a captured frame graph (gs_runtime.hip) re-launches it with updated parameters.
void gs_preprocess_prepare(GsPreprocessLaunch& L, const GsScene& s, const GsUniforms& u, const GsFrame& f, void* gdata, uint32_t* counts,
                           bool tight, uint32_t* arena, uint32_t* rowptr, GsControl* ctl, uint32_t tight_nb) {
    // cull chunks per workgroup (see NB).  Reference binning: 8 for a slab narrower than 30 % of the canvas, 4 up to 75 %, else 1.
    // Tight: 2 on the whole canvas (5 958 workgroups at 6.1 M gaussians: with 8 the 1 490 workgroups were 1.45 residency rounds,
    // the second one half empty), 8 / 4 for slabs as above.
    const uint32_t wcols = f.col1 - f.col0;
    uint32_t nb = f.full || wcols * 4u > f.ntx * 3u ? 1u : (wcols * 10u > f.ntx * 3u ? 4u : 8u);
    if (tight) nb = tight_nb ? tight_nb : (nb == 1u ? 2u : nb);
    L.blocks = (f.n + PRE_G * nb - 1) / (PRE_G * nb);
    if (tight) L.func = nb == 8u ? (const void*)&gs_preprocess_kernel<true, 8> : nb == 4u ? (const void*)&gs_preprocess_kernel<true, 4>
                                                                                       : (const void*)&gs_preprocess_kernel<true, 2>;
    else L.func = nb == 8u ? (const void*)&gs_preprocess_kernel<false, 8> : nb == 4u ? (const void*)&gs_preprocess_kernel<false, 4>
                                                                                        : (const void*)&gs_preprocess_kernel<false, 1>;
    L.s = s; L.u = u; L.f = f; L.gdata = gdata; L.counts = counts;
    L.to.arena = arena; L.to.rowptr = rowptr; L.to.ctl = ctl;
    L.args[0] = &L.s; L.args[1] = &L.u; L.args[2] = &L.f; L.args[3] = &L.gdata; L.args[4] = &L.counts; L.args[5] = &L.to;
}
void gs_launch_preprocess(GsPreprocessLaunch& L, hipStream_t st) {
    if (!L.f.n) return;
    (void)hipLaunchKernel(L.func, dim3(L.blocks), dim3(256), L.args, 0, st);
}
