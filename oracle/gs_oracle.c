/*
 * gs_oracle.c -- CPU oracle for the forward splat-render path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is a plain-C restatement of the WGSL/TypeScript
 * algorithm of ldyken53/gaussian-splatting-wgpu (reference @ 2024-10-08).  It is the checker
 * for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing under gaussian-splatting-wgpu_amd/ links, imports or calls it.
 *
 * PARITY PINNING
 *   scan  : pinned by the reference's own serial definition (exclusive_scan.ts:105-112).
 *   sort  : pinned by the reference's known-answer test (radix_sort/utils.ts:55-81).
 *   everything else (projection, key emit, ranges, blend): PARITY UNPINNED -- the reference
 *   ships no tests, fixtures or golden images for those stages and no WebGPU runtime exists in
 *   this environment.  They are cross-checked by an independent numpy
 *   restatement (oracle/np_oracle.py) and by analytic single-splat cases (tests/).
 *
 * CANONICAL FLOATING-POINT SEMANTICS (what "bit-exact" means for this project)
 *   WGSL leaves operation order, fusion and the accuracy of exp/sqrt/normalize to the
 *   implementation, so a live WebGPU adapter is not bit-defined.  The oracle fixes one legal
 *   evaluation: every expression is evaluated exactly as written in the shader, left to right,
 *   each +,-,*,/ and sqrt individually rounded to IEEE-754 binary32 (no contraction; build with
 *   -ffp-contract=off), matrix products as k-ascending sums, min/max per the WGSL definition
 *   (max(a,b) = a<b ? b : a), f32->i32 conversion saturating with NaN -> 0, and exp() is the
 *   fully specified function gso_expf below (Cody-Waite reduction + degree-5 polynomial built
 *   from IEEE fma/mul/add only, so any conforming machine reproduces it bit for bit).
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GSO_API __attribute__((visibility("default")))

/* ---- layouts (packing.ts:142-152,154-174,233-245; ply.ts:190-198) ------------------- */
#define SPLAT_STRIDE_F 80 /* 320 B: pos@0, log_scale@4, rot@8, opacity@12, sh@16 (+4/coeff) */
#define GDATA_STRIDE_W 16 /* 64 B: uv@0, conic@4, depth@7, color@8, opacity@11, rect@12   */

typedef union { float f; uint32_t u; int32_t i; } f32bits;

/* ---- canonical scalar helpers -------------------------------------------------------- */
static inline float wmaxf(float a, float b) { return (a < b) ? b : a; } /* WGSL max */
static inline float wminf(float a, float b) { return (b < a) ? b : a; } /* WGSL min */
static inline int32_t wmaxi(int32_t a, int32_t b) { return (a < b) ? b : a; }
static inline int32_t wmini(int32_t a, int32_t b) { return (b < a) ? b : a; }

/* f32 -> i32, truncating, saturating, NaN -> 0 (WGSL value conversion). */
static inline int32_t f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
static inline uint32_t f2u_sat(float x) {
    if (x != x) return 0;
    if (x >= 4294967296.0f) return UINT32_MAX;
    if (x <= 0.0f) return 0;
    return (uint32_t)x;
}

/*
 * Canonical exp.  n = rint(x*log2(e)); r = x - n*ln2 (two-constant Cody-Waite, fma);
 * e^r = 1 + r + r^2 * P(r), P of degree 5 (the classic single-precision coefficient set);
 * result = (p * 2^a) * 2^b with a = n>>1, b = n-a, so overflow/denormal results get one
 * correct rounding.  Max error ~1 ulp; WGSL allows 3 + 2|x| ulp.
 */
GSO_API float gso_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -103.97208404541015625f) return 0.0f;
    const float LOG2E = 1.44269502162933349609375f;
    const float LN2_HI = 0.693145751953125f;          /* 0x3f317200 */
    const float LN2_LO = 1.42860677465796470642e-06f; /* 0x35bfbe8e */
    float nf = rintf(x * LOG2E);
    float r = fmaf(-nf, LN2_HI, x);
    r = fmaf(-nf, LN2_LO, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float z = r * r;
    float y = fmaf(p, z, r);
    y = y + 1.0f;
    int32_t n = (int32_t)nf;
    int32_t a = n >> 1;
    int32_t b = n - a;
    f32bits sa, sb;
    sa.u = (uint32_t)(a + 127) << 23;
    sb.u = (uint32_t)(b + 127) << 23;
    return (y * sa.f) * sb.f;
}

/* 3x3 matrices are stored column-major m[c][r] like WGSL.  (A*B)[c][r] = sum_k A[k][r]*B[c][k] */
static inline void mat3_mul(const float A[3][3], const float B[3][3], float out[3][3]) {
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r)
            out[c][r] = (A[0][r] * B[c][0] + A[1][r] * B[c][1]) + A[2][r] * B[c][2];
}
static inline void mat3_transpose(const float A[3][3], float out[3][3]) {
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) out[c][r] = A[r][c];
}
/* mat4 (column-major flat, m[c*4+r]) times (x,y,z,w): ((c0*x + c1*y) + c2*z) + c3*w */
static inline void mat4_mulv(const float* m, float x, float y, float z, float w, float out[4]) {
    for (int r = 0; r < 4; ++r) out[r] = ((m[0 + r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}

/* Uniform block, 160 B (renderer.ts:15-24,371-392; process_gaussians.wgsl:16-25). */
typedef struct {
    float view[16];
    float proj[16];
    float cam_pos[3];
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
} gso_uniforms;

/* process_gaussians.wgsl:127-162 */
static void compute_cov3d(const float ls[3], const float rot[4], float modifier, float cov3d[6]) {
    float s[3] = {gso_expf(ls[0]) * modifier, gso_expf(ls[1]) * modifier, gso_expf(ls[2]) * modifier};
    float len = sqrtf(((rot[0] * rot[0] + rot[1] * rot[1]) + rot[2] * rot[2]) + rot[3] * rot[3]);
    float r = rot[0] / len, x = rot[1] / len, y = rot[2] / len, z = rot[3] / len;
    float R[3][3] = {
        {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
        {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
        {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)},
    };
    /* M = S * R with S diagonal: M[c][r] = s[r] * R[c][r] (the zero terms of the product add +-0) */
    float M[3][3], Mt[3][3], Sigma[3][3];
    for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr) M[c][rr] = s[rr] * R[c][rr];
    mat3_transpose(M, Mt);
    mat3_mul(Mt, M, Sigma);
    cov3d[0] = Sigma[0][0]; cov3d[1] = Sigma[0][1]; cov3d[2] = Sigma[0][2];
    cov3d[3] = Sigma[1][1]; cov3d[4] = Sigma[1][2]; cov3d[5] = Sigma[2][2];
}

/* process_gaussians.wgsl:165-218 */
static void compute_cov2d(const float pos[3], const float ls[3], const float rot[4], const gso_uniforms* u,
                          float out[3]) {
    float cov3d[6];
    compute_cov3d(ls, rot, u->scale_modifier, cov3d);
    float t[4];
    mat4_mulv(u->view, pos[0], pos[1], pos[2], 1.0f, t);
    float limx = 1.3f * u->tan_fovx, limy = 1.3f * u->tan_fovy;
    float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = wminf(limx, wmaxf(-limx, txtz)) * t[2];
    t[1] = wminf(limy, wmaxf(-limy, tytz)) * t[2];
    float fx = u->focal_x, fy = u->focal_y;
    float J[3][3] = {
        {fx / t[2], 0.f, -(fx * t[0]) / (t[2] * t[2])},
        {0.f, fy / t[2], -(fy * t[1]) / (t[2] * t[2])},
        {0.f, 0.f, 0.f},
    };
    const float* V = u->view; /* V[c*4+r] */
    float W[3][3] = {
        {V[0 * 4 + 0], V[1 * 4 + 0], V[2 * 4 + 0]},
        {V[0 * 4 + 1], V[1 * 4 + 1], V[2 * 4 + 1]},
        {V[0 * 4 + 2], V[1 * 4 + 2], V[2 * 4 + 2]},
    };
    float T[3][3], Tt[3][3], Vrk[3][3], Vrkt[3][3], tmp[3][3], cov[3][3];
    mat3_mul(W, J, T);
    Vrk[0][0] = cov3d[0]; Vrk[0][1] = cov3d[1]; Vrk[0][2] = cov3d[2];
    Vrk[1][0] = cov3d[1]; Vrk[1][1] = cov3d[3]; Vrk[1][2] = cov3d[4];
    Vrk[2][0] = cov3d[2]; Vrk[2][1] = cov3d[4]; Vrk[2][2] = cov3d[5];
    mat3_transpose(T, Tt);
    mat3_transpose(Vrk, Vrkt);
    mat3_mul(Tt, Vrkt, tmp);
    mat3_mul(tmp, T, cov);
    cov[0][0] += 0.3f;
    cov[1][1] += 0.3f;
    out[0] = cov[0][0]; out[1] = cov[0][1]; out[2] = cov[1][1];
}

/* process_gaussians.wgsl:221-280 */
static void color_from_sh(const float pos[3], const float* sh /* 16 x stride 4 */, const float cam[3], float out[3]) {
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                         0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                         -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    float d[3] = {pos[0] - cam[0], pos[1] - cam[1], pos[2] - cam[2]};
    float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    float x = d[0] / len, y = d[1] / len, z = d[2] / len;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
    float k4 = C2[0] * xy, k5 = C2[1] * yz, k6 = C2[2] * ((2.f * zz - xx) - yy), k7 = C2[3] * xz,
          k8 = C2[4] * (xx - yy);
    float k9 = (C3[0] * y) * (3.f * xx - yy), k10 = (C3[1] * xy) * z, k11 = (C3[2] * y) * ((4.f * zz - xx) - yy),
          k12 = (C3[3] * z) * ((2.f * zz - 3.f * xx) - 3.f * yy), k13 = (C3[4] * x) * ((4.f * zz - xx) - yy),
          k14 = (C3[5] * z) * (xx - yy), k15 = (C3[6] * x) * (xx - 3.f * yy);
    for (int c = 0; c < 3; ++c) {
#define SH(i) sh[(i) * 4 + c]
        float res = C0 * SH(0);
        res = res + C1 * ((((-y) * SH(1)) + z * SH(2)) - x * SH(3));
        res = ((((res + k4 * SH(4)) + k5 * SH(5)) + k6 * SH(6)) + k7 * SH(7)) + k8 * SH(8);
        res = ((((((res + k9 * SH(9)) + k10 * SH(10)) + k11 * SH(11)) + k12 * SH(12)) + k13 * SH(13)) +
               k14 * SH(14)) + k15 * SH(15);
#undef SH
        res = res + 0.5f;
        out[c] = wmaxf(res, 0.0f);
    }
}

/* process_gaussians.wgsl:282-294 (both branches evaluated, blended by a 0/1 float) */
static float sigmoid_ref(float x) {
    float z = gso_expf(x);
    float cond = (x >= 0.0f) ? 1.0f : 0.0f;
    return (cond * (1.0f / (1.0f + gso_expf(-x)))) + ((1.0f - cond) * (z / (1.0f + z)));
}

static inline uint32_t ceil_div_tiles(uint32_t extent, uint32_t ts) {
    /* ceil(f32(extent)/f32(ts)) evaluated in f32 (process_gaussians.wgsl:79) */
    return (uint32_t)f2i_sat(ceilf((float)extent / (float)ts));
}

/*
 * Number of (x,y) instances of rect whose effective tile column lies in [col0,col1).
 * Column x == ntx aliases to column 0 of the next tile row (write_tile_ids.wgsl:26-31).
 */
static inline uint32_t slab_columns(uint32_t rx0, uint32_t rx1, uint32_t ntx, uint32_t col0, uint32_t col1) {
    uint32_t n = 0;
    for (uint32_t x = rx0; x < rx1; ++x) {
        uint32_t c = (x == ntx) ? 0u : x;
        n += (c >= col0 && c < col1);
    }
    return n;
}

/*
 * Stage A.1: process_gaussians.wgsl:35-106.  gdata (64 B/record) must be writable for n records;
 * culled records are written as all-zero (the reference relies on the per-frame clear,
 * renderer.ts:579).  Tile counts are restricted to tile columns [col0,col1) (single GPU: 0,ntx).
 */
GSO_API void gso_preprocess(const float* splats, uint64_t n, const float* uniforms, uint32_t W, uint32_t H,
                            uint32_t ts, uint32_t col0, uint32_t col1, uint32_t* gdata, uint32_t* tile_counts) {
    const gso_uniforms* u = (const gso_uniforms*)uniforms;
    const uint32_t ntx = ceil_div_tiles(W, ts), nty = ceil_div_tiles(H, ts);
    const float ntxf = ceilf((float)W / (float)ts), ntyf = ceilf((float)H / (float)ts);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const float* g = splats + (size_t)i * SPLAT_STRIDE_F;
        uint32_t* o = gdata + (size_t)i * GDATA_STRIDE_W;
        memset(o, 0, 64);
        tile_counts[i] = 0;
        const float* pos = g;
        /* in_frustum, :108-125 */
        float ph[4], pv[4];
        mat4_mulv(u->proj, pos[0], pos[1], pos[2], 1.0f, ph);
        float pw = 1.0f / (ph[3] + 0.0000001f);
        float ppx = ph[0] * pw, ppy = ph[1] * pw;
        mat4_mulv(u->view, pos[0], pos[1], pos[2], 1.0f, pv);
        if (pv[2] <= 0.2f || (ppx <= -1.1f || ppx >= 1.1f || ppy <= -1.1f || ppy >= 1.1f)) continue;
        /* :50-54 */
        float uvx = (ppx * 0.5f) + 0.5f, uvy = (ppy * 0.5f) + 0.5f;
        float c2[3];
        compute_cov2d(pos, g + 4, g + 8, u, c2);
        float det = c2[0] * c2[2] - c2[1] * c2[1];
        if (det == 0.0f) continue;
        float det_inv = 1.0f / det;
        float conic[3] = {c2[2] * det_inv, (-c2[1]) * det_inv, c2[0] * det_inv};
        float mid = 0.5f * (c2[0] + c2[2]);
        float sq = sqrtf(wmaxf(0.1f, mid * mid - det));
        float l1 = mid + sq, l2 = mid - sq;
        float radius = ceilf(3.f * sqrtf(wmaxf(l1, l2)));
        /* getRect, :297-319 */
        float px = uvx * (float)W, py = uvy * (float)H;
        int32_t t_s = (int32_t)ts;
        int32_t ntxi = f2i_sat(ntxf), ntyi = f2i_sat(ntyf);
        uint32_t rminx = (uint32_t)wmini(ntxi, wmaxi(0, f2i_sat(px - radius) / t_s));
        uint32_t rminy = (uint32_t)wmini(ntyi, wmaxi(0, f2i_sat(py - radius) / t_s));
        uint32_t rmaxx = (uint32_t)(wmini(ntxi, wmaxi(0, f2i_sat(px + radius) / t_s)) + 1);
        uint32_t rmaxy = (uint32_t)(wmini(ntyi, wmaxi(0, f2i_sat(py + radius) / t_s)) + 1);
        (void)nty;
        uint32_t cnt;
        if (col0 == 0 && col1 >= ntx)
            cnt = (rmaxy - rminy) * (rmaxx - rminx); /* :86 */
        else
            cnt = (rmaxy - rminy) * slab_columns(rminx, rmaxx, ntx, col0, col1);
        if (cnt == 0) continue; /* slab mode: no instance in this rank's tile columns -> treated as culled */
        float color[3];
        color_from_sh(pos, g + 16, u->cam_pos, color);
        float opacity = sigmoid_ref(g[12]);
        f32bits b;
        b.f = uvx; o[0] = b.u;
        b.f = uvy; o[1] = b.u;
        b.f = conic[0]; o[4] = b.u;
        b.f = conic[1]; o[5] = b.u;
        b.f = conic[2]; o[6] = b.u;
        b.f = pv[2]; o[7] = b.u;
        b.f = color[0]; o[8] = b.u;
        b.f = color[1]; o[9] = b.u;
        b.f = color[2]; o[10] = b.u;
        b.f = opacity; o[11] = b.u;
        o[12] = rminx; o[13] = rminy; o[14] = rmaxx; o[15] = rmaxy;
        tile_counts[i] = cnt;
    }
}

/* Stage A.2: exclusive_scan.ts:105-112,208-325.  out[i] = sum_{j<i} in[j]; returns the total. */
GSO_API uint32_t gso_scan(const uint32_t* counts, uint64_t n, uint32_t* offsets) {
    uint32_t acc = 0;
    for (uint64_t i = 0; i < n; ++i) {
        offsets[i] = acc;
        acc += counts[i];
    }
    return acc;
}

/* Stage A.3: write_tile_ids.wgsl:18-35 (only i < n emit; y outer, x inner). */
GSO_API void gso_emit(const uint32_t* gdata, const uint32_t* offsets, const uint32_t* tile_counts, uint64_t n,
                      uint32_t W, uint32_t ts, uint32_t col0, uint32_t col1, uint32_t* keys, uint32_t* values) {
    const uint32_t ntx = ceil_div_tiles(W, ts);
    const int full = (col0 == 0 && col1 >= ntx);
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        if (tile_counts[i] == 0) continue;
        const uint32_t* o = gdata + (size_t)i * GDATA_STRIDE_W;
        f32bits d;
        d.u = o[7];
        uint32_t bucket = f2u_sat(wminf(50.0f * d.f, 999.0f));
        uint32_t offs = offsets[i];
        for (uint32_t y = o[13]; y < o[15]; ++y)
            for (uint32_t x = o[12]; x < o[14]; ++x) {
                if (!full) {
                    uint32_t c = (x == ntx) ? 0u : x;
                    if (c < col0 || c >= col1) continue;
                }
                uint32_t tile_id = y * ntx + x;
                keys[offs] = tile_id * 1000u + bucket;
                values[offs] = (uint32_t)i;
                offs++;
            }
    }
}

/*
 * Stage A.4: sort.ts:341-350 + radix_sort.wgsl.  Stable ascending LSD radix sort, 8-bit digits,
 * 4 passes, result back in keys/values (even number of passes).  tmpk/tmpv: scratch of n words.
 */
GSO_API void gso_sort(uint32_t* keys, uint32_t* values, uint64_t n, uint32_t* tmpk, uint32_t* tmpv) {
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    if (n < 65536) nthreads = 1;
    uint64_t* hist = (uint64_t*)malloc((size_t)nthreads * 256 * sizeof(uint64_t));
    uint32_t *src_k = keys, *src_v = values, *dst_k = tmpk, *dst_v = tmpv;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = pass * 8;
        memset(hist, 0, (size_t)nthreads * 256 * sizeof(uint64_t));
#pragma omp parallel num_threads(nthreads)
        {
            int t = 0;
#ifdef _OPENMP
            t = omp_get_thread_num();
#endif
            uint64_t lo = n * (uint64_t)t / (uint64_t)nthreads, hi = n * (uint64_t)(t + 1) / (uint64_t)nthreads;
            uint64_t* h = hist + (size_t)t * 256;
            for (uint64_t i = lo; i < hi; ++i) h[(src_k[i] >> shift) & 255u]++;
#pragma omp barrier
#pragma omp single
            {
                uint64_t acc = 0;
                for (int d = 0; d < 256; ++d)
                    for (int tt = 0; tt < nthreads; ++tt) {
                        uint64_t c = hist[(size_t)tt * 256 + d];
                        hist[(size_t)tt * 256 + d] = acc;
                        acc += c;
                    }
            }
            for (uint64_t i = lo; i < hi; ++i) {
                uint64_t p = h[(src_k[i] >> shift) & 255u]++;
                dst_k[p] = src_k[i];
                dst_v[p] = src_v[i];
            }
        }
        uint32_t* sk = src_k; src_k = dst_k; dst_k = sk;
        uint32_t* sv = src_v; src_v = dst_v; dst_v = sv;
    }
    free(hist);
}

/* Stage A.5/A.6 (canonical): ranges[t] = |{ j < I : key_j/1000 <= t }| for t in [0,T). */
GSO_API void gso_ranges(const uint32_t* keys, uint64_t n, uint32_t T, uint32_t* ranges) {
    uint64_t j = 0;
    for (uint32_t t = 0; t < T; ++t) {
        while (j < n && keys[j] / 1000u <= t) ++j;
        ranges[t] = (uint32_t)j;
    }
}

/*
 * Stage A.7: compute_tiles.wgsl:30-75 + the rgba8unorm store.  No early exit (the reference has
 * none); `processed` (optional) receives, per tile, how many list entries are needed before
 * every pixel of the tile is finished under the exact criterion fl(T*fl(1-c255)) < 1e-4.
 * `illcond` (optional, one byte per pixel) flags pixels where some keep/skip decision lies
 * within the rounding-error margin of an alternative f32 evaluation order (see tests).
 */
GSO_API void gso_blend(const uint32_t* gdata, const uint32_t* values, const uint32_t* ranges, uint32_t W,
                       uint32_t H, uint32_t ts, uint32_t col0, uint32_t col1, uint8_t* rgba8, float* rgbf,
                       uint8_t* illcond, uint32_t* processed) {
    const uint32_t ntx = ceil_div_tiles(W, ts), nty = ceil_div_tiles(H, ts);
    const float c255 = (float)(1.0 / 255.0);
    const float one_minus_c = 1.0f - c255;
    if (col1 > ntx) col1 = ntx;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int32_t ty = 0; ty < (int32_t)nty; ++ty)
        for (int32_t tx = (int32_t)col0; tx < (int32_t)col1; ++tx) {
            uint32_t tile = (uint32_t)tx + (uint32_t)ty * ntx;
            uint32_t start = tile > 0 ? ranges[tile - 1] : 0u, end = ranges[tile];
            uint32_t tile_need = 0;
            for (uint32_t ly = 0; ly < ts; ++ly)
                for (uint32_t lx = 0; lx < ts; ++lx) {
                    uint32_t gx = (uint32_t)tx * ts + lx, gy = (uint32_t)ty * ts + ly;
                    if (gx >= W || gy >= H) continue;
                    float pxf = (float)gx, pyf = (float)gy;
                    float acc[3] = {0.f, 0.f, 0.f};
                    float t_i = 1.0f;
                    double t_relerr = 0.0; /* tracked relative uncertainty of t_i */
                    int ill = 0;
                    uint32_t need = 0;
                    int finished = 0;
                    for (uint32_t i = start; i < end && end > start; ++i) {
                        const uint32_t* o = gdata + (size_t)values[i] * GDATA_STRIDE_W;
                        f32bits q;
                        q.u = o[0]; float uvx = q.f;
                        q.u = o[1]; float uvy = q.f;
                        q.u = o[4]; float cx = q.f;
                        q.u = o[5]; float cy = q.f;
                        q.u = o[6]; float cz = q.f;
                        q.u = o[8]; float cr = q.f;
                        q.u = o[9]; float cg = q.f;
                        q.u = o[10]; float cb = q.f;
                        q.u = o[11]; float op = q.f;
                        float gxy0 = uvx * (float)W, gxy1 = uvy * (float)H;
                        float dx = gxy0 - pxf, dy = gxy1 - pyf;
                        float t1 = cx * dx * dx, t2 = cz * dy * dy, t3 = cy * dx * dy;
                        float power = -0.5f * (t1 + t2) - t3;
                        float alpha = wminf(0.99f, op * gso_expf(power));
                        float test_t = t_i * (1.0f - alpha);
                        int keep = (power <= 0.0f && alpha >= c255 && test_t >= 0.0001f);
                        if (illcond) {
                            /* margin of an alternative (fused / hw-exp2) f32 evaluation */
                            double eps = 6.0e-8;
                            double pe = 8.0 * eps * (fabs(0.5 * t1) + fabs(0.5 * t2) + fabs(t3)) + 1e-30;
                            double arel = pe + (fabs((double)power) + 8.0) * eps; /* relative uncertainty of alpha */
                            double trel = t_relerr + 4.0 * eps + (double)alpha / (1.0 - (double)alpha) * arel;
                            if (fabs((double)power) <= pe && op >= c255 * 0.5f) ill = 1;
                            if (power <= (float)pe && alpha < 0.99f &&
                                fabs((double)alpha - (double)c255) <= arel * (double)c255 * 2.0)
                                ill = 1;
                            if (power <= (float)pe && alpha >= c255 * 0.99f &&
                                fabs((double)test_t - 1e-4) <= trel * 1e-4 * 2.0)
                                ill = 1;
                            if (keep) t_relerr = trel;
                        }
                        float cond = keep ? 1.0f : 0.0f;
                        acc[0] += cond * cr * alpha * t_i;
                        acc[1] += cond * cg * alpha * t_i;
                        acc[2] += cond * cb * alpha * t_i;
                        t_i = cond * test_t + (1.0f - cond) * t_i;
                        if (!finished) {
                            need = i - start + 1;
                            if (t_i * one_minus_c < 0.0001f) finished = 1;
                        }
                    }
                    if (need > tile_need) tile_need = need;
                    size_t p = (size_t)gy * W + gx;
                    for (int c = 0; c < 3; ++c) {
                        float v = acc[c];
                        v = (v != v) ? 0.0f : wminf(wmaxf(v, 0.0f), 1.0f);
                        rgba8[p * 4 + c] = (uint8_t)floorf(v * 255.0f + 0.5f);
                        if (rgbf) rgbf[p * 3 + c] = acc[c];
                    }
                    rgba8[p * 4 + 3] = 255;
                    if (illcond) illcond[p] = (uint8_t)ill;
                }
            if (processed) processed[tile] = tile_need;
        }
}

/*
 * Checker for the product path's opacity-aware binning (DESIGN.md "tight binning"): for every instance
 * (key, value) of a key list, one bit per 8x8 pixel block of the instance's tile (row-major, ts/8 blocks per
 * tile row; a single bit at ts = 8) telling whether ANY in-canvas pixel of that block passes the reference's
 * `power <= 0 && alpha >= 1/255` test (compute_tiles.wgsl:57-63) under the canonical arithmetic of gso_blend.
 * An instance whose mask is 0 can never change a pixel (cond = 0 on every pixel), so a renderer may drop it
 * without changing any output bit.  Tiles >= T (rows/columns one past the grid) give mask 0:
 * compute_ranges ignores them.  Not part of the reference: it only exists to verify what the HIP path dropped.
 */
GSO_API void gso_instance_masks(const uint32_t* gdata, const uint32_t* keys, const uint32_t* values, uint64_t n,
                                uint32_t W, uint32_t H, uint32_t ts, uint32_t* masks) {
    const uint32_t ntx = ceil_div_tiles(W, ts), nty = ceil_div_tiles(H, ts);
    const uint32_t T = ntx * nty, bpr = ts / 8;
    const float c255 = (float)(1.0 / 255.0);
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const uint32_t tile = keys[i] / 1000u;
        uint32_t m = 0;
        if (tile < T) {
            const uint32_t tx = tile % ntx, ty = tile / ntx;
            const uint32_t* o = gdata + (size_t)values[i] * GDATA_STRIDE_W;
            f32bits q;
            q.u = o[0]; float uvx = q.f;
            q.u = o[1]; float uvy = q.f;
            q.u = o[4]; float cx = q.f;
            q.u = o[5]; float cy = q.f;
            q.u = o[6]; float cz = q.f;
            q.u = o[11]; float op = q.f;
            const float gxy0 = uvx * (float)W, gxy1 = uvy * (float)H;
            for (uint32_t ly = 0; ly < ts; ++ly)
                for (uint32_t lx = 0; lx < ts; ++lx) {
                    const uint32_t gx = tx * ts + lx, gy = ty * ts + ly;
                    if (gx >= W || gy >= H) continue;
                    const uint32_t bit = (ly / 8) * bpr + (lx / 8);
                    if (m & (1u << bit)) continue;
                    float dx = gxy0 - (float)gx, dy = gxy1 - (float)gy;
                    float t1 = cx * dx * dx, t2 = cz * dy * dy, t3 = cy * dx * dy;
                    float power = -0.5f * (t1 + t2) - t3;
                    float alpha = wminf(0.99f, op * gso_expf(power));
                    if (power <= 0.0f && alpha >= c255) m |= 1u << bit;
                }
        }
        masks[i] = m;
    }
}

GSO_API int gso_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
GSO_API void gso_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
