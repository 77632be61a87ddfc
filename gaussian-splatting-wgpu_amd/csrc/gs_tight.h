// gs_tight.h -- opacity-aware ("tight") binning of the product path (gs_render / gs_render_to with GS_OPT_TILE_CULL 1).
//
// The reference bins a gaussian into EVERY tile of the square of half-width ceil(3 sigma_max) around its centre
// (process_gaussians.wgsl:74-86,297-319; write_tile_ids.wgsl:26-35).  Its blend then skips the gaussian on a pixel unless
//     power <= 0  &&  alpha = min(0.99, opacity * exp(power)) >= 1/255          (compute_tiles.wgsl:57-63)
// i.e. unless  q(d) = 0.5 (cx dx^2 + cz dy^2) + cy dx dy  <=  ln(255 opacity),  d = centre - pixel.
// On the benchmark scene 58 % of the reference's (gaussian, tile) instances fail that test on every pixel of their tile:
// they are emitted, sorted twice, ranged, gathered and culled for nothing.  Here an instance is emitted only if its
// tile intersects the ellipse  E = { q <= ln(255 opacity) + margin }  and it carries a mask of the sub-blocks of the tile
// that intersect E, so the blend's walkers neither fetch nor test what cannot touch their pixels.
//
// Every test is CONSERVATIVE (margins far above f32 rounding; anything doubtful falls back to the reference's rect), so a
// dropped instance has cond = 0 on every pixel in the reference's arithmetic and no output bit changes -- checked
// against the oracle by tests/gpu_checks.py (oracle.instance_masks).  gs_render_debug never comes here.
//
// Geometry.  For a horizontal strip of pixel rows [y0, y1] the set E ∩ strip is convex, so its x-extent is an interval:
// [min over the strip of the chord's left end, max of its right end].  The right end x_hi(dy) is concave in dy, so its
// maximum over the strip is at the rightmost point of the whole ellipse if that point's dy lies in the strip, else at the
// strip boundary nearer to it; likewise on the left.  One chord (one sqrt) per strip boundary.
//
// The tile count of the projection and the emission must agree EXACTLY, so both call the same functions below on the same
// inputs (the 64-byte GaussianData record); the library is compiled with -ffp-contract=off, IEEE division and sqrt, and
// v_log_f32 is the same instruction everywhere, so the results are bit-identical wherever they are evaluated.
#pragma once
#include "gs_device.h"

#define GS_ID_BITS 28                // sorted values of a tight frame: gaussian id | sub-block mask << 28
#define GS_ID_MASK 0x0FFFFFFFu

struct TightG {
    float gx, gy;      // centre in pixels (compute_tiles.wgsl:52: uv * canvas size)
    float cx, cy, cz;  // conic
    float cxz;         // cx * cz
    float lim2;        // 2 (ln(255 opacity) + margin): the pixel can pass only if cx dx^2 + 2 cy dx dy + cz dy^2 <= lim2
    float rcx;         // 1 / cx
    float xmax;        // half extent of E in x, inflated
    float dyR, eR;     // dy of E's rightmost point (the leftmost is at -dyR) and the uncertainty of that estimate
    uint32_t mode;     // 0: cannot pass anywhere (opacity < 1/255), 1: ellipse test, 2: keep the reference's whole rect
};

__device__ __forceinline__ bool tight_finite(float x) { return __builtin_fabsf(x) < 3.0e38f; } // false for NaN and inf

__device__ __forceinline__ TightG tight_setup(float uvx, float uvy, float cx, float cy, float cz, float op, float Wf, float Hf) {
    TightG g;
    g.gx = uvx * Wf;
    g.gy = uvy * Hf;
    g.cx = cx; g.cy = cy; g.cz = cz;
    g.cxz = cx * cz;
    const float cyy = cy * cy;
    const float D = g.cxz - cyy;
    const float errD = 4.0e-7f * (g.cxz + cyy); // rounding of the two products and the difference, with headroom
    // alpha >= 1/255  <=>  q <= ln(255 op); +0.01 keeps the test conservative (the reference's rounding is ~1e-6 relative)
    const float lim = __builtin_amdgcn_logf(op * 255.0f) * 0.693147182464599609375f + 0.01f; // v_log_f32 (log2) * ln 2
    g.lim2 = 2.0f * lim;
    g.rcx = 1.0f / cx;
    g.xmax = 0.0f; g.dyR = 0.0f; g.eR = 0.0f;
    const bool pd = (cx > 0.0f) && (cz > 0.0f) && (D > 0.0f);
    const bool fin = tight_finite(g.gx) && tight_finite(g.gy) && tight_finite(cx) && tight_finite(cy) && tight_finite(cz) && tight_finite(g.rcx);
    if (lim < 0.0f) { g.mode = 0u; return g; }                                  // (-inf included; NaN falls through to mode 2)
    if (!pd || !fin || !tight_finite(lim) || !(D > 4.0f * errD)) { g.mode = 2u; return g; } // non-PD conic, NaN/inf, or cx cz - cy^2 lost to cancellation
    const float Dlo = D - errD;
    g.xmax = __builtin_sqrtf(g.lim2 * cz / Dlo) * 1.0002f + 0.02f;
    g.dyR = -cy * g.xmax / cz;
    g.eR = __builtin_fabsf(g.dyR) * (errD / Dlo + 4.0e-4f) + 0.02f;
    g.mode = tight_finite(g.xmax) && tight_finite(g.dyR) ? 1u : 2u;
    return g;
}

// Chord of E at vertical offset dy: [xlo, xhi] in dx; false if the line misses E.  The discriminant is inflated by its own
// rounding bound, so a chord is never missed or shortened by cancellation (cy^2 dy^2 against cx cz dy^2).
__device__ __forceinline__ bool tight_chord(const TightG& g, float dy, float& xlo, float& xhi) {
    const float t = g.cy * dy;
    const float p1 = t * t, p2 = g.cxz * (dy * dy), p3 = g.lim2 * g.cx;
    const float up = ((p1 - p2) + p3) + 1.0e-5f * ((p1 + p2) + p3);
    if (!(up >= 0.0f)) return false;
    const float hw = __builtin_sqrtf(up) * g.rcx;
    const float c = -(t * g.rcx);
    xlo = c - hw;
    xhi = c + hw;
    return true;
}

// x-extent, in PIXEL coordinates (px = gx - dx), of E over the strip of pixel rows whose dy = gy - py lies in [a, b];
// (va, lo_a, hi_a) / (vb, lo_b, hi_b) are the chords at a and b.  false: the strip misses E.
__device__ __forceinline__ bool tight_strip(const TightG& g, float a, float b, bool va, float lo_a, float hi_a, bool vb, float lo_b,
                                            float hi_b, float& plo, float& phi) {
    float xlo, xhi;
    if (!va && !vb) {
        if (!(a <= 0.0f && 0.0f <= b)) return false; // both boundaries miss E and its centre row is not between them
        xlo = -g.xmax;
        xhi = g.xmax;
    } else {
        xlo = va ? (vb ? __builtin_fminf(lo_a, lo_b) : lo_a) : lo_b;
        xhi = va ? (vb ? __builtin_fmaxf(hi_a, hi_b) : hi_a) : hi_b;
        if (g.dyR >= a - g.eR && g.dyR <= b + g.eR) xhi = g.xmax;   // the rightmost point of E may lie inside the strip
        if (-g.dyR >= a - g.eR && -g.dyR <= b + g.eR) xlo = -g.xmax; // ... the leftmost
    }
    const float slack = 0.02f + 1.0e-5f * (__builtin_fabsf(xlo) + __builtin_fabsf(xhi) + __builtin_fabsf(g.gx));
    plo = g.gx - xhi - slack;
    phi = g.gx - xlo + slack;
    return true;
}

// Columns of width `w` pixels (column c = pixel centres [c w, c w + w - 1]) that the pixel interval [plo, phi] touches,
// clamped to [cmin, cmax]; empty when lo > hi.
__device__ __forceinline__ void tight_cols(float plo, float phi, float w, int cmin, int cmax, int& lo, int& hi) {
    const float inv = 1.0f / w;
    const float flo = __builtin_ceilf((plo - (w - 1.0f)) * inv), fhi = __builtin_floorf(phi * inv);
    lo = flo <= (float)cmin ? cmin : (flo > (float)cmax ? cmax + 1 : (int)flo);
    hi = fhi >= (float)cmax ? cmax : (fhi < (float)cmin ? cmin - 1 : (int)fhi);
}

// The tiles of tile row ty that gaussian g is emitted to, for a rect whose slab-clipped columns are [xa, xa + wmain)
// (slab_cols: real columns only) plus `alias` (the reference's column ntx, which lands in column 0 of the next row).
// Returns the instance count of the row; tlo..thi = its real tile columns (tlo > thi: none); r.alias = the aliased one is kept.
struct TightRow { int tlo, thi; uint32_t alias; };
// Tile columns [cmin, cmax] of tile row ty that intersect E (mode 1); empty: lo > hi.
__device__ __forceinline__ void tight_row_cols(const TightG& g, uint32_t ty, uint32_t ts, int cmin, int cmax, int& lo, int& hi) {
    lo = cmax + 1; hi = cmin - 1;
    const float b = g.gy - (float)(ty * ts), a = g.gy - (float)((ty + 1u) * ts); // pixel rows [ty ts, (ty+1) ts], continuous
    float lo_a, hi_a, lo_b, hi_b, plo, phi;
    const bool va = tight_chord(g, a, lo_a, hi_a), vb = tight_chord(g, b, lo_b, hi_b);
    if (tight_strip(g, a, b, va, lo_a, hi_a, vb, lo_b, hi_b, plo, phi)) tight_cols(plo, phi, (float)ts, cmin, cmax, lo, hi);
}
__device__ __forceinline__ uint32_t tight_row(const TightG& g, uint32_t ty, uint32_t ts, uint32_t nty, uint32_t xa, uint32_t wmain,
                                              uint32_t alias, TightRow& r) {
    r.tlo = 0; r.thi = -1; r.alias = 0u;
    if (ty >= nty) return 0u; // rows past the grid never reach the blend (compute_ranges ignores tiles >= T)
    if (wmain) {
        if (g.mode == 2u) { r.tlo = (int)xa; r.thi = (int)(xa + wmain) - 1; }
        else tight_row_cols(g, ty, ts, (int)xa, (int)(xa + wmain) - 1, r.tlo, r.thi);
    }
    // the aliased instance of row ty (the reference's column ntx) IS tile (ty + 1, 0): kept if that tile exists and
    // intersects E -- at the far side of the screen from the gaussian, so almost never
    if (alias && ty + 1u < nty) {
        if (g.mode == 2u) r.alias = 1u;
        else {
            int lo, hi;
            tight_row_cols(g, ty + 1u, ts, 0, 0, lo, hi);
            r.alias = lo <= hi ? 1u : 0u;
        }
    }
    return (uint32_t)(r.thi >= r.tlo ? r.thi - r.tlo + 1 : 0) + r.alias;
}

// Sub-block columns (width sub = tile_size/2 pixels, or the whole 8-pixel tile) touched in the upper (s = 0) and lower
// (s = 1) half strip of tile row ty; empty: lo > hi.  Only used for the masks, never for counts.
__device__ __forceinline__ void tight_substrips(const TightG& g, uint32_t ty, uint32_t ts, uint32_t sub, int cmin, int cmax, int lo[2],
                                                int hi[2]) {
    const uint32_t ns = ts / sub; // 1 or 2
    lo[0] = lo[1] = cmax + 1; hi[0] = hi[1] = cmin - 1;
    if (g.mode == 2u) { lo[0] = lo[1] = cmin; hi[0] = hi[1] = cmax; return; }
    float yb = g.gy - (float)(ty * ts);
    float lo_b, hi_b;
    bool vb = tight_chord(g, yb, lo_b, hi_b);
    for (uint32_t s = 0; s < ns; ++s) {
        const float ya = g.gy - (float)(ty * ts + (s + 1u) * sub);
        float lo_a, hi_a, plo, phi;
        const bool va = tight_chord(g, ya, lo_a, hi_a);
        if (tight_strip(g, ya, yb, va, lo_a, hi_a, vb, lo_b, hi_b, plo, phi)) tight_cols(plo, phi, (float)sub, cmin, cmax, lo[s], hi[s]);
        yb = ya; vb = va; lo_b = lo_a; hi_b = hi_a;
    }
}

// Tile count of a gaussian under tight binning: rows [y0, y1) of its rect (the reference's rminy .. rmaxy).
__device__ __forceinline__ uint32_t tight_count(const TightG& g, uint32_t y0, uint32_t y1, uint32_t ts, uint32_t nty, uint32_t xa,
                                                uint32_t wmain, uint32_t alias) {
    if (g.mode == 0u) return 0u;
    uint32_t n = 0;
    if (y1 > nty) y1 = nty;
    TightRow r;
    for (uint32_t ty = y0; ty < y1; ++ty) n += tight_row(g, ty, ts, nty, xa, wmain, alias, r);
    return n;
}
