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
    float xmax, ymax;  // half extents of E in x and y, inflated
    float dyR, eR;     // dy of E's rightmost point (the leftmost is at -dyR) and the uncertainty of that estimate
    float kc, qa, qb;  // chord at offset dy: centre kc dy, half width sqrt(qa dy^2 + qb) / cx  (tight_chord)
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
    g.xmax = 0.0f; g.ymax = 0.0f; g.dyR = 0.0f; g.eR = 0.0f;
    // the chord's discriminant (cy dy)^2 - cx cz dy^2 + lim2 cx, inflated by 1e-5 of the magnitude of its terms (the rounding of
    // the products and of the difference is ~1e-7 of it), as a polynomial in dy^2: two instructions per chord instead of nine
    g.kc = -(cy * g.rcx);
    g.qa = (cyy - g.cxz) + 1.0e-5f * (cyy + g.cxz);
    g.qb = (g.lim2 * cx) * 1.00001f;
    const bool pd = (cx > 0.0f) && (cz > 0.0f) && (D > 0.0f);
    const bool fin = tight_finite(g.gx) && tight_finite(g.gy) && tight_finite(cx) && tight_finite(cy) && tight_finite(cz) && tight_finite(g.rcx);
    // non-finite geometry first: the reference's arithmetic turns such a splat's contribution into 0 * NaN = NaN on every pixel of
    // its rect's tiles whatever its opacity, so it keeps the whole rect (mode 2) even where its opacity alone would drop it
    if (!fin) { g.mode = 2u; return g; }
    if (lim < 0.0f) { g.mode = 0u; return g; }                                  // (-inf included; a NaN opacity falls through to mode 2)
    // Dlo: a lower bound of cx cz - cy^2 that also covers the inflation of tight_chord's discriminant (1e-5 of its terms), so
    // that xmax / ymax bound every chord tight_chord accepts: a row beyond ymax then yields nothing whether it is looked at or
    // skipped (tight_rows), and whole-canvas and slab frames agree instance for instance
    const float Dlo = D - 1.1e-5f * (g.cxz + cyy) - errD;
    if (!pd || !fin || !tight_finite(lim) || !(Dlo > 0.25f * D)) { g.mode = 2u; return g; } // non-PD conic, NaN/inf, or cx cz - cy^2 lost to cancellation
    g.xmax = __builtin_sqrtf(g.lim2 * 1.0001f * cz / Dlo) * 1.0002f + 0.02f;
    g.ymax = __builtin_sqrtf(g.lim2 * 1.0001f * cx / Dlo) * 1.0002f + 0.02f;
    g.dyR = -cy * g.xmax / cz;
    g.eR = __builtin_fabsf(g.dyR) * ((D - Dlo) / Dlo + 4.0e-4f) + 0.02f;
    g.mode = tight_finite(g.xmax) && tight_finite(g.ymax) && tight_finite(g.dyR) ? 1u : 2u;
    return g;
}

// Chord of E at vertical offset dy: [xlo, xhi] in dx; false if the line misses E.  The discriminant is inflated by its own
// rounding bound, so a chord is never missed or shortened by cancellation (cy^2 dy^2 against cx cz dy^2).
__device__ __forceinline__ bool tight_chord(const TightG& g, float dy, float& xlo, float& xhi) {
    const float up = __builtin_fmaf(dy * dy, g.qa, g.qb);
    if (!(up >= 0.0f)) return false;
    const float hw = __builtin_amdgcn_sqrtf(up) * g.rcx; // v_sqrt_f32 (1 ulp): `up` is inflated by 1e-5 of its terms, the strip adds 0.02 px of slack
    const float c = g.kc * dy;
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

// Columns of width w pixels (inv_w = 1/w, exact: w is a power of two; column c = pixel centres [c w, c w + w - 1]) that the
// pixel interval [plo, phi] touches, clamped to [cmin, cmax]; empty when lo > hi.
__device__ __forceinline__ void tight_cols(float plo, float phi, float w, float inv_w, int cmin, int cmax, int& lo, int& hi) {
    const float flo = __builtin_ceilf((plo - (w - 1.0f)) * inv_w), fhi = __builtin_floorf(phi * inv_w);
    lo = flo <= (float)cmin ? cmin : (flo > (float)cmax ? cmax + 1 : (int)flo);
    hi = fhi >= (float)cmax ? cmax : (fhi < (float)cmin ? cmin - 1 : (int)fhi);
}

struct TightChord { bool v; float lo, hi; };
__device__ __forceinline__ TightChord tight_chord_at(const TightG& g, float dy) {
    TightChord c;
    c.lo = c.hi = 0.0f;
    c.v = tight_chord(g, dy, c.lo, c.hi);
    return c;
}
// dy of the boundary between tile rows ty-1 and ty (pixel row ty*ts), the same float wherever it is computed
__device__ __forceinline__ float tight_row_dy(const TightG& g, uint32_t ty, uint32_t ts) { return g.gy - (float)(ty * ts); }

// Tile rows [ra, rb) of the rect's rows [y0, y1) that can hold an instance: the rows E's vertical extent reaches (plus the
// row above them when the rect has the aliased column, whose instance belongs to the NEXT row's column 0).  Mode 2: all rows.
__device__ __forceinline__ void tight_rows(const TightG& g, uint32_t y0, uint32_t y1, uint32_t ts, float inv_ts, uint32_t nty, uint32_t alias,
                                           uint32_t& ra, uint32_t& rb) {
    if (y1 > nty) y1 = nty; // rows past the grid never reach the blend (compute_ranges ignores tiles >= T)
    ra = y0; rb = y1 > y0 ? y1 : y0;
    if (g.mode != 1u) return;
    const float lo = __builtin_floorf((g.gy - g.ymax) * inv_ts) - (alias ? 1.0f : 0.0f), hi = __builtin_floorf((g.gy + g.ymax) * inv_ts) + 1.0f;
    if (lo > (float)ra) ra = lo >= (float)rb ? rb : (uint32_t)lo;
    if (hi < (float)rb) rb = hi <= (float)ra ? ra : (uint32_t)hi;
}

// The tiles of tile row ty that gaussian g is emitted to, for a rect whose slab-clipped real columns are [xa, xa + wmain)
// plus `alias` (the reference's column ntx, which lands in column 0 of the next row).  cb / ca: the chords at the
// row's upper and lower boundary (tight_row_dy(ty), tight_row_dy(ty + 1)).  Returns the instance count of the row;
// tlo..thi = its real tile columns (tlo > thi: none); r.alias = the aliased instance is kept.
struct TightRow { int tlo, thi; uint32_t alias; };
__device__ __forceinline__ uint32_t tight_row(const TightG& g, uint32_t ty, uint32_t ts, float inv_ts, uint32_t nty, uint32_t xa, uint32_t wmain,
                                              uint32_t alias, const TightChord& cb, const TightChord& ca, TightRow& r) {
    r.tlo = 0; r.thi = -1; r.alias = 0u;
    if (ty >= nty) return 0u;
    if (wmain) {
        if (g.mode == 2u) { r.tlo = (int)xa; r.thi = (int)(xa + wmain) - 1; }
        else {
            float plo, phi;
            if (tight_strip(g, tight_row_dy(g, ty + 1u, ts), tight_row_dy(g, ty, ts), ca.v, ca.lo, ca.hi, cb.v, cb.lo, cb.hi, plo, phi))
                tight_cols(plo, phi, (float)ts, inv_ts, (int)xa, (int)(xa + wmain) - 1, r.tlo, r.thi);
            else { r.tlo = (int)xa; r.thi = (int)xa - 1; }
        }
    }
    // the aliased instance of row ty IS tile (ty + 1, 0): kept if that tile exists and intersects E -- at the far side of the
    // screen from the gaussian, so almost never
    if (alias && ty + 1u < nty) {
        if (g.mode == 2u) r.alias = 1u;
        else {
            const TightChord c2 = tight_chord_at(g, tight_row_dy(g, ty + 2u, ts));
            float plo, phi;
            int lo = 1, hi = 0;
            if (tight_strip(g, tight_row_dy(g, ty + 2u, ts), tight_row_dy(g, ty + 1u, ts), c2.v, c2.lo, c2.hi, ca.v, ca.lo, ca.hi, plo, phi))
                tight_cols(plo, phi, (float)ts, inv_ts, 0, 0, lo, hi);
            r.alias = lo <= hi ? 1u : 0u;
        }
    }
    return (uint32_t)(r.thi >= r.tlo ? r.thi - r.tlo + 1 : 0) + r.alias;
}

// Tile count of a gaussian under tight binning: rows [y0, y1) of its rect (the reference's rminy .. rmaxy).  Consecutive rows
// share a boundary, so one chord (one sqrt) per row.
__device__ __forceinline__ uint32_t tight_count(const TightG& g, uint32_t y0, uint32_t y1, uint32_t ts, float inv_ts, uint32_t nty, uint32_t xa,
                                                uint32_t wmain, uint32_t alias) {
    if (g.mode == 0u) return 0u;
    uint32_t ra, rb, n = 0;
    tight_rows(g, y0, y1, ts, inv_ts, nty, alias, ra, rb);
    if (ra >= rb) return 0u;
    TightRow r;
    TightChord cb = tight_chord_at(g, tight_row_dy(g, ra, ts));
    for (uint32_t ty = ra; ty < rb; ++ty) {
        const TightChord ca = tight_chord_at(g, tight_row_dy(g, ty + 1u, ts));
        n += tight_row(g, ty, ts, inv_ts, nty, xa, wmain, alias, cb, ca, r);
        cb = ca;
    }
    return n;
}

// Sub-block columns (width sub = tile_size/2 pixels, or the whole 8-pixel tile) touched in the upper (s = 0) and lower
// (s = 1) half strip of tile row ty; empty: lo > hi.  cb / ca as in tight_row.  Only the masks use this, never a count.
__device__ __forceinline__ void tight_substrips(const TightG& g, uint32_t ty, uint32_t ts, uint32_t sub, float inv_sub, int cmin, int cmax,
                                                const TightChord& cb, const TightChord& ca, int lo[2], int hi[2]) {
    lo[0] = lo[1] = cmax + 1; hi[0] = hi[1] = cmin - 1;
    if (g.mode == 2u) { lo[0] = lo[1] = cmin; hi[0] = hi[1] = cmax; return; }
    const float yb = tight_row_dy(g, ty, ts), ya = tight_row_dy(g, ty + 1u, ts);
    float plo, phi;
    if (sub == ts) { // tile 8: the mask is the tile itself
        if (tight_strip(g, ya, yb, ca.v, ca.lo, ca.hi, cb.v, cb.lo, cb.hi, plo, phi)) tight_cols(plo, phi, (float)sub, inv_sub, cmin, cmax, lo[0], hi[0]);
        return;
    }
    const float ym = g.gy - (float)(ty * ts + sub);
    const TightChord cm = tight_chord_at(g, ym);
    if (tight_strip(g, ym, yb, cm.v, cm.lo, cm.hi, cb.v, cb.lo, cb.hi, plo, phi)) tight_cols(plo, phi, (float)sub, inv_sub, cmin, cmax, lo[0], hi[0]);
    if (tight_strip(g, ya, ym, ca.v, ca.lo, ca.hi, cm.v, cm.lo, cm.hi, plo, phi)) tight_cols(plo, phi, (float)sub, inv_sub, cmin, cmax, lo[1], hi[1]);
}

// ---- row items (the unit of the tight row pipeline, k_rows.hip) ----------------------------------------------------------
// A gaussian's instances are, per tile row of its rect, ONE run of tiles (E ∩ strip is convex) plus at most one aliased
// instance.  The projection computes each run once, with the sub-block intervals of its two half strips, and stores it as a
// 12-byte ROW ITEM; everything downstream (sorting the items by tile row, expanding them into per-tile lists) is integer work
// on these records -- round 2's emission recomputed the chords of every row from the 64-byte GaussianData record
// (295 lane-instructions per instance, instruction bound at 0.147 of the HBM roofline).
//   w0  gaussian id (28 bits); 0xFFFFFFFF = hole (a slot whose row turned out empty)
//   w1  tile row | first tile column << 8 | (tiles - 1) << 16                     (each < 256: tight binning needs ntx, nty <= 255)
//   w2  l0 | h0 << 8 | l1 << 16 | h1 << 24: sub-block columns [l, h) touched in the upper / lower half strip, relative to the
//       run's first sub-block column (sub-block = tile_size / 2, or the whole 8-pixel tile: then only l0, h0); h = 255 = no end
// Slots of a gaussian whose rows are [ra, rb): slot s < rb - ra is the run of row ra + s; with an aliased column (SURVEY A.3:
// column ntx of row ty IS tile (ty + 1, 0)) slot (rb - ra) + s is the aliased instance of row ra + s, an item of its own in
// tile row ty + 1, column 0, one tile long.
#define GS_ROW_HOLE 0xFFFFFFFFu
#define GS_ROW_MAX_DIM 255u
__device__ __forceinline__ uint32_t tight_enc8(int v) { return v <= 0 ? 0u : (v >= 255 ? 255u : (uint32_t)v); }
__device__ __forceinline__ uint32_t tight_pack_intervals(const int lo[2], const int hi[2], int cmin, uint32_t ncols) {
    if (ncols > 254u) return 0xFF00FF00u; // a run too long for 8-bit intervals: every sub-block flagged (conservative)
    uint32_t w = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        uint32_t l = 0, h = 0;
        if (lo[s] <= hi[s]) { l = tight_enc8(lo[s] - cmin); h = tight_enc8(hi[s] + 1 - cmin); }
        w |= (l | (h << 8)) << (16 * s);
    }
    return w;
}
// Slot `slot` of a gaussian (see above).  Returns the number of tiles of the item (0: hole) and its words w1, w2.
__device__ __forceinline__ uint32_t tight_slot_item(const TightG& g, uint32_t slot, uint32_t ra, uint32_t nrows, uint32_t ts, float inv_ts, uint32_t sub,
                                                    float inv_sub, uint32_t ns, uint32_t nty, uint32_t xa, uint32_t wmain, uint32_t alias, uint32_t& w1,
                                                    uint32_t& w2) {
    w1 = 0u; w2 = 0u;
    TightRow r;
    int lo[2], hi[2];
    if (slot < nrows) {
        const uint32_t ty = ra + slot;
        if (ty >= nty || !wmain) return 0u;
        // the two half strips of the tile row in sub-block columns of the (slab-clipped) rect; the row's run of tiles is the hull of
        // the two intervals (E ∩ row strip is the union of its halves): no third strip, and one chord less than strip by strip
        const int cmin = (int)(xa * ns), cmax = (int)((xa + wmain) * ns) - 1;
        const TightChord cb = tight_chord_at(g, tight_row_dy(g, ty, ts)), ca = tight_chord_at(g, tight_row_dy(g, ty + 1u, ts));
        tight_substrips(g, ty, ts, sub, inv_sub, cmin, cmax, cb, ca, lo, hi);
        int slo = cmax + 1, shi = cmin - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (lo[h] <= hi[h]) { slo = lo[h] < slo ? lo[h] : slo; shi = hi[h] > shi ? hi[h] : shi; }
        if (slo > shi) return 0u;
        const uint32_t tlo = (uint32_t)slo / ns, thi = (uint32_t)shi / ns, len = thi - tlo + 1u;
        w1 = ty | (tlo << 8) | ((len - 1u) << 16);
        w2 = tight_pack_intervals(lo, hi, (int)(tlo * ns), len * ns);
        return len;
    }
    if (!alias) return 0u;
    const uint32_t ty = ra + (slot - nrows);
    const TightChord cb = tight_chord_at(g, tight_row_dy(g, ty, ts)), ca = tight_chord_at(g, tight_row_dy(g, ty + 1u, ts));
    if (!tight_row(g, ty, ts, inv_ts, nty, xa, 0u, 1u, cb, ca, r)) return 0u;
    const TightChord c2 = tight_chord_at(g, tight_row_dy(g, ty + 2u, ts));
    tight_substrips(g, ty + 1u, ts, sub, inv_sub, 0, (int)ns - 1, ca, c2, lo, hi);
    w1 = (ty + 1u); // tile (ty + 1, 0), one tile
    w2 = tight_pack_intervals(lo, hi, 0, ns);
    return 1u;
}
// Sub-block mask of tile q (0-based inside the run) of an item: bit 0 / 1 = left / right sub-block of the upper half strip,
// bits 2 / 3 of the lower one (ns = 2); bit 0 = the tile (ns = 1).
__device__ __forceinline__ uint32_t tight_item_mask(uint32_t w2, uint32_t q, uint32_t ns) {
    const uint32_t l0 = w2 & 0xFFu, h0 = (w2 >> 8) & 0xFFu, l1 = (w2 >> 16) & 0xFFu, h1 = w2 >> 24;
    if (ns == 2u) {
        const uint32_t c0 = 2u * q, c1 = c0 + 1u;
        return (uint32_t)(l0 <= c0 && (c0 < h0 || h0 == 255u)) | ((uint32_t)(l0 <= c1 && (c1 < h0 || h0 == 255u)) << 1) |
               ((uint32_t)(l1 <= c0 && (c0 < h1 || h1 == 255u)) << 2) | ((uint32_t)(l1 <= c1 && (c1 < h1 || h1 == 255u)) << 3);
    }
    return (uint32_t)(l0 <= q && (q < h0 || h0 == 255u));
}
