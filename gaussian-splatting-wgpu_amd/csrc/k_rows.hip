// k_rows.hip -- the tight row pipeline: from the projection's row items to the per-tile instance lists the blend walks.
//
// Replaces, on the product path (gs_render / gs_render_to with tight binning), write_tile_ids.wgsl::main (reference
// src/write_tile_ids.wgsl:18-35), the instance passes of GPUSorter (src/radix_sort/sort.ts:249-350, radix_sort.wgsl:256-449) and
// compute_ranges.wgsl::main (src/compute_ranges.wgsl:5-29).  gs_render_debug keeps the reference's stages (k_binning.hip,
// k_sort.hip): every reference tap exists there.
//
// Round 2 emitted one (u16 tile, u32 value) pair per instance in depth order (instruction bound: the emission recomputed every
// row's chords), moved all 18.4 M pairs through two look-back radix sweeps (latency bound) and read the sorted keys once more
// for the ranges: 4 launches, 364 us at config B.  The observation behind this file: a gaussian's instances are, per tile row,
// ONE run of consecutive tiles, so the unit that has to be ordered by tile ROW is the row item (7 M, not 18.4 M), and inside
// one tile row the column is a digit of at most 8 bits.  Hence (MSD first):
//   gs_rows_sort_kernel    the row items, gathered in depth order (through the gaussian-level sort's records), are
//                          partitioned STABLY by tile row: one Onesweep-style pass (ticketed tiles, 4-byte look-back granules),
//                          digit histogram = GsControl::rowhist, which the projection accumulated.
//   gs_rows_count_kernel   chunks of 512 consecutive items of ONE tile row count their instances per tile column (difference
//                          array + scan) -> M3[chunk][column].
//   gs_rows_scan_kernel    per tile row: prefix of M3 over the row's chunks (so no chunk ever waits for another), tile totals,
//                          their prefix inside the row, the row's total.
//   gs_rows_expand_kernel  every chunk expands its items into instances (value = gaussian id | sub-block mask << 28), ranks
//                          them stably by column (wave ballots), reorders them through LDS and stores each column's run at
//                          tile start + earlier chunks' count: the FINAL lists, written once, 4 bytes per instance; no key
//                          is ever materialised, and `ranges` falls out of the scan.
// Order inside a tile = item order inside the tile row = depth order of the gaussians = (bucket, index): the reference's.
// Every kernel is integer work on 12-byte items and 4-byte values; algorithmic bytes per frame (R items, I instances):
// 12 R read + 12 R written by the row sort, 12 R read twice by count / expand, 4 I written.
#include "gs_device.h"
#include "gs_tight.h"

typedef uint32_t gs_item3 __attribute__((ext_vector_type(3), aligned(4)));

#ifndef RA_ITEMS
#define RA_ITEMS 8 // measured (config B): 6: 101-105 us, 8: 98, 10: 96 (two workgroups per CU), 12: 111
#endif
#ifndef RA_WAVES
#define RA_WAVES 8
#endif
#define RA_THREADS (RA_WAVES * 64)
#define RA_TILE (RA_THREADS * RA_ITEMS) // 4096 slots per tile (8 waves x 8 items per lane): 48 KB of LDS for the reorder, three workgroups (24 waves) per CU
#define RA_AGG (1u << 30)
#define RA_PREFIX (2u << 30)
#define RA_FLAGS (3u << 30)
#define RA_VALUE (~RA_FLAGS)

#ifndef RB_CH
#define RB_CH 512u   // items per chunk
#endif
#define RB_IT (RB_CH / 256u)
#ifndef RB_SB
#define RB_SB 2048u
#endif
// RB_SB: instances per sub-batch of the expansion (256 threads x 8): 25 KB of LDS, six workgroups per CU
#define RB_PER (RB_SB / 256u)
#define RB_WSL (RB_SB / 4u) // slots of one wave

// ------------------------------------------------------------------------------------------------
// Row sort: stable partition of the row items by tile row.
// ------------------------------------------------------------------------------------------------
struct RowSortShared {
    uint32_t hist[RA_WAVES][256]; // per-wave digit counts -> exclusive offsets across waves -> + digit start
    uint32_t gbase[256];          // global slot of position 0 of each digit's run, minus the digit's first position
    uint32_t tot[256];
    uint32_t dbase[256];          // exclusive scan of GsControl::rowhist
    uint32_t wsum[RA_WAVES];
    uint32_t tile, nvalid;
    union {
        uint32_t it[RA_TILE * 3];                 // the reorder buffer
        struct {                                  // before it: the gaussians whose slots the tile holds (first slot, arena address)
            uint32_t goff[RA_TILE + 8], gptr[RA_TILE + 8];
            unsigned short mark[RA_TILE];         // mark[s] = k + 1: gaussian k's first slot is tile slot s
        } g;
    } u;
};

// grec / chunk_table: the gaussian-level sort's records {id, slots | bucket << 22, first slot, arena address} in depth order and
// the gaussian holding every 1024th slot (k_gsort.hip).  Slot s of the depth-ordered slot sequence belongs to the gaussian k
// with first_slot[k] <= s < first_slot[k + 1] and lives at arena[address[k] + s - first_slot[k]]: every gaussian of the tile
// marks its first slot in LDS and a running maximum (DPP) hands every slot its owner.
__global__ __launch_bounds__(RA_THREADS) void gs_rows_sort_kernel(const uint32_t* __restrict__ arena, const uint4* __restrict__ grec,
                                                                  const uint32_t* __restrict__ chunk_table, uint32_t* __restrict__ rows_out,
                                                                  GsControl* ctl, uint32_t* __restrict__ status, uint32_t row_cap, uint32_t ndig) {
    // ndig: tile rows of the canvas = digits that exist; holes take digit `hole` (127 when the rows fit 7 bits: one ballot less)
    __shared__ RowSortShared sh;
    const uint32_t hole = ndig < 128u ? 127u : 255u;
    const int nbits = ndig < 128u ? 7 : 8;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t n = ctl->num_slots;
    if (n > row_cap) n = row_cap;
    const uint32_t ntiles = (n + RA_TILE - 1) / RA_TILE;
    const uint32_t nvis = ctl->num_visible;
    {   // first slot of every tile row's run: exclusive scan of the digit histogram (every workgroup for itself)
        const uint32_t c = tid < 256u ? gs_rowhist(ctl, tid) : 0u;
        const uint32_t incl = wave_incl_scan(c, lane);
        if (lane == 63) sh.wsum[w] = incl;
        __syncthreads();
        uint32_t b = 0;
        for (uint32_t k = 0; k < w; ++k) b += sh.wsum[k];
        if (tid < 256u) sh.dbase[tid] = b + incl - c;
        __syncthreads();
    }
    for (;;) {
        if (tid == 0) sh.tile = atomicAdd(&ctl->rows_ticket, 1u);
        __syncthreads();
        const uint32_t tile = sh.tile;
        if (tile >= ntiles) break; // uniform
        const uint32_t s0 = tile * RA_TILE, s_end = (s0 + RA_TILE < n) ? s0 + RA_TILE : n;
        // the gaussians g0 .. g1 hold the tile's slots (g1 holds the first slot of the next tile, or is the last one)
        const uint32_t g0 = chunk_table[s0 >> 10];
        const uint32_t g1 = (s_end < n) ? chunk_table[s_end >> 10] : nvis - 1u;
        uint32_t K = (g1 >= g0) ? g1 - g0 + 1u : 1u;
        if (K > RA_TILE + 8u) K = RA_TILE + 8u; // (cannot happen: every gaussian here has at least one slot)
        for (uint32_t k = tid; k < RA_TILE / 2u; k += RA_THREADS) reinterpret_cast<uint32_t*>(sh.u.g.mark)[k] = 0u;
        __syncthreads();
        for (uint32_t k = tid; k < K; k += RA_THREADS) {
            const uint4 rec = (g0 + k < nvis) ? grec[g0 + k] : make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);
            sh.u.g.goff[k] = rec.z;
            sh.u.g.gptr[k] = rec.w;
            if (rec.z >= s0 && rec.z < s_end) sh.u.g.mark[rec.z - s0] = (unsigned short)(k + 1u);
        }
        __syncthreads();
        uint32_t carry;
        {   // owner of the wave's first slot: the last gaussian whose first slot is <= it.  goff is increasing: 64 samples RA_STR apart
            // find the segment, its RA_STR entries the owner -- two LDS round trips instead of the twelve of a binary search
            constexpr uint32_t RA_STR = (RA_TILE + 8u + 63u) / 64u; // 64 samples cover every k < K <= RA_TILE + 8
            static_assert(RA_STR * 64u >= RA_TILE + 8u && RA_STR <= 128u, "the two-level owner search covers the tile's gaussians");
            const uint32_t x0 = s0 + w * (64 * RA_ITEMS);
            const uint32_t m1 = lane * RA_STR;
            const uint32_t c1 = (uint32_t)__popcll(__ballot(m1 < K && sh.u.g.goff[m1 < K ? m1 : 0u] <= x0)); // >= 1: goff[0] <= s0 <= x0
            const uint32_t b1 = (c1 ? c1 - 1u : 0u) * RA_STR;
            uint32_t c2 = 0;
#pragma unroll
            for (uint32_t t2 = 0; t2 < (RA_STR + 63u) / 64u; ++t2) {
                const uint32_t o2 = t2 * 64u + lane, m2 = b1 + o2;
                c2 += (uint32_t)__popcll(__ballot(o2 < RA_STR && m2 < K && sh.u.g.goff[m2 < K ? m2 : 0u] <= x0));
            }
            const uint32_t e = b1 + (c2 ? c2 - 1u : 0u);
            carry = e + 1u;
        }
        uint32_t ix[RA_ITEMS], iy[RA_ITEMS], iz[RA_ITEMS];
        uint32_t rank2[RA_ITEMS / 2];
#pragma unroll
        for (int j = 0; j < RA_ITEMS; ++j) {
            const uint32_t sp = w * (64 * RA_ITEMS) + j * 64 + lane, slot = s0 + sp;
            uint32_t m = wave_incl_max((uint32_t)sh.u.g.mark[sp]);
            m = m > carry ? m : carry;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
            ix[j] = GS_ROW_HOLE; iy[j] = 0u; iz[j] = 0u;
            if (slot < n) {
                const uint32_t k = m - 1u;
                const uint32_t src = sh.u.g.gptr[k] + (slot - sh.u.g.goff[k]);
                if (src < row_cap) {
                    const gs_item3 v = *reinterpret_cast<const gs_item3*>(arena + (uint64_t)src * 3u);
                    ix[j] = v.x; iy[j] = v.y; iz[j] = v.z;
                }
            }
        }
        for (uint32_t k = lane; k < 256; k += 64) sh.hist[w][k] = 0;
        if (tid < 256u) sh.tot[tid] = 0u;
        __syncthreads();
        // holes (and the slots past the end) take a digit no tile row has: they rank last and are not stored
#pragma unroll
        for (int j = 0; j < RA_ITEMS; ++j) atomicAdd(&sh.tot[ix[j] == GS_ROW_HOLE ? hole : (iy[j] & 0xFFu)], 1u);
        __syncthreads();
        // the tile's digit counts are published BEFORE the ranking: successors rarely meet an unpublished word
        if (tid < ndig) st_agent(status + (uint64_t)tile * 256 + tid, (tile == 0 ? RA_PREFIX : RA_AGG) | sh.tot[tid]);
        // rank inside the wave: peers = lanes holding the same digit (8 ballots), order = (item, lane)
#pragma unroll
        for (int j = 0; j < RA_ITEMS; ++j) {
            const uint32_t d = ix[j] == GS_ROW_HOLE ? hole : (iy[j] & 0xFFu);
            uint32_t plo = 0xFFFFFFFFu, phi = 0xFFFFFFFFu;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b < nbits) {
                    const uint32_t bit = (d >> b) & 1u;
                    const unsigned long long bal = __ballot(bit != 0u);
                    const uint32_t inv = bit - 1u;
                    plo &= (uint32_t)bal ^ inv;
                    phi &= (uint32_t)(bal >> 32) ^ inv;
                }
            }
            const uint32_t below = __popc(plo & (uint32_t)lt_mask) + __popc(phi & (uint32_t)(lt_mask >> 32));
            const uint32_t cnt = __popc(plo) + __popc(phi);
            const uint32_t pre = sh.hist[w][d];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // every peer has read `pre` before the leader's store (one wave, in-order LDS)
            if (below == 0) sh.hist[w][d] = pre + cnt;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t r = pre + below;
            if (j & 1) rank2[j >> 1] |= r << 16;
            else rank2[j >> 1] = r;
        }
        __syncthreads();
        // thread d < 256: counts of digit d per wave -> exclusive offsets across waves, tile total; look-back over the predecessors
        // (only the digits that exist: a canvas has `ndig` tile rows, 68 of 256 at 1080p)
        uint32_t cw[RA_WAVES], total = 0, excl = 0, incl = 0;
        if (tid < 256u) {
#pragma unroll
            for (int k = 0; k < RA_WAVES; ++k) { cw[k] = sh.hist[k][tid]; total += cw[k]; }
            incl = wave_incl_scan(total, lane);
            if (lane == 63) sh.wsum[w] = incl;
            if (tile > 0 && tid < ndig) {
                constexpr int LB = 8;
                bool found = false;
                for (int t = (int)tile - 1; t >= 0 && !found; t -= LB) {
                    uint32_t sv[LB];
#pragma unroll
                    for (int k = 0; k < LB; ++k) sv[k] = (t - k >= 0) ? ld_agent(status + (uint64_t)(t - k) * 256 + tid) : RA_PREFIX;
#pragma unroll
                    for (int k = 0; k < LB; ++k) {
                        if (found) break;
                        uint32_t v = sv[k], spins = 0;
                        while ((v & RA_FLAGS) == 0 && ++spins < GS_SPIN_LIMIT) { // not published yet: poll this one word
                            __builtin_amdgcn_s_sleep(1);
                            v = ld_agent(status + (uint64_t)(t - k) * 256 + tid);
                        }
                        if ((v & RA_FLAGS) == 0) { ctl->fault = 1u; found = true; break; }
                        excl += v & RA_VALUE;
                        if ((v & RA_FLAGS) == RA_PREFIX) found = true;
                    }
                }
                st_agent(status + (uint64_t)tile * 256 + tid, RA_PREFIX | ((excl + total) & RA_VALUE));
            }
        }
        __syncthreads();
        if (tid < 256u) {
            uint32_t wv = 0;
            for (uint32_t k = 0; k < w; ++k) wv += sh.wsum[k];
            uint32_t run = wv + incl - total; // first position of digit `tid` in the tile's sorted order
            if (tid == hole) sh.nvalid = run; // the holes start here (no digit above `hole` occurs)
            sh.gbase[tid] = sh.dbase[tid] + excl - run;
#pragma unroll
            for (int k = 0; k < RA_WAVES; ++k) { sh.hist[k][tid] = run; run += cw[k]; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RA_ITEMS; ++j) {
            const uint32_t d = ix[j] == GS_ROW_HOLE ? hole : (iy[j] & 0xFFu);
            const uint32_t r = (j & 1) ? (rank2[j >> 1] >> 16) : (rank2[j >> 1] & 0xFFFFu);
            const uint32_t pos = sh.hist[w][d] + r;
            sh.u.it[pos * 3 + 0] = ix[j];
            sh.u.it[pos * 3 + 1] = iy[j];
            sh.u.it[pos * 3 + 2] = iz[j];
        }
        __syncthreads();
        const uint32_t nvalid = sh.nvalid;
#pragma unroll
        for (int j = 0; j < RA_ITEMS; ++j) {
            const uint32_t pos = j * RA_THREADS + tid;
            if (pos < nvalid) {
                gs_item3 v;
                v.x = sh.u.it[pos * 3 + 0]; v.y = sh.u.it[pos * 3 + 1]; v.z = sh.u.it[pos * 3 + 2];
                const uint32_t g = sh.gbase[v.y & 0xFFu] + pos;
                if (g < row_cap) *reinterpret_cast<gs_item3*>(rows_out + (uint64_t)g * 3u) = v;
            }
        }
        __syncthreads(); // LDS is reused by the next tile
    }
}

// ------------------------------------------------------------------------------------------------
// Chunk geometry shared by the count / scan / expand kernels: tile row r holds items [ibase[r], ibase[r] + cnt[r]) of the
// row-sorted array and chunks [cbase[r], cbase[r + 1]) of RB_CH items each (the last one shorter).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_incl_scan64(unsigned long long v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = __shfl_up((uint32_t)v, d, 64), hi = __shfl_up((uint32_t)(v >> 32), d, 64);
        if ((int)lane >= d) v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
struct RowTables {
    uint32_t ibase[257]; // exclusive scan of GsControl::rowhist, clamped to the arrays' capacity
    uint32_t cbase[257]; // exclusive scan of ceil(items / RB_CH)
    uint32_t w4[8];
};
// The first 256 threads of the workgroup build the tables (every thread calls: the others only take the barriers); valid after
// the last barrier.  row_cap: slots the row-item arrays hold.  A frame whose items do not fit (the arena overflowed: it is
// flagged and rendered again with grown arrays) still counted ALL its items in the histogram while the projection dropped the
// ones that did not fit: the rows' extents are clamped to the arrays here, so that nothing is read past them (found by
// tests/test_gpu_parity.py::test_row_item_arena_overflow_regrows_and_rerenders: a memory fault).
__device__ __forceinline__ void row_tables(RowTables& T, const GsControl* ctl, uint32_t row_cap, uint32_t tid) {
    const uint32_t lane = tid & 63, w = tid >> 6;
    const bool on = tid < 256u;
    const uint32_t c = on ? gs_rowhist(ctl, tid) : 0u;
    const unsigned long long ic = wave_incl_scan64(c, lane); // (64 bits: the sum of the histogram can pass 2^32 before the clamp)
    if (on && lane == 63) { T.w4[w] = (uint32_t)(ic > 0xFFFFFFFFull ? 0xFFFFFFFFull : ic); }
    __syncthreads();
    uint32_t i0 = 0, cc = 0;
    if (on) {
        unsigned long long b = ic - c;
        for (uint32_t k = 0; k < w; ++k) b += T.w4[k];
        const unsigned long long e = b + c;
        i0 = b > row_cap ? row_cap : (uint32_t)b;
        const uint32_t i1 = e > row_cap ? row_cap : (uint32_t)e;
        cc = i1 - i0;
        T.ibase[tid] = i0;
        if (tid == 255u) T.ibase[256] = i1;
    }
    const uint32_t ch = (cc + RB_CH - 1u) / RB_CH;
    const uint32_t ih = wave_incl_scan(ch, lane);
    if (on && lane == 63) T.w4[4 + w] = ih;
    __syncthreads();
    if (on) {
        uint32_t bh = 0;
        for (uint32_t k = 0; k < w; ++k) bh += T.w4[4 + k];
        T.cbase[tid] = bh + ih - ch;
        if (tid == 255u) T.cbase[256] = bh + ih;
    }
    __syncthreads();
}
// tile row of chunk c (c < cbase[256]): the LAST r with cbase[r] <= c (rows without chunks share their successor's base).
// cbase is non-decreasing, so r = the number of m in 1..255 with cbase[m] <= c: every wave counts them with four reads per lane
// and four ballots -- one LDS round trip instead of the eight dependent ones of a binary search.
__device__ __forceinline__ uint32_t row_of_chunk(const RowTables& T, uint32_t c) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = T.cbase[lane + 64u * k];
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) r += (uint32_t)__popcll(__ballot(v[k] <= c && (lane + 64u * k) != 0u));
    return r;
}

__global__ __launch_bounds__(256) void gs_rows_count_kernel(const uint32_t* __restrict__ rows, const GsControl* ctl, uint32_t* __restrict__ M3,
                                                            uint32_t chunk_cap, uint32_t row_cap) {
    __shared__ RowTables T;
    __shared__ int s_diff[257];
    __shared__ uint32_t s_w[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    row_tables(T, ctl, row_cap, tid);
    uint32_t nch = T.cbase[256];
    if (nch > chunk_cap) nch = chunk_cap;
    for (uint32_t c = blockIdx.x; c < nch; c += gridDim.x) {
        const uint32_t r = row_of_chunk(T, c);
        const uint32_t i0 = T.ibase[r] + (c - T.cbase[r]) * RB_CH;
        const uint32_t i1 = (i0 + RB_CH < T.ibase[r + 1]) ? i0 + RB_CH : T.ibase[r + 1];
        s_diff[tid] = 0;
        if (tid == 0) s_diff[256] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < (int)(RB_CH / 256u); ++k) {
            const uint32_t i = i0 + k * 256u + tid;
            if (i < i1) {
                const uint32_t w1 = rows[(uint64_t)i * 3u + 1u];
                const uint32_t tlo = (w1 >> 8) & 0xFFu, len = ((w1 >> 16) & 0xFFu) + 1u;
                atomicAdd(&s_diff[tlo], 1);
                atomicAdd(&s_diff[tlo + len > 256u ? 256u : tlo + len], -1);
            }
        }
        __syncthreads();
        // instances of column `tid` = inclusive prefix of the difference array
        const uint32_t v = (uint32_t)s_diff[tid];
        const uint32_t incl = wave_incl_scan(v, lane);
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t b = 0;
        for (uint32_t k = 0; k < w; ++k) b += s_w[k];
        M3[(uint64_t)c * 256u + tid] = b + incl;
        __syncthreads();
    }
}

// One workgroup of 1024 threads per tile row: thread (column c, part p) takes a quarter of the row's chunks.
// tileoff[r * 256 + c] = instances of row r in columns < c; rowtot[r] = instances of the row (both saturate at 2^32 - 1).
__global__ __launch_bounds__(1024) void gs_rows_scan_kernel(const GsControl* ctl, uint32_t* __restrict__ M3, uint32_t* __restrict__ tileoff,
                                                            uint32_t* __restrict__ rowtot, uint32_t chunk_cap, uint32_t row_cap) {
    __shared__ RowTables T;
    __shared__ unsigned long long s_part[4][256];
    __shared__ unsigned long long s_ex[256];
    __shared__ unsigned long long s_w[4];
    const uint32_t tid = threadIdx.x, c = tid & 255u, p = tid >> 8, r = blockIdx.x;
    row_tables(T, ctl, row_cap, tid);
    uint32_t c0 = T.cbase[r], c1 = T.cbase[r + 1];
    if (c0 > chunk_cap) c0 = chunk_cap;
    if (c1 > chunk_cap) c1 = chunk_cap;
    const uint32_t nch = c1 - c0, per = (nch + 3u) / 4u;
    const uint32_t j0 = c0 + (p * per < nch ? p * per : nch), j1 = c0 + ((p + 1u) * per < nch ? (p + 1u) * per : nch);
    unsigned long long sum = 0;
    for (uint32_t j = j0; j < j1; j += 8u) { // eight independent loads in flight
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (j + k < j1) ? M3[(uint64_t)(j + k) * 256u + c] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += v[k];
    }
    s_part[p][c] = sum;
    __syncthreads();
    unsigned long long run = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const unsigned long long t = s_part[k][c]; if (k < (int)p) run += t; tot += t; }
    for (uint32_t j = j0; j < j1; j += 8u) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (j + k < j1) ? M3[(uint64_t)(j + k) * 256u + c] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (j + k < j1) M3[(uint64_t)(j + k) * 256u + c] = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)run;
            run += v[k];
        }
    }
    // exclusive scan of the tile totals over the row's columns (threads of part 0)
    if (p == 0) {
        const uint32_t lane = tid & 63, w = tid >> 6;
        unsigned long long incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = __shfl_up((uint32_t)incl, d, 64), hi = __shfl_up((uint32_t)(incl >> 32), d, 64);
            if ((int)lane >= d) incl += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 63) s_w[w] = incl;
        s_ex[c] = incl - tot; // in-wave exclusive
    }
    __syncthreads();
    if (p == 0) {
        const uint32_t w = tid >> 6;
        unsigned long long b = 0, all = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (k < (int)w) b += s_w[k]; all += s_w[k]; }
        const unsigned long long ex = b + s_ex[c];
        tileoff[r * 256u + c] = ex > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ex;
        if (c == 0) rowtot[r] = all > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)all;
    }
}

// ------------------------------------------------------------------------------------------------
// Expansion: the final per-tile lists.
// ------------------------------------------------------------------------------------------------
struct ExpandShared {
    RowTables T;
    uint32_t rbase[257];      // first instance of every tile row in the lists (exclusive scan of rowtot, saturating)
    uint32_t w0[RB_CH], w1[RB_CH], w2[RB_CH];
    uint32_t hist[4][256];
    uint32_t gbase[256];
    uint32_t wsum[8];
    uint32_t nvalid, total;
    unsigned char cols[RB_SB];
    union {                   // during the expansion: item prefix + owner marks; during the reorder: the sorted values
        struct { uint32_t P[RB_CH + 1]; unsigned short mark[RB_SB]; } e;
        uint32_t vals[RB_SB];
    } x;
};

// sticky: see k_binning.hip fold_sticky (frames enqueued before the last one report their overflow / fault / size here)
__global__ __launch_bounds__(256) void gs_rows_expand_kernel(const uint32_t* __restrict__ rows, GsControl* ctl, const uint32_t* __restrict__ M3,
                                                             const uint32_t* __restrict__ tileoff, const uint32_t* __restrict__ rowtot, GsFrame f,
                                                             uint32_t* __restrict__ values, uint32_t* __restrict__ ranges, uint32_t chunk_cap,
                                                             uint32_t row_cap, uint32_t* sticky, GsReport* rep) {
    __shared__ ExpandShared S;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t ns = f.tile_size >= 16u ? 2u : 1u;
    const uint32_t hole = f.ntx < 128u ? 127u : 255u; // the digit of the slots past a sub-batch's end: no tile column has it
    const int nbits = f.ntx < 128u ? 7 : 8;
    row_tables(S.T, ctl, row_cap, tid);
    {   // first instance of every tile row
        const uint32_t v = tid < f.nty ? rowtot[tid] : 0u;
        unsigned long long incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = __shfl_up((uint32_t)incl, d, 64), hi = __shfl_up((uint32_t)(incl >> 32), d, 64);
            if ((int)lane >= d) incl += ((unsigned long long)hi << 32) | lo;
        }
        __shared__ unsigned long long s_r[4];
        if (lane == 63) s_r[w] = incl;
        __syncthreads();
        unsigned long long b = 0;
        for (uint32_t k = 0; k < w; ++k) b += s_r[k];
        const unsigned long long ex = b + incl - v;
        S.rbase[tid] = ex > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ex;
        if (tid == 255u) S.rbase[256] = (b + incl) > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(b + incl);
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid == 0) { // the frame's instance count; its flags and sizes go to the sticky words and the host's report
        const uint32_t I = S.rbase[256];
        ctl->num_intersections = I;
        ctl->num_items = S.T.ibase[256];
        if (I > f.capacity) ctl->overflow = 1u;
        gs_frame_report(ctl, I, S.T.ibase[256], f.capacity, sticky, rep);
    }
    // ranges[t] = end of tile t's list (compute_ranges.wgsl:5-29): workgroup r writes tile row r
    for (uint32_t r = blockIdx.x; r < f.nty; r += gridDim.x) {
        if (tid < f.ntx) {
            const uint32_t endoff = (tid + 1u < 256u) ? tileoff[r * 256u + tid + 1u] : rowtot[r];
            const unsigned long long e = (unsigned long long)S.rbase[r] + endoff;
            ranges[r * f.ntx + tid] = e > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)e;
        }
    }
    uint32_t nch = S.T.cbase[256];
    if (nch > chunk_cap) nch = chunk_cap;
    // The global loads of a chunk (its items, the column's start and earlier-chunk count) are issued one chunk AHEAD: a chunk is
    // a dozen short barrier-separated phases, and with the loads at its head every one of them waited for HBM first.
    gs_item3 nit[RB_IT];
    uint32_t n_r = 0, n_ni = 0, n_tstart = 0, n_done = 0;
    auto prefetch = [&](uint32_t c) {
        n_r = row_of_chunk(S.T, c);
        const uint32_t i0 = S.T.ibase[n_r] + (c - S.T.cbase[n_r]) * RB_CH;
        const uint32_t i1 = (i0 + RB_CH < S.T.ibase[n_r + 1]) ? i0 + RB_CH : S.T.ibase[n_r + 1];
        n_ni = i1 - i0;
#pragma unroll
        for (int k = 0; k < (int)RB_IT; ++k) {
            const uint32_t e = tid * RB_IT + k;
            nit[k] = *reinterpret_cast<const gs_item3*>(rows + (uint64_t)(i0 + (e < n_ni ? e : 0u)) * 3u);
        }
        n_tstart = S.rbase[n_r] + tileoff[n_r * 256u + tid];
        n_done = M3[(uint64_t)c * 256u + tid];
    };
    if (blockIdx.x < nch) prefetch(blockIdx.x);
    for (uint32_t c = blockIdx.x; c < nch; c += gridDim.x) {
        const uint32_t ni = n_ni;
        // ---- the chunk's items (thread t: RB_IT consecutive ones) and the prefix of their lengths ----
        uint32_t len[RB_IT], lsum = 0;
#pragma unroll
        for (int k = 0; k < (int)RB_IT; ++k) {
            const uint32_t e = tid * RB_IT + k;
            len[k] = 0u;
            if (e < ni) {
                S.w0[e] = nit[k].x; S.w1[e] = nit[k].y; S.w2[e] = nit[k].z;
                len[k] = ((nit[k].y >> 16) & 0xFFu) + 1u;
            }
            lsum += len[k];
        }
        const uint32_t tstart = n_tstart; // first list slot of column `tid` for this chunk: tile start ...
        uint32_t done = n_done;           // ... + what the row's earlier chunks put there
        if (c + gridDim.x < nch) prefetch(c + gridDim.x);
        {
            const uint32_t incl = wave_incl_scan(lsum, lane);
            if (lane == 63) S.wsum[w] = incl;
            __syncthreads();
            uint32_t b = 0, all = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { if (k < (int)w) b += S.wsum[k]; all += S.wsum[k]; }
            uint32_t run = b + incl - lsum;
#pragma unroll
            for (int k = 0; k < (int)RB_IT; ++k) { S.x.e.P[tid * RB_IT + k] = run; run += len[k]; }
            if (tid == 0) { S.x.e.P[RB_CH] = all; S.total = all; }
        }
        __syncthreads();
        const uint32_t ninst = S.total;
        for (uint32_t s0 = 0; s0 < ninst; s0 += RB_SB) {
            // ---- owner marks: an item whose first instance falls into the sub-batch marks that slot ----
            if (s0) {
                __syncthreads(); // the previous sub-batch's stores have read x.vals: P / mark are rebuilt (P from the items kept in LDS)
                uint32_t l2[RB_IT], ls = 0;
#pragma unroll
                for (int k = 0; k < (int)RB_IT; ++k) { const uint32_t e = tid * RB_IT + k; l2[k] = e < ni ? ((S.w1[e] >> 16) & 0xFFu) + 1u : 0u; ls += l2[k]; }
                const uint32_t incl = wave_incl_scan(ls, lane);
                if (lane == 63) S.wsum[w] = incl;
                __syncthreads();
                uint32_t b = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) if (k < (int)w) b += S.wsum[k];
                uint32_t run = b + incl - ls;
#pragma unroll
                for (int k = 0; k < (int)RB_IT; ++k) { S.x.e.P[tid * RB_IT + k] = run; run += l2[k]; }
                if (tid == 0) S.x.e.P[RB_CH] = ninst;
            }
            for (uint32_t k = tid; k < RB_SB / 2u; k += 256u) reinterpret_cast<uint32_t*>(S.x.e.mark)[k] = 0u;
            for (uint32_t k = lane; k < 256; k += 64) S.hist[w][k] = 0u;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < (int)RB_IT; ++k) {
                const uint32_t e = tid * RB_IT + k;
                if (e < ni) {
                    const uint32_t p0 = S.x.e.P[e];
                    if (p0 >= s0 && p0 < s0 + RB_SB) S.x.e.mark[p0 - s0] = (unsigned short)(e + 1u);
                }
            }
            __syncthreads();
            // ---- expansion: thread slot (w, j, lane) = instance s0 + w * RB_WSL + j * 64 + lane ----
            // owner of the wave's first slot: the last item whose prefix is <= it
            uint32_t carry;
            {   // (P is non-decreasing and P[0] = 0: the owner is the number of m in 1..ni-1 with P[m] <= x0 -- eight reads per lane
                // and eight ballots instead of nine dependent reads)
                const uint32_t x0 = s0 + w * RB_WSL;
                uint32_t pv[RB_CH / 64u];
#pragma unroll
                for (int k = 0; k < (int)(RB_CH / 64u); ++k) pv[k] = S.x.e.P[lane + 64u * k];
                uint32_t e = 0;
#pragma unroll
                for (int k = 0; k < (int)(RB_CH / 64u); ++k) {
                    const uint32_t m = lane + 64u * k;
                    e += (uint32_t)__popcll(__ballot(m != 0u && m < ni && pv[k] <= x0));
                }
                carry = e + 1u;
            }
            uint32_t val[RB_PER], col4[RB_PER / 4u];
            uint32_t rank2[RB_PER / 2u];
#pragma unroll
            for (int j = 0; j < (int)(RB_PER / 4u); ++j) col4[j] = 0u;
            // this wave's batches of 64 slots that hold an instance (the rest of its RB_PER are skipped: a chunk averages two
            // thirds of a sub-batch)
            const uint32_t wfirst = s0 + w * RB_WSL;
            const uint32_t jn = wfirst >= ninst ? 0u : ((ninst - wfirst + 63u) / 64u < RB_PER ? (ninst - wfirst + 63u) / 64u : RB_PER);
#pragma unroll
            for (int j = 0; j < (int)RB_PER; ++j) {
                if ((uint32_t)j >= jn) break;
                const uint32_t sp = w * RB_WSL + j * 64u + lane, x = s0 + sp;
                uint32_t m = wave_incl_max((uint32_t)S.x.e.mark[sp]);
                m = m > carry ? m : carry;
                carry = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
                uint32_t d = hole;
                val[j] = 0u;
                if (x < ninst) {
                    const uint32_t e = m - 1u;
                    const uint32_t q = x - S.x.e.P[e], i1w = S.w1[e];
                    d = ((i1w >> 8) & 0xFFu) + q;
                    val[j] = (S.w0[e] & GS_ID_MASK) | (tight_item_mask(S.w2[e], q, ns) << GS_ID_BITS);
                }
                col4[j >> 2] |= d << (8 * (j & 3));
            }
            // ---- rank by column inside the wave (slots past the end take the digit `hole`) ----
#pragma unroll
            for (int j = 0; j < (int)RB_PER; ++j) {
                if ((uint32_t)j >= jn) break;
                const uint32_t d = (col4[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                uint32_t plo = 0xFFFFFFFFu, phi = 0xFFFFFFFFu;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    if (b < nbits) {
                        const uint32_t bit = (d >> b) & 1u;
                        const unsigned long long bal = __ballot(bit != 0u);
                        const uint32_t inv = bit - 1u;
                        plo &= (uint32_t)bal ^ inv;
                        phi &= (uint32_t)(bal >> 32) ^ inv;
                    }
                }
                const uint32_t below = __popc(plo & (uint32_t)lt_mask) + __popc(phi & (uint32_t)(lt_mask >> 32));
                const uint32_t cnt = __popc(plo) + __popc(phi);
                const uint32_t pre = S.hist[w][d];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (below == 0) S.hist[w][d] = pre + cnt;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t rr = pre + below;
                if (j & 1) rank2[j >> 1] |= rr << 16;
                else rank2[j >> 1] = rr;
            }
            __syncthreads(); // (also: every wave is done with P / mark, which the sorted values overwrite below)
            // ---- thread = column: exclusive offsets across waves, start of the column's run in the sub-batch's sorted order ----
            {
                uint32_t cw[4], total = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) { cw[k] = S.hist[k][tid]; total += cw[k]; }
                const uint32_t incl = wave_incl_scan(total, lane);
                if (lane == 63) S.wsum[4 + w] = incl;
                __syncthreads();
                uint32_t b = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) if (k < (int)w) b += S.wsum[4 + k];
                uint32_t run = b + incl - total;
                if (tid == hole) S.nvalid = run;
                S.gbase[tid] = tstart + done - run; // (wraps are harmless: only gbase + position is used, and it is bounds-checked)
                done += total;
#pragma unroll
                for (int k = 0; k < 4; ++k) { S.hist[k][tid] = run; run += cw[k]; }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < (int)RB_PER; ++j) {
                if ((uint32_t)j >= jn) break;
                const uint32_t d = (col4[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                const uint32_t rr = (j & 1) ? (rank2[j >> 1] >> 16) : (rank2[j >> 1] & 0xFFFFu);
                const uint32_t pos = S.hist[w][d] + rr;
                S.x.vals[pos] = val[j];
                S.cols[pos] = (unsigned char)d;
            }
            __syncthreads();
            const uint32_t nvalid = S.nvalid;
#pragma unroll
            for (int j = 0; j < (int)RB_PER; ++j) {
                const uint32_t pos = j * 256u + tid;
                if (pos < nvalid) {
                    const uint32_t dst = S.gbase[S.cols[pos]] + pos;
                    if (dst < f.capacity) values[dst] = S.x.vals[pos];
                }
            }
        }
        __syncthreads(); // the items / tables of the next chunk overwrite LDS
    }
}

// GS_BUF_KEYS tap of a tight frame: key = tile * 1000 + depth bucket of the gaussian (write_tile_ids.wgsl:31); the tile of list
// entry j is found in `ranges`, the bucket in the high bits of the gaussian's count word.
__global__ __launch_bounds__(256) void gs_rows_rebuild_keys_kernel(const uint32_t* __restrict__ ranges, uint32_t T, const uint32_t* __restrict__ vals,
                                                                   const uint32_t* __restrict__ counts, uint32_t count, uint32_t n,
                                                                   uint32_t* __restrict__ keys) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        uint32_t lo = 0, hi = T; // first tile whose end is > i
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (ranges[mid] > i) hi = mid; else lo = mid + 1u;
        }
        const uint32_t g = vals[i] & GS_ID_MASK;
        keys[i] = lo * 1000u + (g < n ? counts[g] >> GS_COUNT_BITS : 0u);
    }
}

#ifndef RB_CNT_WG
#define RB_CNT_WG 8u // workgroups per CU of the count ...
#endif
#ifndef RB_EXP_WG
#define RB_EXP_WG 6u // ... and of the expansion (residency)
#endif
// ---- host launchers --------------------------------------------------------------------------------
uint32_t gs_rows_sort_tiles(uint64_t row_cap) { return (uint32_t)((row_cap + RA_TILE - 1) / RA_TILE); }
uint32_t gs_rows_chunks(uint64_t row_cap) { return (uint32_t)(row_cap / RB_CH + 256u); }
// cus: compute units (grids are sized by residency: three 8-wave workgroups of the sort, five of the expansion, eight of the count fit a CU)
void gs_launch_rows(const uint32_t* arena, const void* grec, const uint32_t* chunk_table, uint32_t* rows_sorted, GsControl* ctl, uint32_t* sort_status, uint32_t row_cap,
                    uint32_t* M3, uint32_t* tileoff, uint32_t* rowtot, const GsFrame& f, uint32_t* values, uint32_t* ranges, uint32_t cus,
                    uint32_t* sticky, GsReport* rep, hipStream_t st, void (*mark)(void*, int), void* mark_arg) {
    const uint32_t chunk_cap = gs_rows_chunks(row_cap);
    if (!cus) cus = 1;
    hipLaunchKernelGGL(gs_rows_sort_kernel, dim3(cus * 3u), dim3(RA_THREADS), 0, st, arena, (const uint4*)grec, chunk_table, rows_sorted, ctl, sort_status,
                       row_cap, f.nty);
    if (mark) mark(mark_arg, 3);
    hipLaunchKernelGGL(gs_rows_count_kernel, dim3(cus * RB_CNT_WG), dim3(256), 0, st, (const uint32_t*)rows_sorted, (const GsControl*)ctl, M3, chunk_cap, row_cap);
    hipLaunchKernelGGL(gs_rows_scan_kernel, dim3(f.nty), dim3(1024), 0, st, (const GsControl*)ctl, M3, tileoff, rowtot, chunk_cap, row_cap);
    hipLaunchKernelGGL(gs_rows_expand_kernel, dim3(cus * RB_EXP_WG), dim3(256), 0, st, (const uint32_t*)rows_sorted, ctl, (const uint32_t*)M3,
                       (const uint32_t*)tileoff, (const uint32_t*)rowtot, f, values, ranges, chunk_cap, row_cap, sticky, rep);
}
void gs_launch_rows_rebuild_keys(const uint32_t* ranges, uint32_t T, const uint32_t* vals, const uint32_t* counts, uint32_t count, uint32_t n,
                                 uint32_t* keys, hipStream_t st) {
    if (!count) return;
    const uint32_t blocks = (count + 255u) / 256u;
    hipLaunchKernelGGL(gs_rows_rebuild_keys_kernel, dim3(blocks < 4096u ? blocks : 4096u), dim3(256), 0, st, ranges, T, vals, counts, count, n, keys);
}
