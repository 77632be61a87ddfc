// k_gsort.hip -- the gaussian-level stage of the depth-ordered pipeline: the frame's visible gaussians, sorted by the depth
// bucket of their sort key, with their tile counts scanned in that order.
//
// The reference sorts every (tile, gaussian) INSTANCE by tile*1000 + bucket (write_tile_ids.wgsl:31, radix_sort.wgsl).  The
// order it needs inside a tile is (bucket, gaussian index); sorting the N_vis visible GAUSSIANS by bucket first (stable, so
// index order survives inside a bucket: 7-16x fewer elements than instances) and emitting their instances in that order leaves
// only the tile id for the stable instance sort.  Round 1 did this with two 5-bit sweeps of the generic look-back radix sort
// plus a second scan over the sorted counts (4 dependent, latency-bound launches for 20 MB: 150 us).  Here it is one stable
// 10-bit counting sort without any look-back or polling, and the scan of the tile counts in sorted order falls out of the
// same pass, because the per-(bucket, tile) table carries two channels -- how many gaussians, how many instances:
//   hist     every tile of 4096 visible gaussians counts its buckets          -> M[bucket][tile] = (gaussians, instances)
//   rowscan  every bucket's row of M is scanned over the tiles (exclusive)    -> row totals
//   scatter  every tile ranks its gaussians by bucket (wave ballots, stable), reorders them through LDS, scans their tile
//            counts in that order, and writes  perm / counts / offsets  at  base[bucket] + M[bucket][tile] + rank,
//            plus the emission's chunk table (first gaussian of every EMIT_CHUNK output slots).
// All three are HBM-trivial (10-40 MB); what they cost is their dependent launches.
#include "gs_device.h"

#ifndef GT_ITEMS
#define GT_ITEMS 20 // 5120 per tile: config B's 2.43 M visible gaussians are 475 tiles = ONE residency round at two workgroups per CU
#endif              // (16: 594 tiles, two rounds, scatter 42 us)
#define GT_THREADS 256
#define GT (GT_THREADS * GT_ITEMS) // visible gaussians per tile
#define GBINS 1024                 // bucket = u32(min(50 depth, 999)) < 1000 (write_tile_ids.wgsl:31)
#define GS_EMIT_CHUNK_SHIFT 10     // = EMIT_CHUNK_SHIFT of k_binning.hip

__global__ __launch_bounds__(GT_THREADS) void gs_gsort_hist_kernel(const uint32_t* __restrict__ words, const GsControl* ctl, uint2* __restrict__ M,
                                                                   uint32_t NT) {
    __shared__ uint32_t s_cnt[GBINS], s_sum[GBINS];
    const uint32_t nvis = ctl->num_visible, nt = (nvis + GT - 1) / GT, tile = blockIdx.x, tid = threadIdx.x;
    if (tile >= nt) return;
    for (uint32_t b = tid; b < GBINS; b += GT_THREADS) { s_cnt[b] = 0u; s_sum[b] = 0u; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < GT_ITEMS; ++j) {
        const uint32_t k = tile * GT + j * GT_THREADS + tid;
        if (k < nvis) {
            const uint32_t w = words[k], b = w >> GS_COUNT_BITS;
            atomicAdd(&s_cnt[b], 1u);
            atomicAdd(&s_sum[b], w & GS_COUNT_MASK);
        }
    }
    __syncthreads();
    for (uint32_t b = tid; b < GBINS; b += GT_THREADS) M[(uint64_t)b * NT + tile] = make_uint2(s_cnt[b], s_sum[b]);
}

// exclusive scan of two channels over a workgroup: each thread holds `items` consecutive elements (already summed into v)
__device__ __forceinline__ uint2 block_excl2(uint2 v, uint32_t tid, uint2* s_w /*[4]*/, uint2& total) {
    const uint32_t lane = tid & 63, w = tid >> 6;
    const uint32_t ix = wave_incl_scan(v.x, lane), iy = wave_incl_scan(v.y, lane);
    if (lane == 63) s_w[w] = make_uint2(ix, iy);
    __syncthreads();
    uint2 base = make_uint2(0u, 0u);
    total = make_uint2(0u, 0u);
#pragma unroll
    for (int k = 0; k < GT_THREADS / 64; ++k) {
        const uint2 t = s_w[k];
        if (k < (int)w) { base.x += t.x; base.y += t.y; }
        total.x += t.x; total.y += t.y;
    }
    __syncthreads();
    return make_uint2(base.x + ix - v.x, base.y + iy - v.y);
}

__global__ __launch_bounds__(GT_THREADS) void gs_gsort_rowscan_kernel(uint2* __restrict__ M, uint32_t NT, const GsControl* ctl, uint2* __restrict__ rowtot) {
    __shared__ uint2 s_w[4];
    const uint32_t nvis = ctl->num_visible, nt = (nvis + GT - 1) / GT, b = blockIdx.x, tid = threadIdx.x;
    uint2* row = M + (uint64_t)b * NT;
    const uint32_t per = (nt + GT_THREADS - 1) / GT_THREADS; // consecutive tiles per thread
    uint2 acc = make_uint2(0u, 0u);
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t t = tid * per + i;
        if (t < nt) { const uint2 v = row[t]; acc.x += v.x; acc.y += v.y; }
    }
    uint2 total;
    uint2 run = block_excl2(acc, tid, s_w, total);
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t t = tid * per + i;
        if (t < nt) { const uint2 v = row[t]; row[t] = run; run.x += v.x; run.y += v.y; }
    }
    if (tid == 0) rowtot[b] = total;
}

struct GsortShared {
    uint2 base[GBINS];                 // first sorted position / first instance offset of (bucket, this tile)
    uint32_t binstart[GBINS];          // first slot of the bucket in the tile's sorted order
    uint32_t whist[(GT > (GT_THREADS / 64) * GBINS ? GT : (GT_THREADS / 64) * GBINS) / GBINS][GBINS]; // per-wave running counts while ranking ([wave][bucket]); then the prefix of the sorted tile counts (GT words)
    uint32_t id[GT], word[GT];
    uint2 w2[4];
};

__global__ __launch_bounds__(GT_THREADS) void gs_gsort_scatter_kernel(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ words,
                                                                      const GsControl* ctl, const uint2* __restrict__ M, uint32_t NT,
                                                                      const uint2* __restrict__ rowtot, uint32_t* __restrict__ perm,
                                                                      uint32_t* __restrict__ scounts, uint32_t* __restrict__ offsets,
                                                                      uint32_t* __restrict__ chunk_table, uint32_t chunk_cap) {
    __shared__ GsortShared S;
    const uint32_t nvis = ctl->num_visible, nt = (nvis + GT - 1) / GT, tile = blockIdx.x;
    if (tile >= nt) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    constexpr uint32_t BPT = GBINS / GT_THREADS; // buckets per thread (4 consecutive)

    // ---- bucket bases: exclusive scan of the row totals, plus this tile's entry of the scanned table ----
    {
        uint2 t[BPT], acc = make_uint2(0u, 0u);
#pragma unroll
        for (uint32_t i = 0; i < BPT; ++i) { t[i] = rowtot[tid * BPT + i]; acc.x += t[i].x; acc.y += t[i].y; }
        uint2 total;
        uint2 run = block_excl2(acc, tid, S.w2, total);
#pragma unroll
        for (uint32_t i = 0; i < BPT; ++i) {
            const uint32_t b = tid * BPT + i;
            const uint2 m = M[(uint64_t)b * NT + tile];
            S.base[b] = make_uint2(run.x + m.x, run.y + m.y);
            run.x += t[i].x; run.y += t[i].y;
        }
    }
    for (uint32_t k = lane; k < GBINS; k += 64) S.whist[w][k] = 0u;
    // ---- the tile's gaussians: wave w owns 1024 consecutive ones, item j of lane l = element w*1024 + j*64 + l ----
    uint32_t gid[GT_ITEMS], wd[GT_ITEMS];
    uint32_t rank2[GT_ITEMS / 2];
    const uint32_t e0 = tile * GT + w * (64 * GT_ITEMS) + lane;
#pragma unroll
    for (int j = 0; j < GT_ITEMS; ++j) {
        const uint32_t k = e0 + j * 64;
        const bool in = k < nvis;
        gid[j] = in ? ids[k] : 0u;
        wd[j] = in ? words[k] : 0xFFFFFFFFu; // absent: bucket 1023, sorts behind every real one and is never stored
    }
    __syncthreads();
    // rank inside the wave: peers = lanes holding the same bucket (10 ballots), order = (item, lane): stable
#pragma unroll
    for (int j = 0; j < GT_ITEMS; ++j) {
        const uint32_t d = wd[j] >> GS_COUNT_BITS;
        uint32_t plo = 0xFFFFFFFFu, phi = 0xFFFFFFFFu;
#pragma unroll
        for (int b = 0; b < 10; ++b) {
            const uint32_t bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit != 0u);
            const uint32_t inv = bit - 1u;
            plo &= (uint32_t)bal ^ inv;
            phi &= (uint32_t)(bal >> 32) ^ inv;
        }
        const uint32_t below = __popc(plo & (uint32_t)lt_mask) + __popc(phi & (uint32_t)(lt_mask >> 32));
        const uint32_t cnt = __popc(plo) + __popc(phi);
        const uint32_t pre = S.whist[w][d];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // every peer has read `pre` before the leader's store (one wave, in-order LDS)
        if (below == 0) S.whist[w][d] = pre + cnt;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t r = pre + below; // < 1024
        if (j & 1) rank2[j >> 1] |= r << 16;
        else rank2[j >> 1] = r;
    }
    __syncthreads();
    // ---- per-wave counts -> exclusive across waves; bucket starts inside the tile ----
    {
        uint32_t tot[BPT], acc = 0;
#pragma unroll
        for (uint32_t i = 0; i < BPT; ++i) {
            const uint32_t b = tid * BPT + i;
            uint32_t run = 0;
#pragma unroll
            for (int k = 0; k < GT_THREADS / 64; ++k) { const uint32_t t = S.whist[k][b]; S.whist[k][b] = run; run += t; }
            tot[i] = run;
            acc += run;
        }
        uint2 total;
        const uint2 ex = block_excl2(make_uint2(acc, 0u), tid, S.w2, total);
        uint32_t run = ex.x;
#pragma unroll
        for (uint32_t i = 0; i < BPT; ++i) { S.binstart[tid * BPT + i] = run; run += tot[i]; }
    }
    __syncthreads();
    // ---- reorder through LDS ----
#pragma unroll
    for (int j = 0; j < GT_ITEMS; ++j) {
        const uint32_t d = wd[j] >> GS_COUNT_BITS;
        const uint32_t r = (j & 1) ? (rank2[j >> 1] >> 16) : (rank2[j >> 1] & 0xFFFFu);
        const uint32_t pos = S.binstart[d] + S.whist[w][d] + r;
        S.id[pos] = gid[j];
        S.word[pos] = wd[j];
    }
    __syncthreads();
    // ---- exclusive scan of the tile counts in sorted order (thread t: slots 16t .. 16t+15) ----
    uint32_t* P = &S.whist[0][0]; // GT words: the ranking counters are dead now
    {
        uint32_t c[GT_ITEMS], acc = 0;
#pragma unroll
        for (int j = 0; j < GT_ITEMS; ++j) {
            const uint32_t wv = S.word[tid * GT_ITEMS + j];
            c[j] = wv == 0xFFFFFFFFu ? 0u : (wv & GS_COUNT_MASK);
            acc += c[j];
        }
        uint2 total;
        const uint2 ex = block_excl2(make_uint2(acc, 0u), tid, S.w2, total);
        uint32_t run = ex.x;
        __syncthreads(); // (block_excl2 ends with a barrier; this one orders the reuse of whist as P for every wave)
#pragma unroll
        for (int j = 0; j < GT_ITEMS; ++j) { P[tid * GT_ITEMS + j] = run; run += c[j]; }
    }
    __syncthreads();
    // ---- store: coalesced over the sorted slots ----
#pragma unroll
    for (int j = 0; j < GT_ITEMS; ++j) {
        const uint32_t pos = j * GT_THREADS + tid;
        const uint32_t wv = S.word[pos];
        if (wv == 0xFFFFFFFFu) continue;
        const uint32_t b = wv >> GS_COUNT_BITS, first = S.binstart[b];
        const uint2 base = S.base[b];
        const uint32_t g = base.x + (pos - first);
        const uint32_t off = base.y + (P[pos] - P[first]);
        perm[g] = S.id[pos];
        scounts[g] = wv;
        offsets[g] = off;
        const uint32_t cnt = wv & GS_COUNT_MASK; // > 0: only visible gaussians are here
        const uint32_t last = (off + cnt - 1u) >> GS_EMIT_CHUNK_SHIFT;
        for (uint32_t c = (off + (1u << GS_EMIT_CHUNK_SHIFT) - 1u) >> GS_EMIT_CHUNK_SHIFT; c <= last && c < chunk_cap; ++c) chunk_table[c] = g;
    }
}

// ---- host launchers --------------------------------------------------------------------------------
uint32_t gs_gsort_tiles(uint32_t n) { return (n + GT - 1) / GT; }
uint64_t gs_gsort_scratch_bytes(uint32_t n) { return ((uint64_t)GBINS * gs_gsort_tiles(n ? n : 1) + GBINS) * sizeof(uint2); }
// ids / words: the visible gaussians in index order with their tile-count words (count | bucket << 22), ctl->num_visible of them
// (the scan's compaction); scratch: gs_gsort_scratch_bytes(n_max) bytes.  Outputs in (bucket, index) order.
void gs_launch_gsort(const uint32_t* ids, const uint32_t* words, const GsControl* ctl, uint32_t n_max, void* scratch, uint32_t* perm,
                     uint32_t* scounts, uint32_t* offsets, uint32_t* chunk_table, uint32_t chunk_cap, hipStream_t st) {
    const uint32_t NT = gs_gsort_tiles(n_max ? n_max : 1);
    uint2* M = (uint2*)scratch;
    uint2* rowtot = M + (uint64_t)GBINS * NT;
    hipLaunchKernelGGL(gs_gsort_hist_kernel, dim3(NT), dim3(GT_THREADS), 0, st, words, ctl, M, NT);
    hipLaunchKernelGGL(gs_gsort_rowscan_kernel, dim3(GBINS), dim3(GT_THREADS), 0, st, M, NT, ctl, rowtot);
    hipLaunchKernelGGL(gs_gsort_scatter_kernel, dim3(NT), dim3(GT_THREADS), 0, st, ids, words, ctl, (const uint2*)M, NT, (const uint2*)rowtot, perm, scounts,
                       offsets, chunk_table, chunk_cap);
}
