// k_gsort.hip -- the gaussian-level stage of the depth-ordered pipeline: the frame's visible gaussians, sorted by the depth
// bucket of their sort key, with a per-gaussian quantity (tile count, or row-item slots) scanned in that order.
//
// The reference sorts every (tile, gaussian) INSTANCE by tile*1000 + bucket (write_tile_ids.wgsl:31, radix_sort.wgsl).  The
// order it needs inside a tile is (bucket, gaussian index); sorting the N_vis visible GAUSSIANS by bucket first (stable, so
// index order survives inside a bucket: 7-16x fewer elements than instances) and binning their instances in that order
// leaves only the tile id for the stable instance binning.
//
// Round 2 fed this stage from a chained look-back scan that compacted the visible gaussians (31 us for 24 MB, bound by the
// chain) and counted buckets over tiles of the COMPACTED list (4 launches, 85 us).  Here the tiles are chunks of 2048
// consecutive gaussian INDICES, so nothing has to be compacted or scanned beforehand and no workgroup waits for another:
//   hist     every chunk counts the buckets of its visible gaussians      -> M[bucket][chunk] = (gaussians, quantity)
//   rowscan  every bucket's row of M is scanned over the chunks (exclusive) -> row totals
//   scatter  every chunk compacts its visible gaussians (index order), ranks them by bucket (wave ballots, stable), reorders
//            them through LDS, scans their quantity in that order and writes one record {id, word, prefix, aux} per gaussian
//            at  base[bucket] + M[bucket][chunk] + rank;  plus the chunk table (first gaussian of every 1024 units of the
//            quantity) that the reference-binning emission and the row sort of the tight pipeline (k_rows.hip) start from.
// Position in (bucket, index) order = bucket base + gaussians of that bucket in earlier chunks + rank inside the chunk: the
// chunks are index ranges, so the order inside a bucket is the index order the reference's stable sort keeps.
// All three are HBM-trivial (24-50 MB); what they cost is their launches.
#include "gs_device.h"

#define GC_ITEMS 8
#define GC_THREADS 256
#define GC (GC_THREADS * GC_ITEMS) // gaussian indices per chunk
#define GBINS 1024                 // bucket = u32(min(50 depth, 999)) < 1000 (write_tile_ids.wgsl:31)
#define GS_EMIT_CHUNK_SHIFT 10     // = EMIT_CHUNK_SHIFT of k_binning.hip
// The (bucket, chunk) table is stored [bucket / 8][chunk][bucket % 8]: the 1024 entries a chunk's workgroup writes (hist) or
// reads (scatter) are 128 whole 64-byte sectors instead of 1024 partial ones, and a bucket's row is still a strided stream.
__device__ __forceinline__ uint64_t m_index(uint32_t b, uint32_t chunk, uint32_t NT) { return ((uint64_t)(b >> 3) * NT + chunk) * 8u + (b & 7u); }

__device__ __forceinline__ uint32_t sat32(unsigned long long v) { return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v; }

__global__ __launch_bounds__(GC_THREADS) void gs_gsort_hist_kernel(const uint32_t* __restrict__ words, uint32_t n, uint2* __restrict__ M, uint32_t NT) {
    __shared__ uint32_t s_cnt[GBINS];
    __shared__ unsigned long long s_sum[GBINS]; // a chunk's quantity can exceed 32 bits (4096 x 2^22 tiles): saturated on store
    const uint32_t chunk = blockIdx.x, tid = threadIdx.x;
    for (uint32_t b = tid; b < GBINS; b += GC_THREADS) { s_cnt[b] = 0u; s_sum[b] = 0ull; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < GC_ITEMS; ++j) {
        const uint32_t k = chunk * GC + j * GC_THREADS + tid;
        if (k < n) {
            const uint32_t w = words[k];
            if (w & GS_COUNT_MASK) {
                const uint32_t b = w >> GS_COUNT_BITS;
                atomicAdd(&s_cnt[b], 1u);
                atomicAdd(&s_sum[b], (unsigned long long)(w & GS_COUNT_MASK));
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < GBINS / GC_THREADS; ++i) { // thread t: buckets 4t .. 4t+3 (32 contiguous bytes)
        const uint32_t b = tid * (GBINS / GC_THREADS) + i;
        M[m_index(b, chunk, NT)] = make_uint2(s_cnt[b], sat32(s_sum[b]));
    }
}

// exclusive scan of (count, quantity) over a workgroup: each thread holds the sum of its consecutive elements in v; the
// quantity channel is 64 bits wide inside the scan and saturates where it is stored
struct GsPair { uint32_t x; unsigned long long y; };
__device__ __forceinline__ unsigned long long wave_incl_scan64(unsigned long long v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = __shfl_up((uint32_t)v, d, 64), hi = __shfl_up((uint32_t)(v >> 32), d, 64);
        if ((int)lane >= d) v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ GsPair block_excl2(GsPair v, uint32_t tid, GsPair* s_w /*[GC_THREADS / 64]*/, GsPair& total) {
    const uint32_t lane = tid & 63, w = tid >> 6;
    const uint32_t ix = wave_incl_scan(v.x, lane);
    const unsigned long long iy = wave_incl_scan64(v.y, lane);
    if (lane == 63) { s_w[w].x = ix; s_w[w].y = iy; }
    __syncthreads();
    GsPair base; base.x = 0u; base.y = 0ull;
    total.x = 0u; total.y = 0ull;
#pragma unroll
    for (int k = 0; k < GC_THREADS / 64; ++k) {
        const GsPair t = s_w[k];
        if (k < (int)w) { base.x += t.x; base.y += t.y; }
        total.x += t.x; total.y += t.y;
    }
    __syncthreads();
    GsPair r; r.x = base.x + ix - v.x; r.y = base.y + iy - v.y;
    return r;
}

// One workgroup of 1024 threads per 8 buckets (one 64-byte sector per chunk): thread (cl = t / 8, sub = t % 8) owns the chunks
// [cl * per, (cl + 1) * per) of bucket 8 * blockIdx + sub -- a dozen at 6.1 M gaussians, two batches of independent loads; the
// 128 partial sums of a bucket are scanned by lane shifts of 8 inside a wave and through LDS across the 16 waves.
#define GR_THREADS 1024
__device__ __forceinline__ void scan_stride8(uint32_t& x, unsigned long long& y, uint32_t lane) { // inclusive over lanes l, l-8, l-16, ...
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {
        const uint32_t ox = __shfl_up(x, d, 64), lo = __shfl_up((uint32_t)y, d, 64), hi = __shfl_up((uint32_t)(y >> 32), d, 64);
        if ((int)lane >= d) { x += ox; y += ((unsigned long long)hi << 32) | lo; }
    }
}
__global__ __launch_bounds__(GR_THREADS) void gs_gsort_rowscan_kernel(uint2* __restrict__ M, uint32_t NT, uint32_t n, uint2* __restrict__ rowtot) {
    __shared__ uint32_t s_x[GR_THREADS / 64][8];
    __shared__ unsigned long long s_y[GR_THREADS / 64][8];
    const uint32_t nt = (n + GC - 1) / GC, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, cl = tid >> 3, sub = tid & 7u;
    const uint32_t per = (nt + GR_THREADS / 8 - 1u) / (GR_THREADS / 8);
    const uint32_t c0 = cl * per < nt ? cl * per : nt, c1 = (cl + 1u) * per < nt ? (cl + 1u) * per : nt;
    uint2* row = M + (uint64_t)blockIdx.x * NT * 8u + sub;
    uint32_t ax = 0;
    unsigned long long ay = 0;
    for (uint32_t c = c0; c < c1; c += 8u) {
        uint2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (c + k < c1) ? row[(uint64_t)(c + k) * 8u] : make_uint2(0u, 0u);
#pragma unroll
        for (int k = 0; k < 8; ++k) { ax += v[k].x; ay += v[k].y; }
    }
    uint32_t ix = ax;
    unsigned long long iy = ay;
    scan_stride8(ix, iy, lane);
    if (lane >= 56u) { s_x[w][sub] = ix; s_y[w][sub] = iy; } // the wave's total of each of its 8 sub-buckets
    __syncthreads();
    uint32_t rx = ix - ax, tx = 0;
    unsigned long long ry = iy - ay, ty = 0;
#pragma unroll
    for (uint32_t k = 0; k < GR_THREADS / 64; ++k) {
        const uint32_t vx = s_x[k][sub];
        const unsigned long long vy = s_y[k][sub];
        if (k < w) { rx += vx; ry += vy; }
        tx += vx; ty += vy;
    }
    for (uint32_t c = c0; c < c1; c += 8u) {
        uint2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (c + k < c1) ? row[(uint64_t)(c + k) * 8u] : make_uint2(0u, 0u);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (c + k < c1) row[(uint64_t)(c + k) * 8u] = make_uint2(rx, sat32(ry));
            rx += v[k].x; ry += v[k].y;
        }
    }
    if (tid < 8u) rowtot[blockIdx.x * 8u + sub] = make_uint2(tx, sat32(ty));
}

struct GsortShared {
    uint2 base[GBINS];                 // first sorted position / first quantity offset of (bucket, this chunk)
    unsigned short binstart[GBINS];    // first slot of the bucket in the chunk's sorted order
    union {
        unsigned short whist[GC_THREADS / 64][GBINS]; // per-wave running counts while ranking ([wave][bucket]) ...
        uint32_t P[GC];                               // ... then the prefix of the sorted quantities
    } u;
    unsigned short id[GC];             // index inside the chunk
    uint32_t word[GC], aux[GC];
    GsPair w2[GC_THREADS / 64];
    uint32_t wcnt[GC_THREADS / 64];
};
static_assert(sizeof(unsigned short) * (GC_THREADS / 64) * GBINS == sizeof(uint32_t) * GC, "the ranking counters and the prefix share their storage");

// Output: ONE 16-byte record per visible gaussian in (bucket, index) order -- {gaussian id, count word, exclusive prefix of the
// quantity, aux} -- because a (chunk, bucket) run is about two gaussians long: four separate arrays were four scattered 4-byte
// stores each (round 2), a record is one 16-byte store.  aux_in (optional, tight row pipeline): a second per-gaussian word
// (the arena address of its row-item slots).  chunk_table[c] = the gaussian whose quantity interval holds c * 1024 (the
// reference-binning emission and the row sort start there).  tot_*: visible gaussians, total quantity (saturated).
// 38 KB of LDS: four workgroups per CU; every global load of the workgroup is issued before the first barrier.
__global__ __launch_bounds__(GC_THREADS) void gs_gsort_scatter_kernel(const uint32_t* __restrict__ words, const uint32_t* __restrict__ aux_in, uint32_t n,
                                                                      const uint2* __restrict__ M, uint32_t NT, const uint2* __restrict__ rowtot,
                                                                      uint4* __restrict__ grec, uint32_t* __restrict__ chunk_table,
                                                                      uint32_t chunk_cap, uint32_t* __restrict__ tot_visible,
                                                                      uint32_t* __restrict__ tot_quantity) {
    __shared__ GsortShared S;
    const uint32_t chunk = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    constexpr uint32_t BPT = GBINS / GC_THREADS; // buckets per thread (4 consecutive)

    // ---- every global load up front: the chunk's count words (+ aux), the row totals, this chunk's entries of the table ----
    uint32_t wv[GC_ITEMS], av[GC_ITEMS];
    const uint32_t e0 = chunk * GC + w * (64 * GC_ITEMS) + lane; // wave w reads indices w * 512 + j * 64 + lane
#pragma unroll
    for (int j = 0; j < GC_ITEMS; ++j) {
        const uint32_t k = e0 + j * 64;
        wv[j] = (k < n) ? words[k] : 0u;
        av[j] = (aux_in && k < n) ? aux_in[k] : 0u;
    }
    uint2 t[BPT], mm[BPT];
#pragma unroll
    for (uint32_t i = 0; i < BPT; ++i) { t[i] = rowtot[tid * BPT + i]; mm[i] = M[m_index(tid * BPT + i, chunk, NT)]; }

    // ---- bucket bases: exclusive scan of the row totals, plus this chunk's entry of the scanned table ----
    {
        GsPair acc; acc.x = 0u; acc.y = 0ull;
#pragma unroll
        for (uint32_t i = 0; i < BPT; ++i) { acc.x += t[i].x; acc.y += t[i].y; }
        GsPair total;
        GsPair run = block_excl2(acc, tid, S.w2, total);
#pragma unroll
        for (uint32_t i = 0; i < BPT; ++i) {
            S.base[tid * BPT + i] = make_uint2(run.x + mm[i].x, sat32(run.y + mm[i].y));
            run.x += t[i].x; run.y += t[i].y;
        }
        if (chunk == 0 && tid == 0) { *tot_visible = total.x; *tot_quantity = sat32(total.y); }
    }
    for (uint32_t k = lane; k < GBINS; k += 64) S.u.whist[w][k] = 0;

    // ---- the chunk's visible gaussians, compacted in index order ----
    uint32_t wave_vis = 0;
#pragma unroll
    for (int j = 0; j < GC_ITEMS; ++j) {
        if (!(wv[j] & GS_COUNT_MASK)) wv[j] = 0u;
        wave_vis += (uint32_t)__popcll(__ballot(wv[j] != 0u));
    }
    if (lane == 0) S.wcnt[w] = wave_vis;
    __syncthreads();
    uint32_t cbase = 0, nc = 0;
#pragma unroll
    for (int k = 0; k < GC_THREADS / 64; ++k) { if (k < (int)w) cbase += S.wcnt[k]; nc += S.wcnt[k]; }
#pragma unroll
    for (int j = 0; j < GC_ITEMS; ++j) {
        const unsigned long long bal = __ballot(wv[j] != 0u);
        if (wv[j]) {
            const uint32_t p = cbase + (uint32_t)__popcll(bal & lt_mask);
            S.id[p] = (unsigned short)(w * (64 * GC_ITEMS) + j * 64 + lane);
            S.word[p] = wv[j];
            S.aux[p] = av[j];
        }
        cbase += (uint32_t)__popcll(bal);
    }
    for (uint32_t p = nc + tid; p < GC; p += GC_THREADS) S.word[p] = 0xFFFFFFFFu; // absent: bucket 1023, sorts behind every real one, never stored
    __syncthreads();
    if (nc == 0) return; // (uniform)

    // ---- rank by bucket: wave w owns the compacted positions [w*Q, w*Q + Q), item j of lane l = w*Q + j*64 + l ----
    const uint32_t T = (nc + GC_THREADS - 1) / GC_THREADS; // items per thread (uniform over the workgroup)
    const uint32_t Q = T * 64;
    uint32_t gid[GC_ITEMS], wd[GC_ITEMS], ax[GC_ITEMS];
    uint32_t rank2[GC_ITEMS / 2];
#pragma unroll
    for (int j = 0; j < GC_ITEMS; ++j) {
        const uint32_t p = w * Q + j * 64 + lane;
        const bool in = (uint32_t)j < T; // (p < GC then: T * 256 <= GC)
        gid[j] = in ? (uint32_t)S.id[p] : 0u;
        wd[j] = in ? S.word[p] : 0xFFFFFFFFu;
        ax[j] = in ? S.aux[p] : 0u;
    }
    // peers = lanes holding the same bucket (10 ballots), order = (item, lane): stable
#pragma unroll
    for (int j = 0; j < GC_ITEMS; ++j) {
        if ((uint32_t)j < T) {
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
            const uint32_t pre = S.u.whist[w][d];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // every peer has read `pre` before the leader's store (one wave, in-order LDS)
            if (below == 0) S.u.whist[w][d] = (unsigned short)(pre + cnt);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t r = pre + below; // < 2048
            if (j & 1) rank2[j >> 1] |= r << 16;
            else rank2[j >> 1] = r;
        }
    }
    __syncthreads();
    // ---- per-wave counts -> exclusive across waves; bucket starts inside the chunk ----
    {
        uint32_t tot[BPT], acc = 0;
#pragma unroll
        for (uint32_t i = 0; i < BPT; ++i) {
            const uint32_t b = tid * BPT + i;
            uint32_t run = 0;
#pragma unroll
            for (int k = 0; k < GC_THREADS / 64; ++k) { const uint32_t c = S.u.whist[k][b]; S.u.whist[k][b] = (unsigned short)run; run += c; }
            tot[i] = run;
            acc += run;
        }
        GsPair total, in;
        in.x = acc; in.y = 0ull;
        const GsPair ex = block_excl2(in, tid, S.w2, total);
        uint32_t run = ex.x;
#pragma unroll
        for (uint32_t i = 0; i < BPT; ++i) { S.binstart[tid * BPT + i] = (unsigned short)run; run += tot[i]; }
    }
    __syncthreads();
    // ---- reorder through LDS (every thread holds its items in registers: the arrays can be overwritten) ----
    uint32_t mypos[GC_ITEMS];
#pragma unroll
    for (int j = 0; j < GC_ITEMS; ++j) {
        mypos[j] = 0xFFFFFFFFu;
        if ((uint32_t)j < T) {
            const uint32_t d = wd[j] >> GS_COUNT_BITS;
            const uint32_t r = (j & 1) ? (rank2[j >> 1] >> 16) : (rank2[j >> 1] & 0xFFFFu);
            mypos[j] = (uint32_t)S.binstart[d] + (uint32_t)S.u.whist[w][d] + r;
        }
    }
    __syncthreads(); // every wave has read the ranking counters: the prefix below overwrites them
#pragma unroll
    for (int j = 0; j < GC_ITEMS; ++j) {
        if ((uint32_t)j < T) {
            S.id[mypos[j]] = (unsigned short)gid[j];
            S.word[mypos[j]] = wd[j];
            S.aux[mypos[j]] = ax[j];
        }
    }
    __syncthreads();
    // ---- exclusive scan of the quantities in sorted order (thread t: slots 8t .. 8t+7) ----
    {
        uint32_t c[GC_ITEMS];
        unsigned long long acc = 0;
#pragma unroll
        for (int j = 0; j < GC_ITEMS; ++j) {
            const uint32_t p = tid * GC_ITEMS + j;
            const uint32_t x = (p < T * GC_THREADS) ? S.word[p] : 0xFFFFFFFFu;
            c[j] = x == 0xFFFFFFFFu ? 0u : (x & GS_COUNT_MASK);
            acc += c[j];
        }
        GsPair total, in;
        in.x = 0u; in.y = acc;
        const GsPair ex = block_excl2(in, tid, S.w2, total);
        unsigned long long run = ex.y;
#pragma unroll
        for (int j = 0; j < GC_ITEMS; ++j) { S.u.P[tid * GC_ITEMS + j] = sat32(run); run += c[j]; }
    }
    __syncthreads();
    // ---- store: coalesced over the sorted slots ----
#pragma unroll
    for (int j = 0; j < GC_ITEMS; ++j) {
        const uint32_t pos = j * GC_THREADS + tid;
        if (pos >= nc) continue;
        const uint32_t x = S.word[pos];
        const uint32_t b = x >> GS_COUNT_BITS, first = S.binstart[b];
        const uint2 base = S.base[b];
        const uint32_t g = base.x + (pos - first);
        const unsigned long long off64 = (unsigned long long)base.y + (S.u.P[pos] - S.u.P[first]);
        const uint32_t off = sat32(off64);
        grec[g] = make_uint4(chunk * GC + (uint32_t)S.id[pos], x, off, S.aux[pos]);
        const uint32_t cnt = x & GS_COUNT_MASK; // > 0: only visible gaussians are here
        if (off != 0xFFFFFFFFu) {
            const unsigned long long lastq = (off64 + cnt - 1ull) >> GS_EMIT_CHUNK_SHIFT;
            const uint32_t last = lastq > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)lastq;
            for (uint32_t c = (uint32_t)((off64 + (1u << GS_EMIT_CHUNK_SHIFT) - 1ull) >> GS_EMIT_CHUNK_SHIFT); c <= last && c < chunk_cap; ++c) chunk_table[c] = g;
        }
    }
}

// ---- host launchers --------------------------------------------------------------------------------
uint32_t gs_gsort_tiles(uint32_t n) { return (n + GC - 1) / GC; }
uint64_t gs_gsort_scratch_bytes(uint32_t n) { return ((uint64_t)GBINS * gs_gsort_tiles(n ? n : 1) + GBINS) * sizeof(uint2); }
// words: one word per gaussian INDEX (quantity in the low 22 bits, depth bucket in the high 10; 0 = not visible); scratch:
// gs_gsort_scratch_bytes(n) bytes.  Output records in (bucket, index) order; tot_visible / tot_quantity: device words.
void gs_launch_gsort(const uint32_t* words, const uint32_t* aux_in, uint32_t n, void* scratch, void* grec, uint32_t* chunk_table, uint32_t chunk_cap,
                     uint32_t* tot_visible, uint32_t* tot_quantity, hipStream_t st) {
    if (!n) return;
    const uint32_t NT = gs_gsort_tiles(n);
    uint2* M = (uint2*)scratch;
    uint2* rowtot = M + (uint64_t)GBINS * NT;
    hipLaunchKernelGGL(gs_gsort_hist_kernel, dim3(NT), dim3(GC_THREADS), 0, st, words, n, M, NT);
    hipLaunchKernelGGL(gs_gsort_rowscan_kernel, dim3(GBINS / 8), dim3(GR_THREADS), 0, st, M, NT, n, rowtot);
    hipLaunchKernelGGL(gs_gsort_scatter_kernel, dim3(NT), dim3(GC_THREADS), 0, st, words, aux_in, n, (const uint2*)M, NT, (const uint2*)rowtot,
                       (uint4*)grec, chunk_table, chunk_cap, tot_visible, tot_quantity);
}
