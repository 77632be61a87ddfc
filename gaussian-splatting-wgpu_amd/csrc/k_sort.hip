// k_sort.hip -- stable LSD radix sort of (key,value) u32 pairs, 8-bit digits, one sweep per digit.
//
// Replaces GPUSorter (reference src/radix_sort/sort.ts:52-363) and the five entry points of
// src/radix_sort/radix_sort.wgsl (zero_histograms :47-81, calculate_histogram :127-137,
// prefix_histogram :171-189, scatter_even/odd :451-498).  Semantics kept: stable, ascending by the
// full key, equal keys keep emission order.  What changed, for CDNA4:
//   * wave64 ballots compute the match mask directly (the reference emulates a 32-wide subgroup
//     match with 32 workgroup-memory round trips per key, radix_sort.wgsl:263-288);
//   * passes whose digit is zero for every possible key are skipped (the reference always runs 4);
//   * tiles are handed out by an atomic ticket, so a tile only ever waits on tiles that have
//     already started (forward progress for the decoupled look-back), and the grid is persistent:
//     its size does not depend on I, which therefore never has to visit the host;
//   * look-back words are single 4-byte granules {flag:2,count:30} stored/polled with relaxed
//     agent-scope atomics (sc1) -- the "data is the flag" form; no fences, no stale L1 lines;
//   * keys are re-ordered through LDS before the global scatter so each digit's run leaves the
//     workgroup as contiguous stores.
// HBM-bound: histogram 4 B/key once, each sweep 8 B/key read + 8 B/key written.
#include "gs_device.h"

#define RS_ITEMS 16
#define RS_WAVES 8 // waves per workgroup: a tile is RS_WAVES * 64 * RS_ITEMS keys
#define RS_THREADS (RS_WAVES * 64)
#define RS_TILE (RS_THREADS * RS_ITEMS)
#define RS_AGG (1u << 30)
#define RS_PREFIX (2u << 30)
#define RS_FLAGS (3u << 30)
#define RS_VALUE (~RS_FLAGS)

// Digit of pass `pass`: `bits` bits of the sort word, which is the key itself (mode 0) or the key's tile id
// key/1000 (mode 1: the depth-ordered pipeline sorts instances by tile only, see gs_runtime.hip).
struct SortDigits {
    uint32_t bits;    // bits per pass (<= 8)
    uint32_t by_tile; // 0: word = key, 1: word = key / 1000 (u16 tile-id keys are their own word: mode 0)
};
__device__ __forceinline__ uint32_t sort_digit(uint32_t key, uint32_t pass, const SortDigits& sd) {
    // pad keys (0xFFFFFFFF, only in a tile's tail) must keep the largest digit in every pass so that they rank last
    const uint32_t word = (sd.by_tile && key != 0xFFFFFFFFu) ? key / 1000u : key;
    return (word >> (pass * sd.bits)) & ((1u << sd.bits) - 1u);
}

// ---- digit histograms of all passes in one read of the keys ---------------------------------------
__global__ __launch_bounds__(256) void gs_sort_hist_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ hist /*[passes][256]*/,
                                                            const uint32_t* __restrict__ n_ptr, uint32_t capacity,
                                                            uint32_t passes, SortDigits sd) {
    __shared__ uint32_t s_h[4][256];
    for (uint32_t k = threadIdx.x; k < 4 * 256; k += 256) (&s_h[0][0])[k] = 0;
    __syncthreads();
    uint32_t n = *n_ptr;
    if (n > capacity) n = capacity;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    const uint64_t nvec = n / 4;
    auto add = [&](uint32_t k) {
        const uint32_t word = sd.by_tile ? k / 1000u : k;
        const uint32_t mask = (1u << sd.bits) - 1u;
        atomicAdd(&s_h[0][word & mask], 1u);
        if (passes > 1) atomicAdd(&s_h[1][(word >> sd.bits) & mask], 1u);
        if (passes > 2) atomicAdd(&s_h[2][(word >> (2 * sd.bits)) & mask], 1u);
        if (passes > 3) atomicAdd(&s_h[3][(word >> (3 * sd.bits)) & mask], 1u);
    };
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        const uint4 q = reinterpret_cast<const uint4*>(keys)[i];
        add(q.x); add(q.y); add(q.z); add(q.w);
    }
    for (uint64_t i = nvec * 4 + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) add(keys[i]);
    __syncthreads();
    for (uint32_t p = 0; p < passes; ++p) {
        const uint32_t c = s_h[p][threadIdx.x];
        if (c) atomicAdd(&hist[p * 256 + threadIdx.x], c);
    }
}

// ---- one digit sweep ----------------------------------------------------------------------------------
struct SweepShared {
    uint32_t hist[RS_WAVES][256];  // per-wave digit counts -> exclusive offsets across waves -> + digit start
    uint32_t gbase[256];    // global address of slot 0 of each digit's run, minus the digit's first slot
    uint32_t keys[RS_TILE];
    uint32_t vals[RS_TILE];
    uint32_t wsum[4]; // inclusive digit-total sums of the four waves that own the 256 digits
    uint32_t tot[256]; // early digit counts of the tile
    uint32_t tile;
};

// KT: the keys' type in memory.  uint16_t = the depth-ordered pipeline's instance arrays, whose sort word is the tile id
// alone (< 65535; the full key tile*1000 + bucket is rebuilt only for the tap): 12 instead of 16 bytes moved per pair
// and sweep.  Registers and LDS hold the widened word; pads (0xFFFFFFFF) exist only there.
template <bool FULL, typename KT>
__device__ __forceinline__ void sweep_tile(SweepShared& sh, const KT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                           KT* __restrict__ keys_out, uint32_t* __restrict__ vals_out, GsControl* ctl,
                                           const uint32_t* __restrict__ hist, uint32_t pass, SortDigits sd, uint32_t* status,
                                           uint32_t tile, uint32_t valid, const uint32_t* __restrict__ aux_table,
                                           uint32_t* __restrict__ aux_out) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t wbase = w * (64 * RS_ITEMS) + lane; // this lane's first slot in the tile
    const KT* kp = keys_in + (uint64_t)tile * RS_TILE + wbase;
    const uint32_t* vp = vals_in + (uint64_t)tile * RS_TILE + wbase;

    uint32_t key[RS_ITEMS];
    uint32_t rank2[RS_ITEMS / 2]; // two 16-bit in-wave ranks per register
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        if (FULL) key[j] = kp[j * 64];
        else key[j] = (wbase + j * 64 < valid) ? kp[j * 64] : 0xFFFFFFFFu; // pads sort last and are never stored
    }
    for (uint32_t k = lane; k < 256; k += 64) sh.hist[w][k] = 0;
    // Early aggregate: the tile's digit counts are cheap (16 LDS atomics per thread) next to the ranking below, and they
    // are all a successor needs from this tile.  Publishing them BEFORE the ranking puts the ranking time between "my
    // aggregate is visible" and "I look at my predecessors'", so the look-back rarely finds an unpublished word.
    if (tid < 256u) sh.tot[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) atomicAdd(&sh.tot[sort_digit(key[j], pass, sd)], 1u);
    __syncthreads();
    if (tid < 256u) {
        const uint32_t c = sh.tot[tid]; // (pads only exist in the last tile, whose counts no tile reads)
        st_agent(status + (uint64_t)tile * 256 + tid, (tile == 0 ? RS_PREFIX : RS_AGG) | c);
    }

    // rank inside the wave: peers = lanes holding the same digit (8 ballots), order = (item, lane)
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const uint32_t d = sort_digit(key[j], pass, sd);
        uint32_t plo = 0xFFFFFFFFu, phi = 0xFFFFFFFFu;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint32_t bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit != 0u);
            const uint32_t inv = bit - 1u; // 0 when the bit is set, ~0 when clear: peers &= bit ? bal : ~bal
            plo &= (uint32_t)bal ^ inv;
            phi &= (uint32_t)(bal >> 32) ^ inv;
        }
        const uint32_t below = __popc(plo & (uint32_t)lt_mask) + __popc(phi & (uint32_t)(lt_mask >> 32));
        const uint32_t cnt = __popc(plo) + __popc(phi);
        const uint32_t pre = sh.hist[w][d];
        // every peer has read `pre` before the leader's store is issued: one wave, in-order LDS
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (below == 0) sh.hist[w][d] = pre + cnt;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t r = pre + below; // < 1024
        if (j & 1) rank2[j >> 1] |= r << 16;
        else rank2[j >> 1] = r;
    }
    __syncthreads();

    // thread d (d < 256): counts of digit d per wave -> exclusive offsets across waves, tile total
    uint32_t cw[RS_WAVES], total = 0, incl = 0, excl = 0;
    const bool owner = tid < 256u;
    if (owner) {
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k) { cw[k] = sh.hist[k][tid]; total += cw[k]; }
        incl = wave_incl_scan(total, lane);
        if (lane == 63) sh.wsum[w] = incl;

        // publish this tile's digit count, then walk back over the predecessors' words
        uint32_t* my = status + (uint64_t)tile * 256 + tid;
        if (tile > 0) { // (the aggregate, or tile 0's prefix, was published before the ranking)
            // Walk back over the predecessors' words LB at a time: the loads of one round are independent and in
            // flight together, so a walk of k tiles costs ~k/LB L2 round trips instead of k (all resident
            // workgroups start together, so the first tiles of a launch walk back hundreds of tiles).
            constexpr int LB = 8;
            bool found = false;
            for (int t = (int)tile - 1; t >= 0 && !found; t -= LB) {
                uint32_t sv[LB];
#pragma unroll
                for (int k = 0; k < LB; ++k) sv[k] = (t - k >= 0) ? ld_agent(status + (uint64_t)(t - k) * 256 + tid) : RS_PREFIX;
#pragma unroll
                for (int k = 0; k < LB; ++k) {
                    if (found) break;
                    uint32_t v = sv[k], spins = 0;
                    while ((v & RS_FLAGS) == 0 && ++spins < GS_SPIN_LIMIT) { // not published yet: poll this one word
                        __builtin_amdgcn_s_sleep(1);
                        v = ld_agent(status + (uint64_t)(t - k) * 256 + tid);
                    }
                    if ((v & RS_FLAGS) == 0) { ctl->fault = 1u; found = true; break; }
                    excl += v & RS_VALUE;
                    if ((v & RS_FLAGS) == RS_PREFIX) found = true;
                }
            }
            st_agent(my, RS_PREFIX | ((excl + total) & RS_VALUE));
        }
    }
    __syncthreads();
    if (owner) {
        uint32_t wv = 0;
        for (uint32_t k = 0; k < w; ++k) wv += sh.wsum[k];
        uint32_t run = wv + incl - total; // first slot of digit `tid` in the tile's sorted order
        sh.gbase[tid] = hist[tid] + excl - run;
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k) { sh.hist[k][tid] = run; run += cw[k]; }
    }
    __syncthreads();

    // reorder through LDS, then store each digit's run contiguously (payloads are only loaded now:
    // holding them across the ranking phase costs 16 VGPRs and a wave of occupancy)
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        uint32_t v;
        if (FULL) v = vp[j * 64];
        else v = (wbase + j * 64 < valid) ? vp[j * 64] : 0u;
        const uint32_t d = sort_digit(key[j], pass, sd);
        const uint32_t r = (j & 1) ? (rank2[j >> 1] >> 16) : (rank2[j >> 1] & 0xFFFFu);
        const uint32_t pos = sh.hist[w][d] + r;
        sh.keys[pos] = key[j];
        sh.vals[pos] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const uint32_t pos = j * RS_THREADS + tid;
        if (FULL || pos < valid) {
            const uint32_t k = sh.keys[pos];
            const uint32_t g = sh.gbase[sort_digit(k, pass, sd)] + pos;
            const uint32_t pv = sh.vals[pos];
            keys_out[g] = (KT)k;
            vals_out[g] = pv;
            if (aux_out) aux_out[g] = aux_table[pv]; // last sweep of the gaussian-level sort: tile counts in sorted order
        }
    }
}

template <typename KT>
__global__ __launch_bounds__(RS_THREADS, 4) void gs_sort_sweep_kernel(const KT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                             KT* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                             GsControl* ctl, uint32_t* __restrict__ ticket, const uint32_t* __restrict__ hist,
                                                             const uint32_t* __restrict__ n_ptr, uint32_t capacity, uint32_t pass,
                                                             SortDigits sd, uint32_t* status, const uint32_t* __restrict__ aux_table,
                                                             uint32_t* __restrict__ aux_out) {
    __shared__ SweepShared sh;
    __shared__ uint32_t s_dbase[256]; // exclusive scan of this pass's digit histogram (first slot of every digit's run)
    uint32_t n = *n_ptr;
    if (n > capacity) n = capacity;
    const uint32_t ntiles = (n + RS_TILE - 1) / RS_TILE;
    {   // every workgroup scans the 256 raw counts itself: cheaper than a one-workgroup kernel between launches
        const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
        uint32_t c = 0, incl = 0;
        if (tid < 256u) {
            c = hist[tid];
            incl = wave_incl_scan(c, lane);
            if (lane == 63) sh.wsum[w] = incl;
        }
        __syncthreads();
        if (tid < 256u) {
            uint32_t b = 0;
            for (uint32_t k = 0; k < w; ++k) b += sh.wsum[k];
            s_dbase[tid] = b + incl - c;
        }
        __syncthreads();
    }
    for (;;) {
        if (threadIdx.x == 0) sh.tile = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t tile = sh.tile;
        if (tile >= ntiles) break; // uniform: every thread read the same ticket
        const uint32_t valid = (n - tile * RS_TILE < RS_TILE) ? n - tile * RS_TILE : RS_TILE;
        if (valid == RS_TILE) sweep_tile<true, KT>(sh, keys_in, vals_in, keys_out, vals_out, ctl, s_dbase, pass, sd, status, tile, valid, aux_table, aux_out);
        else sweep_tile<false, KT>(sh, keys_in, vals_in, keys_out, vals_out, ctl, s_dbase, pass, sd, status, tile, valid, aux_table, aux_out);
        __syncthreads(); // LDS is reused by the next tile
    }
}

// ---- host launchers --------------------------------------------------------------------------------
uint32_t gs_sort_tiles(uint64_t capacity) { return (uint32_t)((capacity + RS_TILE - 1) / RS_TILE); }
// Sorts `n` (device word *n_ptr) pairs by `passes` digits of `bits` bits of the sort word (the key, or key/1000 when
// by_tile).  Returns in *out_keys/*out_vals which of the two buffer pairs holds the result.  tickets[passes], hist[passes*256]
// and status (passes * gs_sort_tiles(capacity) * 256 words) must have been zeroed by the caller.
// have_hist: the caller already accumulated the digit counts into hist (the scan does it for the gaussian-level sort).
// aux_table/aux_out (optional): the last sweep also writes aux_out[i] = aux_table[value_i] in sorted order.
// keys16: the key arrays hold uint16_t sort words (by_tile must be 0 and have_hist true).
void gs_launch_sort(uint32_t* keysA, uint32_t* valsA, uint32_t* keysB, uint32_t* valsB, GsControl* ctl, uint32_t* tickets, uint32_t* hist,
                    const uint32_t* n_ptr, uint32_t capacity, uint32_t passes, uint32_t bits, uint32_t by_tile, uint32_t* status,
                    uint32_t grid, bool have_hist, const uint32_t* aux_table, uint32_t* aux_out, hipStream_t st, uint32_t** out_keys,
                    uint32_t** out_vals, bool keys16) {
    SortDigits sd;
    sd.bits = bits;
    sd.by_tile = by_tile;
    if (!have_hist) hipLaunchKernelGGL(gs_sort_hist_kernel, dim3(grid), dim3(256), 0, st, keysA, hist, n_ptr, capacity, passes, sd);
    const uint64_t per_pass = (uint64_t)gs_sort_tiles(capacity) * 256;
    uint32_t *ki = keysA, *vi = valsA, *ko = keysB, *vo = valsB;
    for (uint32_t p = 0; p < passes; ++p) {
        const bool last = (p + 1 == passes);
        if (keys16)
            hipLaunchKernelGGL(gs_sort_sweep_kernel<uint16_t>, dim3(grid), dim3(RS_THREADS), 0, st, (const uint16_t*)ki, vi, (uint16_t*)ko, vo, ctl,
                               tickets + p, hist + p * 256, n_ptr, capacity, p, sd, status + p * per_pass,
                               last ? aux_table : (const uint32_t*)nullptr, last ? aux_out : (uint32_t*)nullptr);
        else
            hipLaunchKernelGGL(gs_sort_sweep_kernel<uint32_t>, dim3(grid), dim3(RS_THREADS), 0, st, ki, vi, ko, vo, ctl, tickets + p, hist + p * 256,
                               n_ptr, capacity, p, sd, status + p * per_pass, last ? aux_table : (const uint32_t*)nullptr,
                               last ? aux_out : (uint32_t*)nullptr);
        uint32_t* t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    *out_keys = ki;
    *out_vals = vi;
}
