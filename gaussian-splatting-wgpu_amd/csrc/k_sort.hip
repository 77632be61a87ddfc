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
#define RS_TILE (256 * RS_ITEMS)
#define RS_AGG (1u << 30)
#define RS_PREFIX (2u << 30)
#define RS_FLAGS (3u << 30)
#define RS_VALUE (~RS_FLAGS)

// ---- digit histograms of all passes in one read of the keys ---------------------------------------
__global__ __launch_bounds__(256) void gs_sort_hist_kernel(const uint32_t* __restrict__ keys, GsControl* ctl,
                                                            const uint32_t* __restrict__ n_ptr, uint32_t capacity,
                                                            uint32_t passes) {
    __shared__ uint32_t s_h[4][256];
    for (uint32_t k = threadIdx.x; k < 4 * 256; k += 256) (&s_h[0][0])[k] = 0;
    __syncthreads();
    uint32_t n = *n_ptr;
    if (n > capacity) n = capacity;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint32_t k = keys[i];
        atomicAdd(&s_h[0][k & 255u], 1u);
        if (passes > 1) atomicAdd(&s_h[1][(k >> 8) & 255u], 1u);
        if (passes > 2) atomicAdd(&s_h[2][(k >> 16) & 255u], 1u);
        if (passes > 3) atomicAdd(&s_h[3][k >> 24], 1u);
    }
    __syncthreads();
    for (uint32_t p = 0; p < passes; ++p) {
        const uint32_t c = s_h[p][threadIdx.x];
        if (c) atomicAdd(&ctl->hist[p][threadIdx.x], c);
    }
}

// ---- exclusive scan of each 256-bin histogram (one workgroup; thread d owns bin d) ----------------
__global__ __launch_bounds__(256) void gs_sort_hist_scan_kernel(GsControl* ctl, uint32_t passes) {
    __shared__ uint32_t s_w[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (uint32_t p = 0; p < passes; ++p) {
        const uint32_t c = ctl->hist[p][tid];
        const uint32_t incl = wave_incl_scan(c, lane);
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t k = 0; k < w; ++k) base += s_w[k];
        ctl->hist[p][tid] = base + incl - c;
        __syncthreads();
    }
}

// ---- one digit sweep ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gs_sort_sweep_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                             uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                             GsControl* ctl, const uint32_t* __restrict__ n_ptr, uint32_t capacity,
                                                             uint32_t pass, uint32_t* status) {
    __shared__ uint32_t s_hist[4][256];  // per-wave digit counts, then exclusive offsets across waves
    __shared__ uint32_t s_dstart[256];   // first slot of each digit in the tile's sorted order
    __shared__ uint32_t s_gbase[256];    // global address of slot 0 of each digit's run, minus s_dstart
    __shared__ uint32_t s_keys[RS_TILE];
    __shared__ uint32_t s_vals[RS_TILE];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_tile;

    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t shift = pass * 8;
    uint32_t n = *n_ptr;
    if (n > capacity) n = capacity;
    const uint32_t ntiles = (n + RS_TILE - 1) / RS_TILE;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    for (;;) {
        if (tid == 0) s_tile = atomicAdd(&ctl->sort_ticket[pass], 1u);
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= ntiles) break; // uniform: every thread read the same ticket
        const uint32_t tile_base = tile * RS_TILE;
        const uint32_t valid = (n - tile_base < RS_TILE) ? n - tile_base : RS_TILE;

        uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const uint32_t li = w * (64 * RS_ITEMS) + j * 64 + lane;
            const bool ok = li < valid;
            key[j] = ok ? keys_in[tile_base + li] : 0xFFFFFFFFu; // pads sort last and are never stored
            val[j] = ok ? vals_in[tile_base + li] : 0u;
        }
        for (uint32_t k = lane; k < 256; k += 64) s_hist[w][k] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // rank inside the wave: peers = lanes holding the same digit (8 ballots), order = (item, lane)
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const uint32_t d = (key[j] >> shift) & 255u;
            unsigned long long peers = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
            const uint32_t below = (uint32_t)__popcll(peers & lt_mask);
            const uint32_t cnt = (uint32_t)__popcll(peers);
            const uint32_t pre = s_hist[w][d];
            // every peer has read `pre` before the leader's store is issued: one wave, in-order LDS
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (below == 0) s_hist[w][d] = pre + cnt;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            rank[j] = pre + below;
        }
        __syncthreads();

        // thread d: counts of digit d per wave -> exclusive offsets across waves, tile total
        uint32_t total;
        {
            const uint32_t c0 = s_hist[0][tid], c1 = s_hist[1][tid], c2 = s_hist[2][tid], c3 = s_hist[3][tid];
            s_hist[0][tid] = 0;
            s_hist[1][tid] = c0;
            s_hist[2][tid] = c0 + c1;
            s_hist[3][tid] = c0 + c1 + c2;
            total = c0 + c1 + c2 + c3;
        }
        const uint32_t incl = wave_incl_scan(total, lane);
        if (lane == 63) s_wsum[w] = incl;

        // publish this tile's digit count, then walk back over the predecessors' words
        uint32_t* my = status + (uint64_t)tile * 256 + tid;
        uint32_t excl = 0;
        if (tile == 0) {
            st_agent(my, RS_PREFIX | total);
        } else {
            st_agent(my, RS_AGG | total);
            for (int t = (int)tile - 1; t >= 0; --t) {
                const uint32_t* p = status + (uint64_t)t * 256 + tid;
                uint32_t sv, spins = 0;
                do {
                    sv = ld_agent(p);
                    if (sv & RS_FLAGS) break;
                    __builtin_amdgcn_s_sleep(1);
                } while (++spins < GS_SPIN_LIMIT);
                if ((sv & RS_FLAGS) == 0) { ctl->fault = 1u; break; }
                excl += sv & RS_VALUE;
                if ((sv & RS_FLAGS) == RS_PREFIX) break;
            }
            st_agent(my, RS_PREFIX | ((excl + total) & RS_VALUE));
        }
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t k = 0; k < w; ++k) wbase += s_wsum[k];
        const uint32_t dstart = wbase + incl - total;
        s_dstart[tid] = dstart;
        s_gbase[tid] = ctl->hist[pass][tid] + excl - dstart;
        __syncthreads();

        // reorder through LDS, then store each digit's run contiguously
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const uint32_t d = (key[j] >> shift) & 255u;
            const uint32_t pos = s_dstart[d] + s_hist[w][d] + rank[j];
            s_keys[pos] = key[j];
            s_vals[pos] = val[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const uint32_t pos = j * 256 + tid;
            if (pos < valid) {
                const uint32_t k = s_keys[pos];
                const uint32_t g = s_gbase[(k >> shift) & 255u] + pos;
                keys_out[g] = k;
                vals_out[g] = s_vals[pos];
            }
        }
        __syncthreads(); // LDS is reused by the next tile
    }
}

// ---- host launchers --------------------------------------------------------------------------------
uint32_t gs_sort_tiles(uint64_t capacity) { return (uint32_t)((capacity + RS_TILE - 1) / RS_TILE); }
// Sorts `n` (device word *n_ptr) pairs; `passes` 8-bit digits starting at bit 0.  Returns in *out_keys/*out_vals which
// of the two buffer pairs holds the result.  status: passes * gs_sort_tiles(capacity) * 256 words, zeroed by the caller;
// ctl->hist and ctl->sort_ticket zeroed by the caller.
void gs_launch_sort(uint32_t* keysA, uint32_t* valsA, uint32_t* keysB, uint32_t* valsB, GsControl* ctl, const uint32_t* n_ptr,
                    uint32_t capacity, uint32_t passes, uint32_t* status, uint32_t grid, hipStream_t st, uint32_t** out_keys,
                    uint32_t** out_vals) {
    hipLaunchKernelGGL(gs_sort_hist_kernel, dim3(grid), dim3(256), 0, st, keysA, ctl, n_ptr, capacity, passes);
    hipLaunchKernelGGL(gs_sort_hist_scan_kernel, dim3(1), dim3(256), 0, st, ctl, passes);
    const uint64_t per_pass = (uint64_t)gs_sort_tiles(capacity) * 256;
    uint32_t *ki = keysA, *vi = valsA, *ko = keysB, *vo = valsB;
    for (uint32_t p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(gs_sort_sweep_kernel, dim3(grid), dim3(256), 0, st, ki, vi, ko, vo, ctl, n_ptr, capacity, p,
                           status + p * per_pass);
        uint32_t* t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    *out_keys = ki;
    *out_vals = vi;
}
