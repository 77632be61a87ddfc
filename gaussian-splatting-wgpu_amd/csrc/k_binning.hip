// k_binning.hip -- tile binning: exclusive scan of tile counts, (key,value) emission, tile ranges.
//
// Replaces
//   ExclusiveScanner.scan + prefix_sum/block_prefix_sum/add_block_sums
//       (reference src/exclusive_scan.ts:208-325, src/prefix_sum.wgsl:6-45, src/block_prefix_sum.wgsl:6-45,
//        src/add_block_sums.wgsl:4-9): 3 dispatches per 262 144-element chunk + a blocking readback of I;
//   write_tile_ids.wgsl::main (src/write_tile_ids.wgsl:18-35): one thread loops over a whole rect;
//   compute_ranges.wgsl::main (src/compute_ranges.wgsl:5-29).
// Here: ONE single-pass chained scan (decoupled look-back, wave64 window) that leaves I on the device,
// a wave-cooperative load-balanced emission with fully coalesced key/value stores, and a ranges
// kernel that writes every tile exactly once (no per-frame memset of `ranges`).
// All three are HBM-bound integer kernels: scan 8 B/gaussian, emit 24 B/visible + 8 B/entry,
// ranges 4 B/entry + 4 B/tile.
#include "gs_device.h"

// ------------------------------------------------------------------------------------------------
// Exclusive scan, 4096 counts per workgroup, status granule = {flag:2, visible:30, sum:32} in one 8-byte
// word: the count of visible gaussians rides along the same look-back (a separate atomic counter on one
// hot word serialises at ~88 atomics/us on this chip).
// ------------------------------------------------------------------------------------------------
#define EMIT_CHUNK_SHIFT 10
#define EMIT_CHUNK (1u << EMIT_CHUNK_SHIFT) // output slots one wave emits at a time in the balanced emission
#define SCAN_ITEMS 24 // 24 576 counts per workgroup: 6.1 M gaussians are 249 workgroups, one residency round of the 256 CUs
#define SCAN_THREADS 1024 // 16 waves: the look-back chain advances 64 workgroups per step, so fewer, larger workgroups finish sooner
#define SCAN_WAVES (SCAN_THREADS / 64)
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)
#define ST_AGG (1ull << 62)
#define ST_PREFIX (2ull << 62)
#define ST_MASK (3ull << 62)

// Tile-count sums saturate at 2^32-1 instead of wrapping (saturating addition is associative, so every prefix is
// min(true sum, 2^32-1)): a frame whose intersections exceed 32 bits then reports I = 0xFFFFFFFF > capacity and
// gs_wait answers GS_ERR_CAPACITY instead of rendering a wrapped, truncated list as GS_OK.
__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b) {
    const uint32_t s = a + b;
    return s < a ? 0xFFFFFFFFu : s;
}
__device__ __forceinline__ uint32_t wave_incl_scan_sat(uint32_t v, uint32_t) {
    v = sat_add(v, GS_DPP(v, 0x111, 0xf));
    v = sat_add(v, GS_DPP(v, 0x112, 0xf));
    v = sat_add(v, GS_DPP(v, 0x114, 0xf));
    v = sat_add(v, GS_DPP(v, 0x118, 0xf));
    v = sat_add(v, GS_DPP(v, 0x142, 0xa));
    v = sat_add(v, GS_DPP(v, 0x143, 0xc));
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_sat(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_sat(v, 0u), 63);
}

// counts[] words are packed by the preprocess: tile count in the low 22 bits, depth bucket (the low part of
// the sort key, write_tile_ids.wgsl:31) in the high 10.  offsets = exclusive prefix of the tile counts in gaussian-index
// order (the reference's order); I and the number of visible gaussians go to the control block.  (Round 2 also compacted
// the visible gaussians here for the depth-ordered pipeline; that pipeline now sorts straight from the count words, k_gsort.hip.)
__global__ __launch_bounds__(SCAN_THREADS) void gs_scan_kernel(const uint32_t* __restrict__ counts, uint32_t n, uint32_t* __restrict__ offsets,
                                                       unsigned long long* status, uint32_t* ticket, GsControl* ctl) {
    __shared__ uint32_t s_bid;
    __shared__ uint32_t s_wsum[SCAN_WAVES];
    __shared__ uint32_t s_wnz[SCAN_WAVES];
    __shared__ uint32_t s_prefix[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (tid == 0) s_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t bid = s_bid;
    if (bid >= nblocks) return; // uniform; every block that can be waited on has a lower ticket
    const uint32_t base = bid * SCAN_TILE + tid * SCAN_ITEMS;

    uint32_t v[SCAN_ITEMS];
    if (base + SCAN_ITEMS <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(counts + base);
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS / 4; ++j) {
            const uint4 q = p[j];
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; ++j) v[j] = (base + j < n) ? counts[base + j] : 0u;
    }
    uint32_t tsum = 0, tnz = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { tsum += v[j] & GS_COUNT_MASK; tnz += ((v[j] & GS_COUNT_MASK) != 0u); }
    const uint32_t incl = wave_incl_scan_sat(tsum, lane); // a thread's 24 counts (< 2^22 each) cannot wrap; wider sums saturate
    const uint32_t incl_nz = wave_incl_scan(tnz, lane);
    if (lane == 63) { s_wsum[w] = incl; s_wnz[w] = incl_nz; }
    __syncthreads();
    uint32_t wave_excl = 0, block_total = 0, block_nz = 0;
#pragma unroll
    for (int k = 0; k < SCAN_WAVES; ++k) {
        const uint32_t t = s_wsum[k], z = s_wnz[k];
        if (k < (int)w) wave_excl = sat_add(wave_excl, t);
        block_total = sat_add(block_total, t);
        block_nz += z;
    }
    uint32_t run = sat_add(wave_excl, incl - tsum); // incl - tsum: the exclusive in-wave prefix (exact unless incl saturated)

    if (w == 0) {
        const unsigned long long mine = ((unsigned long long)block_nz << 32) | (unsigned long long)block_total;
        if (lane == 0) st_agent64(&status[bid], (bid == 0 ? ST_PREFIX : ST_AGG) | mine);
        uint32_t excl = 0, excl_nz = 0;
        if (bid > 0) {
            int look = (int)bid - 1;
            for (;;) {
                const int idx = look - (int)lane;
                unsigned long long sv = ST_PREFIX; // lanes before block 0 contribute a zero prefix
                if (idx >= 0) {
                    uint32_t spins = 0;
                    do {
                        sv = ld_agent64(&status[idx]);
                        if ((sv & ST_MASK) != 0) break;
                        __builtin_amdgcn_s_sleep(1);
                    } while (++spins < GS_SPIN_LIMIT);
                    if ((sv & ST_MASK) == 0) { // gave up: report, then terminate the chain
                        ctl->fault = 1u;
                        sv = ST_PREFIX;
                    }
                }
                const unsigned long long pmask = __ballot((sv & ST_MASK) == ST_PREFIX);
                const uint32_t first = pmask ? (uint32_t)__builtin_ctzll(pmask) : 64u;
                const uint32_t contrib = (lane <= first) ? (uint32_t)sv : 0u;
                const uint32_t contrib_nz = (lane <= first) ? (uint32_t)((sv & ~ST_MASK) >> 32) : 0u;
                excl = sat_add(excl, wave_sum_sat(contrib));
                excl_nz += wave_sum(contrib_nz);
                if (pmask) break;
                look -= 64;
            }
        }
        if (lane == 0) {
            if (bid > 0)
                st_agent64(&status[bid], ST_PREFIX | ((unsigned long long)(excl_nz + block_nz) << 32) | (unsigned long long)sat_add(excl, block_total));
            s_prefix[0] = excl;
            if (bid == nblocks - 1) {
                ctl->num_visible = excl_nz + block_nz;
                ctl->num_intersections = sat_add(excl, block_total);
            }
        }
    }
    __syncthreads();
    run = sat_add(run, s_prefix[0]);
    if (!offsets) return;
    if (base + SCAN_ITEMS <= n) {
        uint4* p = reinterpret_cast<uint4*>(offsets + base);
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS / 4; ++j) {
            uint4 q;
            q.x = run; run += v[4 * j] & GS_COUNT_MASK;
            q.y = run; run += v[4 * j + 1] & GS_COUNT_MASK;
            q.z = run; run += v[4 * j + 2] & GS_COUNT_MASK;
            q.w = run; run += v[4 * j + 3] & GS_COUNT_MASK;
            p[j] = q;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; ++j) {
            if (base + j < n) offsets[base + j] = run;
            run += v[j] & GS_COUNT_MASK;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Emission.  One wave owns 64 consecutive gaussians; their instances are one contiguous run of the
// output (offsets[first] .. offsets[last]+count[last]), written 64 entries per step, lane k of a
// step finding its gaussian by binary search in the wave's 64 local prefixes (LDS).  Key layout
// is the reference's: (y*ntx + x)*1000 + u32(min(50*depth, 999))  (write_tile_ids.wgsl:29-31),
// instance order y outer / x inner, gaussians in index order -- so a stable sort reproduces the
// reference's tie order bit for bit.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void slab_cols_emit(uint32_t rx0, uint32_t rx1, const GsFrame& f, uint32_t& xa, uint32_t& wmain,
                                               uint32_t& alias) {
    const uint32_t hi = rx1 < f.ntx ? rx1 : f.ntx;
    xa = rx0 > f.col0 ? rx0 : f.col0;
    const uint32_t xb = hi < f.col1 ? hi : f.col1;
    wmain = xb > xa ? xb - xa : 0u;
    alias = (rx1 == f.ntx + 1u && f.col0 == 0u) ? 1u : 0u;
}

// perm/n_dev: optional order of emission (element k of the launch is gaussian perm[k], k < *n_dev): the
// depth-ordered pipeline emits gaussians sorted by depth bucket so that the instance sort only has to order by tile.
__global__ __launch_bounds__(256) void gs_emit_kernel(const uint4* __restrict__ gdata, const uint32_t* __restrict__ counts,
                                                       const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ perm,
                                                       const uint32_t* __restrict__ n_dev, GsFrame f,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ values,
                                                       GsControl* ctl) {
    __shared__ uint32_t s_pref[4][64];
    __shared__ uint32_t s_row[4][64]; // xa | wmain<<16 | alias<<31
    __shared__ uint32_t s_yb[4][64];  // y0 | bucket<<16
    __shared__ uint32_t s_gid[4][64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t k = blockIdx.x * 256 + tid;
    const uint32_t n = n_dev ? *n_dev : f.n;
    uint32_t cnt = 0, off = 0, i = 0, packed = 0;
    if (k < n) {
        i = perm ? perm[k] : k;
        packed = counts[i];
        cnt = packed & GS_COUNT_MASK;
        off = offsets[k];
    }
    const uint32_t first_off = __shfl(off, 0, 64);
    // the last lanes of the wave may be past n: take the run length from the inclusive scan
    const uint32_t incl = wave_incl_scan(cnt, lane);
    const uint32_t wave_total = __shfl(incl, 63, 64);
    if (wave_total == 0) return;
    uint32_t row = 0, yb = 0;
    if (cnt) {
        const uint4 rect = gdata[(uint64_t)i * 4 + 3];
        const uint32_t bucket = packed >> GS_COUNT_BITS; // u32(min(50*depth, 999)), computed by the preprocess
        uint32_t xa, wmain, alias;
        slab_cols_emit(rect.x, rect.z, f, xa, wmain, alias);
        row = xa | (wmain << 16) | (alias << 31);
        yb = rect.y | (bucket << 16);
    }
    s_pref[w][lane] = incl - cnt; // exclusive, relative to the wave's first entry
    s_row[w][lane] = row;
    s_yb[w][lane] = yb;
    s_gid[w][lane] = i;
    // single-wave producer/consumer of these LDS rows: no workgroup barrier needed, LDS ops of one
    // wave complete in order; the fence only stops the compiler from reordering.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t e = lane; e < wave_total; e += 64) {
        // largest g with pref[g] <= e
        uint32_t lo = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const uint32_t mid = lo + step;
            if (mid < 64 && s_pref[w][mid] <= e) lo = mid;
        }
        const uint32_t local = e - s_pref[w][lo];
        const uint32_t r = s_row[w][lo], y_b = s_yb[w][lo];
        const uint32_t xa = r & 0xFFFFu, wmain = (r >> 16) & 0x7FFFu, alias = r >> 31;
        const uint32_t wtot = wmain + alias;
        const uint32_t yy = local / wtot, xx = local - yy * wtot;
        const uint32_t x = (xx < wmain) ? xa + xx : f.ntx;
        const uint32_t y = (y_b & 0xFFFFu) + yy;
        const uint32_t key = (y * f.ntx + x) * 1000u + (y_b >> 16);
        const uint32_t dst = first_off + e;
        if (dst < f.capacity) {
            keys[dst] = key;
            values[dst] = s_gid[w][lo];
        } else {
            ctl->overflow = 1u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Work-balanced emission for the depth-ordered pipeline.  There the gaussians arrive sorted by depth
// bucket, so the nearest (largest on screen) ones are neighbours: giving a wave 64 consecutive gaussians
// (as gs_emit_kernel does) hands a few waves millions of instances.  Here a wave owns EMIT_CHUNK
// consecutive OUTPUT slots instead; chunk_table (written by the scan) names the gaussian that covers the
// chunk's first slot, and the wave walks forward from it 64 gaussians at a time.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gs_emit_balanced_kernel(const uint4* __restrict__ gdata, const uint4* __restrict__ grec,
                                                                const uint32_t* __restrict__ chunk_table, GsFrame f,
                                                                uint32_t* __restrict__ keys, uint32_t* __restrict__ values,
                                                                GsControl* ctl, uint32_t hist_bits, uint32_t hist_passes, uint32_t keys16) {
    __shared__ uint32_t s_off[4][64];
    __shared__ uint32_t s_hist[4][256]; // digit counts of the instance sort (digits of key/1000), flushed once per workgroup
    for (uint32_t k = threadIdx.x; k < 4 * 256; k += 256) (&s_hist[0][0])[k] = 0u;
    __syncthreads();
    const uint32_t hmask = (1u << hist_bits) - 1u;
    __shared__ uint32_t s_row[4][64]; // xa | wmain<<16 | alias<<31
    __shared__ uint32_t s_yb[4][64];  // y0 | bucket<<16
    __shared__ uint32_t s_gid[4][64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t nvis = ctl->num_visible;
    uint32_t total = ctl->num_intersections;
    if (total > f.capacity) { // the frame does not fit: flag it, emit what fits (gs_wait grows and re-renders)
        if (tid == 0 && blockIdx.x == 0) ctl->overflow = 1u;
        total = f.capacity;
    }
    const uint32_t nchunks = (total + EMIT_CHUNK - 1u) >> EMIT_CHUNK_SHIFT;
    for (uint32_t c = blockIdx.x * 4u + w; c < nchunks; c += gridDim.x * 4u) {
        const uint32_t c1 = (c + 1u) * EMIT_CHUNK < total ? (c + 1u) * EMIT_CHUNK : total;
        uint32_t e = c * EMIT_CHUNK;
        uint32_t kbase = chunk_table[c];
        while (e < c1 && kbase < nvis) {
            const uint32_t k = kbase + lane;
            uint32_t off = 0xFFFFFFFFu, row = 0, yb = 0, gid = 0, cnt = 0;
            if (k < nvis) {
                const uint4 gr = grec[k]; // {gaussian id, count word, first output slot, -} in (bucket, index) order (k_gsort.hip)
                gid = gr.x;
                const uint32_t packed = gr.y;
                cnt = packed & GS_COUNT_MASK;
                off = gr.z;
                const uint4 rect = gdata[(uint64_t)gid * 4 + 3];
                uint32_t xa, wmain, alias;
                slab_cols_emit(rect.x, rect.z, f, xa, wmain, alias);
                row = xa | (wmain << 16) | (alias << 31);
                yb = rect.y | ((packed >> GS_COUNT_BITS) << 16);
            }
            s_off[w][lane] = off;
            s_row[w][lane] = row;
            s_yb[w][lane] = yb;
            s_gid[w][lane] = gid;
            // slots covered by this group of 64 gaussians end where its last member's instances end
            const uint32_t endk = (k < nvis) ? off + cnt : 0u;
            uint32_t gend = endk;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t o = __shfl_xor(gend, d, 64);
                gend = o > gend ? o : gend;
            }
            const uint32_t stop = gend < c1 ? gend : c1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // two output slots per lane and trip: their owner searches (six dependent LDS reads each) are independent,
            // so the compiler interleaves them and half of that latency is hidden
            for (uint32_t x0 = e + lane; x0 < stop; x0 += 128) {
                const uint32_t x1 = x0 + 64;
                const bool two = x1 < stop;
                uint32_t lo0 = 0, lo1 = 0; // largest j with s_off[j] <= x (absent members carry 0xFFFFFFFF)
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const uint32_t m0 = lo0 + step, m1 = lo1 + step;
                    const uint32_t o0 = s_off[w][m0 & 63u], o1 = s_off[w][m1 & 63u];
                    if (m0 < 64 && o0 <= x0) lo0 = m0;
                    if (m1 < 64 && o1 <= x1) lo1 = m1;
                }
                auto put = [&](uint32_t x, uint32_t lo) {
                    const uint32_t local = x - s_off[w][lo];
                    const uint32_t r = s_row[w][lo], y_b = s_yb[w][lo];
                    const uint32_t xa = r & 0xFFFFu, wmain = (r >> 16) & 0x7FFFu, alias = r >> 31;
                    const uint32_t wtot = wmain + alias;
                    // local / wtot with local < 2^22 and wtot < 2^16: the float quotient is off by at most one
                    uint32_t yy = (uint32_t)((float)local * __builtin_amdgcn_rcpf((float)wtot));
                    uint32_t xx = local - yy * wtot;
                    if ((int32_t)xx < 0) { --yy; xx += wtot; }
                    if (xx >= wtot) { ++yy; xx -= wtot; }
                    const uint32_t px = (xx < wmain) ? xa + xx : f.ntx;
                    const uint32_t py = (y_b & 0xFFFFu) + yy;
                    const uint32_t tile_id = py * f.ntx + px;
                    // the instance sort that follows only orders by tile id (the depth order is already there): with keys16
                    // the sort word is stored alone, as 16 bits; the full key is rebuilt for the tap (gs_rebuild_keys_kernel)
                    if (keys16) reinterpret_cast<uint16_t*>(keys)[x] = (uint16_t)tile_id;
                    else keys[x] = tile_id * 1000u + (y_b >> 16);
                    values[x] = s_gid[w][lo];
                    // the instance sort orders by digits of the tile id: count them here, where the key is in a register,
                    // instead of re-reading all keys in a histogram kernel
                    atomicAdd(&s_hist[0][tile_id & hmask], 1u);
                    if (hist_passes > 1) atomicAdd(&s_hist[1][(tile_id >> hist_bits) & hmask], 1u);
                    if (hist_passes > 2) atomicAdd(&s_hist[2][(tile_id >> (2 * hist_bits)) & hmask], 1u);
                    if (hist_passes > 3) atomicAdd(&s_hist[3][(tile_id >> (3 * hist_bits)) & hmask], 1u);
                };
                put(x0, lo0);
                if (two) put(x1, lo1);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            e = stop;
            kbase += 64;
        }
    }
    __syncthreads();
    for (uint32_t p = 0; p < hist_passes; ++p) {
        const uint32_t c = s_hist[p][threadIdx.x];
        if (c) atomicAdd(&ctl->hist[p][threadIdx.x], c);
    }
}

// ------------------------------------------------------------------------------------------------
// Ranges: ranges[t] = |{ j < I : key_j/1000 <= t }| (entries with tile >= T ignored, A.6).
// Boundary j in [0, I] owns the tiles t with tile[j-1] <= t < tile[j]  (tile[-1] = 0 bound, tile[I] = T):
// every tile is written exactly once, so `ranges` never needs the reference's per-frame clear.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ranges_boundary(uint32_t j, uint32_t lo, uint32_t hi, uint32_t T, uint32_t* __restrict__ ranges) {
    if (lo > T) lo = T;
    if (hi > T) hi = T;
    for (uint32_t t = lo; t < hi; ++t) ranges[t] = j; // almost always empty: 42 M keys, 8 k boundaries
}

// Four consecutive sorted keys per thread (one 16-byte load + the neighbour before them), several
// chunks in flight per thread: a pure streaming read.  Boundary j in [0, I] owns the tiles t with
// tile[j-1] <= t < tile[j] (tile[-1] = 0, tile[I] = T).
// sticky (optional): words that survive the per-frame memset of the control block.  Every frame folds its overflow / fault
// flags and its instance count into them, so gs_wait also learns about frames that were enqueued BEFORE the last one
// ([0] frames that overflowed the capacity, [1] a bounded spin gave up, [2] largest instance count seen).
__device__ __forceinline__ void fold_sticky(const GsControl* ctl, uint32_t capacity, uint32_t* sticky, GsReport* rep) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    gs_frame_report(ctl, ctl->num_intersections, 0u, capacity, sticky, rep);
}

__global__ __launch_bounds__(256) void gs_ranges_kernel(const uint32_t* __restrict__ keys, const GsControl* ctl, uint32_t capacity,
                                                         uint32_t T, uint32_t* __restrict__ ranges, uint32_t* sticky, GsReport* rep) {
    fold_sticky(ctl, capacity, sticky, rep);
    uint32_t I = ctl->num_intersections;
    if (I > capacity) I = capacity;
    const uint64_t nchunks = (uint64_t)I / 4 + 1; // chunk c covers boundaries 4c .. 4c+3 (those <= I)
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += stride) {
        const uint32_t j0 = (uint32_t)(c * 4);
        uint32_t k[4];
        if (j0 + 4 <= I) {
            const uint4 q = *reinterpret_cast<const uint4*>(keys + j0);
            k[0] = q.x; k[1] = q.y; k[2] = q.z; k[3] = q.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) k[i] = (j0 + i < I) ? keys[j0 + i] : 0xFFFFFFFFu;
        }
        uint32_t prev = (j0 == 0) ? 0u : keys[j0 - 1] / 1000u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t j = j0 + i;
            if (j > I) break;
            const uint32_t cur = (j == I) ? T : k[i] / 1000u;
            if (cur != prev || j == I) ranges_boundary(j, prev, cur, T, ranges);
            prev = cur;
        }
    }
}

// The same over sorted 16-bit tile ids (depth-ordered pipeline): eight per 16-byte load.
__global__ __launch_bounds__(256) void gs_ranges16_kernel(const uint16_t* __restrict__ tiles, const GsControl* ctl, uint32_t capacity,
                                                           uint32_t T, uint32_t* __restrict__ ranges, uint32_t* sticky, GsReport* rep) {
    fold_sticky(ctl, capacity, sticky, rep);
    uint32_t I = ctl->num_intersections;
    if (I > capacity) I = capacity;
    const uint64_t nchunks = (uint64_t)I / 8 + 1; // chunk c covers boundaries 8c .. 8c+7 (those <= I)
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += stride) {
        const uint32_t j0 = (uint32_t)(c * 8);
        uint32_t k[8];
        if (j0 + 8 <= I) {
            const uint4 q = *reinterpret_cast<const uint4*>(tiles + j0);
            k[0] = q.x & 0xFFFFu; k[1] = q.x >> 16; k[2] = q.y & 0xFFFFu; k[3] = q.y >> 16;
            k[4] = q.z & 0xFFFFu; k[5] = q.z >> 16; k[6] = q.w & 0xFFFFu; k[7] = q.w >> 16;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) k[i] = (j0 + i < I) ? tiles[j0 + i] : 0xFFFFu;
        }
        uint32_t prev = (j0 == 0) ? 0u : tiles[j0 - 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t j = j0 + i;
            if (j > I) break;
            const uint32_t cur = (j == I) ? T : k[i];
            if (cur != prev || j == I) ranges_boundary(j, prev, cur, T, ranges);
            prev = cur;
        }
    }
}

// GS_BUF_KEYS tap of a frame sorted on 16-bit tile ids: key = tile*1000 + depth bucket of the gaussian (the high bits of
// its tile-count word), write_tile_ids.wgsl:31.
__global__ __launch_bounds__(256) void gs_rebuild_keys_kernel(const uint16_t* __restrict__ tiles, const uint32_t* __restrict__ vals,
                                                               const uint32_t* __restrict__ counts, uint32_t count, uint32_t n, uint32_t id_mask,
                                                               uint32_t* __restrict__ keys) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        const uint32_t g = vals[i] & id_mask;
        keys[i] = (uint32_t)tiles[i] * 1000u + (g < n ? counts[g] >> GS_COUNT_BITS : 0u);
    }
}

// ---- host launchers --------------------------------------------------------------------------------
void gs_launch_ranges16(const uint16_t* tiles, const GsControl* ctl, uint32_t capacity, uint32_t T, uint32_t* ranges, uint32_t grid,
                        uint32_t* sticky, GsReport* rep, hipStream_t st) {
    hipLaunchKernelGGL(gs_ranges16_kernel, dim3(grid), dim3(256), 0, st, tiles, ctl, capacity, T, ranges, sticky, rep);
}
void gs_launch_rebuild_keys(const uint16_t* tiles, const uint32_t* vals, const uint32_t* counts, uint32_t count, uint32_t n, uint32_t id_mask,
                            uint32_t* keys, hipStream_t st) {
    if (!count) return;
    const uint32_t blocks = (count + 255u) / 256u;
    hipLaunchKernelGGL(gs_rebuild_keys_kernel, dim3(blocks < 4096u ? blocks : 4096u), dim3(256), 0, st, tiles, vals, counts, count, n, id_mask, keys);
}
// The per-frame zeroing of the control block and status words, as a kernel of our own (16-byte stores; `bytes` a multiple of 16):
// a hipMemsetAsync captured into the frame graph came back as a frame with a garbage control block after other work on the
// stream between two replays (tools/fuzz_sequence.py); a plain kernel node holds its two arguments by value.
__global__ __launch_bounds__(256) void gs_zero_kernel(uint4* __restrict__ p, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
void gs_launch_zero(void* p, uint64_t bytes, hipStream_t st) {
    const uint64_t n16 = bytes / 16;
    if (!n16) return;
    const uint64_t blocks = (n16 + 255) / 256;
    hipLaunchKernelGGL(gs_zero_kernel, dim3((uint32_t)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, (uint4*)p, n16);
}
uint32_t gs_scan_blocks(uint32_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
void gs_launch_scan(const uint32_t* counts, uint32_t n, uint32_t* offsets, unsigned long long* status, uint32_t* ticket, GsControl* ctl,
                    hipStream_t st) {
    const uint32_t blocks = gs_scan_blocks(n);
    if (!blocks) return;
    hipLaunchKernelGGL(gs_scan_kernel, dim3(blocks), dim3(SCAN_THREADS), 0, st, counts, n, offsets, status, ticket, ctl);
}
uint64_t gs_emit_chunks(uint64_t capacity) { return (capacity >> EMIT_CHUNK_SHIFT) + 2; }
void gs_launch_emit_balanced(const void* gdata, const void* grec, const uint32_t* chunk_table, const GsFrame& f, uint32_t* keys, uint32_t* values,
                             GsControl* ctl, uint32_t grid, uint32_t hist_bits, uint32_t hist_passes, bool keys16, hipStream_t st) {
    hipLaunchKernelGGL(gs_emit_balanced_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)gdata, (const uint4*)grec, chunk_table, f, keys,
                       values, ctl, hist_bits, hist_passes, keys16 ? 1u : 0u);
}
void gs_launch_emit(const void* gdata, const uint32_t* counts, const uint32_t* offsets, const uint32_t* perm, const uint32_t* n_dev,
                    const GsFrame& f, uint32_t* keys, uint32_t* values, GsControl* ctl, hipStream_t st) {
    const uint32_t blocks = (f.n + 255) / 256;
    if (!blocks) return;
    hipLaunchKernelGGL(gs_emit_kernel, dim3(blocks), dim3(256), 0, st, (const uint4*)gdata, counts, offsets, perm, n_dev, f, keys, values, ctl);
}
void gs_launch_ranges(const uint32_t* keys, const GsControl* ctl, uint32_t capacity, uint32_t T, uint32_t* ranges, uint32_t grid,
                      uint32_t* sticky, GsReport* rep, hipStream_t st) {
    hipLaunchKernelGGL(gs_ranges_kernel, dim3(grid), dim3(256), 0, st, keys, ctl, capacity, T, ranges, sticky, rep);
}
