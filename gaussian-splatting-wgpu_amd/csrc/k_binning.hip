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
#include "gs_tight.h"

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
// the sort key, write_tile_ids.wgsl:31) in the high 10.
//   gather   : optional permutation; element k of the scan is counts[gather[k]] (depth-ordered pipeline)
//   n_dev    : optional device word holding the element count (then n_static is only the launch bound)
//   offsets  : optional output, exclusive prefix of the tile counts
//   vkey/vval: optional ordered compaction of the non-zero elements: (bucket, element index)
//   chunk_table: optional; chunk_table[c] = the element whose instances contain output slot c*EMIT_CHUNK, which
//              lets the balanced emission start every chunk without searching
__global__ __launch_bounds__(SCAN_THREADS) void gs_scan_kernel(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ gather,
                                                       const uint32_t* __restrict__ n_dev, uint32_t n_static,
                                                       uint32_t* __restrict__ offsets, uint32_t* __restrict__ vkey,
                                                       uint32_t* __restrict__ vval, uint32_t* __restrict__ chunk_table,
                                                       uint32_t chunk_cap, unsigned long long* status, uint32_t* ticket,
                                                       GsControl* ctl, uint32_t write_totals, uint32_t* __restrict__ ccounts,
                                                       uint32_t* __restrict__ coffsets) {
    __shared__ uint32_t s_bid;
    __shared__ uint32_t s_wsum[SCAN_WAVES];
    __shared__ uint32_t s_wnz[SCAN_WAVES];
    __shared__ uint32_t s_prefix[2];
    __shared__ uint32_t s_gh[64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n = n_dev ? *n_dev : n_static;
    const uint32_t nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (tid == 0) s_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t bid = s_bid;
    if (bid >= nblocks) return; // uniform; every block that can be waited on has a lower ticket
    const uint32_t base = bid * SCAN_TILE + tid * SCAN_ITEMS;

    uint32_t v[SCAN_ITEMS];
    if (!gather && base + SCAN_ITEMS <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(counts + base);
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS / 4; ++j) {
            const uint4 q = p[j];
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; ++j) v[j] = (base + j < n) ? counts[gather ? gather[base + j] : base + j] : 0u;
    }
    uint32_t tsum = 0, tnz = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { tsum += v[j] & GS_COUNT_MASK; tnz += ((v[j] & GS_COUNT_MASK) != 0u); }
    const uint32_t incl = wave_incl_scan_sat(tsum, lane); // a thread's 24 counts (< 2^22 each) cannot wrap; wider sums saturate
    const uint32_t incl_nz = wave_incl_scan(tnz, lane);
    if (lane == 63) { s_wsum[w] = incl; s_wnz[w] = incl_nz; }
    __syncthreads();
    uint32_t wave_excl = 0, block_total = 0, wave_excl_nz = 0, block_nz = 0;
#pragma unroll
    for (int k = 0; k < SCAN_WAVES; ++k) {
        const uint32_t t = s_wsum[k], z = s_wnz[k];
        if (k < (int)w) { wave_excl = sat_add(wave_excl, t); wave_excl_nz += z; }
        block_total = sat_add(block_total, t);
        block_nz += z;
    }
    uint32_t run = sat_add(wave_excl, incl - tsum); // incl - tsum: the exclusive in-wave prefix (exact unless incl saturated)
    uint32_t run_nz = wave_excl_nz + incl_nz - tnz;

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
            s_prefix[1] = excl_nz;
            if (write_totals && bid == nblocks - 1) {
                ctl->num_visible = excl_nz + block_nz;
                ctl->num_intersections = sat_add(excl, block_total);
            }
        }
    }
    __syncthreads();
    run = sat_add(run, s_prefix[0]);
    run_nz += s_prefix[1];
    const uint32_t first_nz = run_nz;
    if (vkey) { // ordered compaction of the elements with a non-zero tile count
        // ... and (round 1's gaussian-level radix sort only, a profiling path now: no ccounts) the digit histograms of the sort
        // that follows, two 5-bit digits of the bucket: per-workgroup LDS counters, one global atomic per non-empty bin.  The
        // counting sort of k_gsort.hip builds its own table, and 64 lanes adding into 32 counters serialise: skipped otherwise.
        const bool want_gh = !ccounts;
        if (tid < 64u) s_gh[tid] = 0u;
        __syncthreads();
        uint32_t roff = run;
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; ++j) {
            if ((v[j] & GS_COUNT_MASK) != 0u) {
                const uint32_t bucket = v[j] >> GS_COUNT_BITS;
                if (want_gh) vkey[run_nz] = bucket; // (the sort key of the radix path; the count word below carries the bucket too)
                vval[run_nz] = base + j;
                if (ccounts) { // the compacted list with its own counts and offsets: what an emission in index order walks
                    ccounts[run_nz] = v[j];
                    if (coffsets) coffsets[run_nz] = roff;
                }
                ++run_nz;
                roff += v[j] & GS_COUNT_MASK;
                if (want_gh) {
                    atomicAdd(&s_gh[bucket & 31u], 1u);
                    atomicAdd(&s_gh[32u + ((bucket >> 5) & 31u)], 1u);
                }
            }
        }
        __syncthreads();
        if (want_gh && tid < 64u && s_gh[tid]) atomicAdd(&ctl->ghist[tid >> 5][tid & 31u], s_gh[tid]);
    }
    if (chunk_table) { // (with ccounts: positions in the compacted list, otherwise element indices)
        uint32_t r2 = run, knz = first_nz;
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; ++j) {
            const uint32_t cnt = v[j] & GS_COUNT_MASK;
            if (cnt) {
                const uint32_t last = (r2 + cnt - 1u) >> EMIT_CHUNK_SHIFT;
                for (uint32_t c = (r2 + EMIT_CHUNK - 1u) >> EMIT_CHUNK_SHIFT; c <= last && c < chunk_cap; ++c) chunk_table[c] = ccounts ? knz : base + j;
                ++knz;
            }
            r2 += cnt;
        }
    }
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
__global__ __launch_bounds__(256) void gs_emit_balanced_kernel(const uint4* __restrict__ gdata, const uint32_t* __restrict__ counts,
                                                                const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ perm,
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
                gid = perm[k];
                const uint32_t packed = counts[k]; // sorted order, like offsets and perm
                cnt = packed & GS_COUNT_MASK;
                off = offsets[k];
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
// Tight emission (product path, gs_tight.h).  Same work split as gs_emit_balanced_kernel -- a wave owns EMIT_CHUNK
// consecutive OUTPUT slots and walks the gaussians that cover them 64 at a time -- but a gaussian's instances are no
// longer "every tile of its rect": they are, per tile row of the rect, the run of tiles that intersect its
// alpha >= 1/255 ellipse (tight_row; the projection counted exactly these with the same function).  Two levels:
//   row-items : lane = one (gaussian, tile row) of the group; computes the row's run and, by a wave scan segmented by
//               gaussian, the output slot of its first instance;
//   instances : lane = one output slot of the batch; finds its row-item by binary search in the batch's prefix (LDS),
//               builds the sub-block mask from the two half-strip intervals of the row and stores (tile id, id | mask << 28).
// The order of a gaussian's instances is free (a stable sort by tile follows and a gaussian meets a tile at most once,
// the aliased duplicate excepted, which is identical); only the order of the gaussians matters, and that is `perm`'s.
// by_index: the list is in gaussian-index order (reference order): the keys are the full tile*1000 + bucket sort words.
// ------------------------------------------------------------------------------------------------
struct EmitTightWave {
    uint32_t off[64], rp[64], run[64], gid[64], y0b[64], cols[64];
    float4 pA[64], pB[64], pC[64];
    uint32_t incl[64], slot[64], rowbase[64], tlo[64], first[64], s0[64], s1[64], rgid[64], amask[64], mark[64]; // (5 workgroups of 4 waves + the histogram = 160 KB)
};

__global__ __launch_bounds__(256, 5) void gs_emit_tight_kernel(const uint4* __restrict__ gdata, const uint32_t* __restrict__ counts,
                                                             const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ perm,
                                                             const uint32_t* __restrict__ chunk_table, GsFrame f,
                                                             uint32_t* __restrict__ keys, uint32_t* __restrict__ values, GsControl* ctl,
                                                             uint32_t hist_bits, uint32_t hist_passes, uint32_t keys16, uint32_t by_index) {
    __shared__ EmitTightWave s_w[4];
    __shared__ uint32_t s_hist[4][256];
    for (uint32_t k = threadIdx.x; k < 4 * 256; k += 256) (&s_hist[0][0])[k] = 0u;
    __syncthreads();
    const uint32_t hmask = (1u << hist_bits) - 1u;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    EmitTightWave& S = s_w[w];
    const uint32_t nel = ctl->num_visible; // elements of the emission order: the visible gaussians (index or depth order)
    uint32_t total = ctl->num_intersections;
    if (total > f.capacity) { // the frame does not fit: flag it, emit what fits (gs_wait grows and re-renders)
        if (tid == 0 && blockIdx.x == 0) ctl->overflow = 1u;
        total = f.capacity;
    }
    const uint32_t ts = f.tile_size, sub = ts >= 16u ? ts / 2u : ts, ns = ts / sub; // half strips per tile row: 1 (tile 8) or 2
    const float inv_ts = 1.0f / (float)ts, inv_sub = 1.0f / (float)sub;
    const float Wf = (float)f.width, Hf = (float)f.height;
    const uint32_t nchunks = (total + EMIT_CHUNK - 1u) >> EMIT_CHUNK_SHIFT;
    // (static split: 18 000 chunks drawn from ONE ticket word cost 200 us -- a hot word serves ~88 atomics per microsecond)
    for (uint32_t c = blockIdx.x * 4u + w; c < nchunks; c += gridDim.x * 4u) {
        const uint32_t S0 = c * EMIT_CHUNK;
        const uint32_t S1 = (c + 1u) * EMIT_CHUNK < total ? (c + 1u) * EMIT_CHUNK : total;
        uint32_t e = S0;
        uint32_t kbase = chunk_table[c];
        while (e < S1 && kbase < nel) {
            // ---- the group: 64 consecutive elements of the emission order, one per lane ----
            const uint32_t k = kbase + lane;
            uint32_t off = 0xFFFFFFFFu, cnt = 0, nrows = 0, gid = 0, y0b = 0, colsw = 0;
            TightG tg;
            tg.gx = tg.gy = tg.cx = tg.cy = tg.cz = tg.cxz = tg.lim2 = tg.rcx = tg.xmax = tg.ymax = tg.dyR = tg.eR = 0.0f;
            tg.mode = 0u;
            if (k < nel) {
                const uint32_t packed = counts[k];
                cnt = packed & GS_COUNT_MASK;
                off = offsets[k];
                if (cnt && off < S1) {
                    gid = perm ? perm[k] : k;
                    const uint4 r0 = gdata[(uint64_t)gid * 4 + 0], r1 = gdata[(uint64_t)gid * 4 + 1], r3 = gdata[(uint64_t)gid * 4 + 3];
                    const float op = __uint_as_float(gdata[(uint64_t)gid * 4 + 2].w);
                    tg = tight_setup(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r1.x), __uint_as_float(r1.y),
                                     __uint_as_float(r1.z), op, Wf, Hf);
                    uint32_t xa, wmain, alias;
                    slab_cols_emit(r3.x, r3.z, f, xa, wmain, alias);
                    uint32_t ra, rb;
                    tight_rows(tg, r3.y, r3.w, ts, inv_ts, f.nty, alias, ra, rb); // the rows the projection counted
                    nrows = rb - ra;
                    y0b = ra | ((packed >> GS_COUNT_BITS) << 16);
                    colsw = xa | (wmain << 16) | (alias << 31);
                }
            }
            // slots covered by this group end where its last present member's instances end
            const uint32_t gend = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_max((k < nel) ? off + cnt : 0u), 63);
            const uint32_t stop = gend < S1 ? gend : S1;
            const uint32_t rincl = wave_incl_scan(nrows, lane);
            const uint32_t R = (uint32_t)__builtin_amdgcn_readlane((int)rincl, 63);
            const uint32_t myrp = rincl - nrows;
            uint32_t jcarry = 0u; // owner (+1) of the row-item just before the batch
            S.off[lane] = off;
            S.rp[lane] = rincl - nrows;
            S.run[lane] = 0u;
            S.gid[lane] = gid;
            S.y0b[lane] = y0b;
            S.cols[lane] = colsw;
            S.pA[lane] = make_float4(tg.gx, tg.gy, tg.cx, tg.cy);
            S.pB[lane] = make_float4(tg.cz, tg.cxz, tg.lim2, tg.rcx);
            S.pC[lane] = make_float4(tg.xmax, tg.dyR, tg.eR, __uint_as_float(tg.mode));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t rb = 0; rb < R; rb += 64) {
                // ---- row-items: lane = (gaussian j of the group, tile row) ----
                const uint32_t ri = rb + lane;
                uint32_t len = 0, mainlen = 0, slot0 = 0, j = 0, rowbase = 0, tlo = 0, w0 = 0, w1 = 0, am = 0;
                // owner of a row-item = the member whose rows [rp, rp + nrows) hold it: every member with rows marks the batch
                // position of its first one, a running maximum spreads the marks (DPP, no chain of dependent LDS reads)
                S.mark[lane] = 0u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (nrows && myrp >= rb && myrp < rb + 64u) S.mark[myrp - rb] = lane + 1u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                {
                    uint32_t m = wave_incl_max(S.mark[lane]);
                    m = m > jcarry ? m : jcarry;
                    jcarry = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
                    j = m ? m - 1u : 0u;
                }
                if (ri < R) {
                    const float4 a = S.pA[j], b = S.pB[j], cc = S.pC[j];
                    TightG g;
                    g.gx = a.x; g.gy = a.y; g.cx = a.z; g.cy = a.w; g.cz = b.x; g.cxz = b.y; g.lim2 = b.z; g.rcx = b.w;
                    g.xmax = cc.x; g.dyR = cc.y; g.eR = cc.z; g.mode = __float_as_uint(cc.w); g.ymax = 0.0f; // (ymax only picks the rows)
                    const uint32_t yb = S.y0b[j], cw = S.cols[j];
                    const uint32_t ty = (yb & 0xFFFFu) + (ri - S.rp[j]);
                    const uint32_t xa = cw & 0xFFFFu, wmain = (cw >> 16) & 0x7FFFu, alias = cw >> 31;
                    TightRow r;
                    const TightChord cb = tight_chord_at(g, tight_row_dy(g, ty, ts)), ca = tight_chord_at(g, tight_row_dy(g, ty + 1u, ts));
                    len = tight_row(g, ty, ts, inv_ts, f.nty, xa, wmain, alias, cb, ca, r);
                    mainlen = len - r.alias;
                    rowbase = ty * f.ntx;
                    tlo = (uint32_t)r.tlo;
                    if (mainlen) {
                        int lo[2], hi[2];
                        const int cmin = (int)(tlo * ns), cmax = (int)((tlo + mainlen) * ns) - 1;
                        tight_substrips(g, ty, ts, sub, inv_sub, cmin, cmax, cb, ca, lo, hi);
                        w0 = (uint32_t)lo[0] | ((uint32_t)(hi[0] + 1) << 16);
                        w1 = (uint32_t)lo[1] | ((uint32_t)(hi[1] + 1) << 16);
                    }
                    if (r.alias) { // its sub-blocks: tile (ty + 1, 0)
                        int lo[2], hi[2];
                        const TightChord c2 = tight_chord_at(g, tight_row_dy(g, ty + 2u, ts));
                        tight_substrips(g, ty + 1u, ts, sub, inv_sub, 0, (int)ns - 1, ca, c2, lo, hi);
                        am = (uint32_t)(lo[0] <= 0 && 0 <= hi[0]);
                        if (ns == 2u) am |= ((uint32_t)(lo[0] <= 1 && 1 <= hi[0]) << 1) | ((uint32_t)(lo[1] <= 0 && 0 <= hi[1]) << 2) | ((uint32_t)(lo[1] <= 1 && 1 <= hi[1]) << 3);
                    }
                }
                // output slot of the row's first instance: the gaussian's offset + its rows before this batch + its rows before
                // this one inside the batch (a wave scan, segmented by gaussian through the position of its first row-item)
                const uint32_t lincl = wave_incl_scan(len, lane);
                const uint32_t lexcl = lincl - len;
                const uint32_t rp_j = (ri < R) ? S.rp[j] : 0u;
                const uint32_t fl = rp_j > rb ? rp_j - rb : 0u; // lane of this gaussian's first row-item in the batch
                const uint32_t excl_first = __shfl(lexcl, fl, 64);
                if (ri < R) {
                    const uint32_t within = lexcl - excl_first;
                    const uint32_t carry = S.run[j];
                    slot0 = S.off[j] + carry + within;
                    // the gaussian's last row-item of the batch records what the batch added (one wave, in-order LDS: every
                    // lane has read run[j] before this store is issued)
                    const uint32_t rows_j = (j < 63u ? S.rp[j + 1] : R) - rp_j;
                    const bool last_of_j = (ri + 1u == rp_j + rows_j) || lane == 63u;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    if (last_of_j) S.run[j] = carry + within + len;
                }
                // clip to this wave's slots [S0, S1)
                uint32_t j0 = 0, j1 = len;
                if (slot0 < S0) j0 = S0 - slot0 < len ? S0 - slot0 : len;
                if (slot0 + len > S1) j1 = S1 > slot0 ? S1 - slot0 : 0u;
                const uint32_t lenx = j1 > j0 ? j1 - j0 : 0u;
                const uint32_t xincl = wave_incl_scan(lenx, lane);
                const uint32_t xtotal = (uint32_t)__builtin_amdgcn_readlane((int)xincl, 63);
                const uint32_t xex = xincl - lenx;
                S.incl[lane] = xex;
                S.slot[lane] = slot0 + j0;
                S.rowbase[lane] = rowbase;
                S.tlo[lane] = tlo;
                S.first[lane] = j0 | (mainlen << 16);
                S.s0[lane] = w0;
                S.s1[lane] = w1;
                S.amask[lane] = am | ((ri < R) ? ((S.y0b[j] >> 16) << 4) : 0u); // alias sub-blocks, depth bucket
                S.rgid[lane] = (ri < R) ? S.gid[j] : 0u; // (through LDS, not a shuffle: the instance loop's last trip is divergent)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // ---- instances: lane = one output slot of the batch ----
                uint32_t icarry = 0u;
                for (uint32_t t0 = 0; t0 < xtotal; t0 += 64) {
                    // owner of an output slot: the row-item whose clipped run [xex, xex + lenx) holds it (marks as above)
                    S.mark[lane] = 0u;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    if (lenx && xex >= t0 && xex < t0 + 64u) S.mark[xex - t0] = lane + 1u;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    uint32_t mm = wave_incl_max(S.mark[lane]);
                    mm = mm > icarry ? mm : icarry;
                    icarry = (uint32_t)__builtin_amdgcn_readlane((int)mm, 63);
                    const uint32_t t = t0 + lane;
                    if (t >= xtotal) continue;
                    const uint32_t i = mm - 1u;
                    const uint32_t ex = S.incl[i];
                    const uint32_t fw = S.first[i];
                    const uint32_t q = (fw & 0xFFFFu) + (t - ex); // instance of the row
                    const uint32_t ml = fw >> 16;
                    const uint32_t dst = S.slot[i] + (t - ex);
                    uint32_t tile_id, mask;
                    if (q < ml) {
                        const uint32_t tc = S.tlo[i] + q;
                        tile_id = S.rowbase[i] + tc;
                        const uint32_t a0 = S.s0[i], a1 = S.s1[i];
                        if (ns == 2u) {
                            const uint32_t c0 = 2u * tc, c1 = c0 + 1u;
                            const uint32_t l0 = a0 & 0xFFFFu, h0 = a0 >> 16, l1 = a1 & 0xFFFFu, h1 = a1 >> 16;
                            mask = (uint32_t)(l0 <= c0 && c0 < h0) | ((uint32_t)(l0 <= c1 && c1 < h0) << 1) |
                                   ((uint32_t)(l1 <= c0 && c0 < h1) << 2) | ((uint32_t)(l1 <= c1 && c1 < h1) << 3);
                        } else {
                            mask = (uint32_t)((a0 & 0xFFFFu) <= tc && tc < (a0 >> 16));
                        }
                    } else { // the aliased instance: column ntx of this row = tile (row + 1, 0) (write_tile_ids.wgsl:29)
                        tile_id = S.rowbase[i] + f.ntx;
                        mask = S.amask[i] & 15u;
                    }
                    const uint32_t og = S.rgid[i], ob = S.amask[i] >> 4;
                    if (keys16) reinterpret_cast<uint16_t*>(keys)[dst] = (uint16_t)tile_id;
                    else keys[dst] = tile_id * 1000u + ob;
                    values[dst] = og | (mask << GS_ID_BITS);
                    if (!by_index) { // digits of the tile id: the sort word of the depth-ordered pipeline (index order: the sort's own histogram pass)
                        atomicAdd(&s_hist[0][tile_id & hmask], 1u);
                        if (hist_passes > 1) atomicAdd(&s_hist[1][(tile_id >> hist_bits) & hmask], 1u);
                        if (hist_passes > 2) atomicAdd(&s_hist[2][(tile_id >> (2 * hist_bits)) & hmask], 1u);
                        if (hist_passes > 3) atomicAdd(&s_hist[3][(tile_id >> (3 * hist_bits)) & hmask], 1u);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            e = stop;
            kbase += 64;
        }
    }
    __syncthreads();
    if (!by_index)
        for (uint32_t p = 0; p < hist_passes; ++p) {
            const uint32_t cnt = s_hist[p][threadIdx.x];
            if (cnt) atomicAdd(&ctl->hist[p][threadIdx.x], cnt);
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
__device__ __forceinline__ void fold_sticky(const GsControl* ctl, uint32_t capacity, uint32_t* sticky) {
    if (!sticky || blockIdx.x != 0 || threadIdx.x != 0) return;
    const uint32_t I = ctl->num_intersections;
    if (ctl->overflow || I > capacity) atomicAdd(&sticky[0], 1u);
    if (ctl->fault) atomicOr(&sticky[1], 1u);
    atomicMax(&sticky[2], I);
}

__global__ __launch_bounds__(256) void gs_ranges_kernel(const uint32_t* __restrict__ keys, const GsControl* ctl, uint32_t capacity,
                                                         uint32_t T, uint32_t* __restrict__ ranges, uint32_t* sticky) {
    fold_sticky(ctl, capacity, sticky);
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
                                                           uint32_t T, uint32_t* __restrict__ ranges, uint32_t* sticky) {
    fold_sticky(ctl, capacity, sticky);
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
                        uint32_t* sticky, hipStream_t st) {
    hipLaunchKernelGGL(gs_ranges16_kernel, dim3(grid), dim3(256), 0, st, tiles, ctl, capacity, T, ranges, sticky);
}
void gs_launch_rebuild_keys(const uint16_t* tiles, const uint32_t* vals, const uint32_t* counts, uint32_t count, uint32_t n, uint32_t id_mask,
                            uint32_t* keys, hipStream_t st) {
    if (!count) return;
    const uint32_t blocks = (count + 255u) / 256u;
    hipLaunchKernelGGL(gs_rebuild_keys_kernel, dim3(blocks < 4096u ? blocks : 4096u), dim3(256), 0, st, tiles, vals, counts, count, n, id_mask, keys);
}
uint32_t gs_scan_blocks(uint32_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
void gs_launch_scan(const uint32_t* counts, const uint32_t* gather, const uint32_t* n_dev, uint32_t n_static, uint32_t* offsets,
                    uint32_t* vkey, uint32_t* vval, uint32_t* chunk_table, uint32_t chunk_cap, unsigned long long* status, uint32_t* ticket,
                    GsControl* ctl, uint32_t write_totals, hipStream_t st, uint32_t* ccounts, uint32_t* coffsets) {
    const uint32_t blocks = gs_scan_blocks(n_static);
    if (!blocks) return;
    hipLaunchKernelGGL(gs_scan_kernel, dim3(blocks), dim3(SCAN_THREADS), 0, st, counts, gather, n_dev, n_static, offsets, vkey, vval, chunk_table,
                       chunk_cap, status, ticket, ctl, write_totals, ccounts, coffsets);
}
uint64_t gs_emit_chunks(uint64_t capacity) { return (capacity >> EMIT_CHUNK_SHIFT) + 2; }
void gs_launch_emit_balanced(const void* gdata, const uint32_t* counts, const uint32_t* offsets, const uint32_t* perm,
                             const uint32_t* chunk_table, const GsFrame& f, uint32_t* keys, uint32_t* values, GsControl* ctl, uint32_t grid,
                             uint32_t hist_bits, uint32_t hist_passes, bool keys16, hipStream_t st) {
    hipLaunchKernelGGL(gs_emit_balanced_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)gdata, counts, offsets, perm, chunk_table, f, keys,
                       values, ctl, hist_bits, hist_passes, keys16 ? 1u : 0u);
}
void gs_launch_emit_tight(const void* gdata, const uint32_t* counts, const uint32_t* offsets, const uint32_t* perm,
                          const uint32_t* chunk_table, const GsFrame& f, uint32_t* keys, uint32_t* values, GsControl* ctl, uint32_t grid,
                          uint32_t hist_bits, uint32_t hist_passes, bool keys16, bool by_index, hipStream_t st) {
    hipLaunchKernelGGL(gs_emit_tight_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)gdata, counts, offsets, perm, chunk_table, f, keys,
                       values, ctl, hist_bits, hist_passes, keys16 ? 1u : 0u, by_index ? 1u : 0u);
}
void gs_launch_emit(const void* gdata, const uint32_t* counts, const uint32_t* offsets, const uint32_t* perm, const uint32_t* n_dev,
                    const GsFrame& f, uint32_t* keys, uint32_t* values, GsControl* ctl, hipStream_t st) {
    const uint32_t blocks = (f.n + 255) / 256;
    if (!blocks) return;
    hipLaunchKernelGGL(gs_emit_kernel, dim3(blocks), dim3(256), 0, st, (const uint4*)gdata, counts, offsets, perm, n_dev, f, keys, values, ctl);
}
void gs_launch_ranges(const uint32_t* keys, const GsControl* ctl, uint32_t capacity, uint32_t T, uint32_t* ranges, uint32_t grid,
                      uint32_t* sticky, hipStream_t st) {
    hipLaunchKernelGGL(gs_ranges_kernel, dim3(grid), dim3(256), 0, st, keys, ctl, capacity, T, ranges, sticky);
}
