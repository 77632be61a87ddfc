// Host emulation of gaussian-splatting-wgpu_amd/csrc/gs_tight.h: a LOGIC check of the product path's tight binning against
// the oracle's exact per-instance contribution masks (oracle.instance_masks), runnable without a GPU.  The device version
// differs in the last bits (v_log_f32); parity proper is tests/gpu_checks.py on the GPU.  Built and run by run.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#define __device__
#define __forceinline__ inline
#define __builtin_amdgcn_logf(x) log2f(x)
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#include "gs_tight_host.h" // = gs_tight.h without its gs_device.h include (made by run.py)

static std::vector<uint32_t> rd(const char* p) {
    FILE* f = fopen(p, "rb");
    if (!f) exit(3);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint32_t> v(n / 4);
    if (fread(v.data(), 4, v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}
int main(int argc, char** argv) {
    if (argc < 8) return 4;
    auto gd = rd(argv[1]); auto keys = rd(argv[2]); auto vals = rd(argv[3]); auto masks = rd(argv[4]);
    const uint32_t W = atoi(argv[5]), H = atoi(argv[6]), ts = atoi(argv[7]);
    const uint32_t ntx = (uint32_t)ceilf((float)W / ts), nty = (uint32_t)ceilf((float)H / ts);
    const uint32_t sub = ts >= 16 ? ts / 2 : ts, ns = ts / sub;
    const size_t n = gd.size() / 16;
    std::vector<std::map<uint32_t, uint32_t>> sets(n);
    std::vector<char> done(n, 0);
    uint64_t kept = 0, bits = 0, bad = 0, badmask = 0, contributing = 0, exactbits = 0, total_tight = 0;
    for (size_t i = 0; i < keys.size(); ++i) {
        const uint32_t g = vals[i];
        if (!done[g]) {
            done[g] = 1;
            const uint32_t* o = &gd[(size_t)g * 16];
            auto F = [&](int k) { float f; memcpy(&f, &o[k], 4); return f; };
            TightG tg = tight_setup(F(0), F(1), F(4), F(5), F(6), F(11), (float)W, (float)H);
            const uint32_t rx0 = o[12], ry0 = o[13], rx1 = o[14], ry1 = o[15];
            const uint32_t hi = rx1 < ntx ? rx1 : ntx, xa = rx0, wmain = hi > xa ? hi - xa : 0, alias = (rx1 == ntx + 1);
            const float inv_ts = 1.0f / (float)ts, inv_sub = 1.0f / (float)sub;
            uint32_t got = 0;
            if (tg.mode != 0) {
                // through the row items of the tight row pipeline (gs_tight.h: tight_slot_item / tight_item_mask)
                uint32_t ra, rb;
                tight_rows(tg, ry0, ry1, ts, inv_ts, nty, alias, ra, rb);
                const uint32_t nrows = rb - ra, nslots = nrows * (1 + alias);
                for (uint32_t sl = 0; sl < nslots; ++sl) {
                    uint32_t w1, w2;
                    const uint32_t len = tight_slot_item(tg, sl, ra, nrows, ts, inv_ts, sub, inv_sub, ns, nty, xa, wmain, alias, w1, w2);
                    got += len;
                    if (!len) continue;
                    const uint32_t trow = w1 & 0xFF, tlo = (w1 >> 8) & 0xFF, l2 = ((w1 >> 16) & 0xFF) + 1;
                    if (l2 != len) { printf("len mismatch\n"); return 1; }
                    for (uint32_t q = 0; q < len; ++q) {
                        const uint32_t m = tight_item_mask(w2, q, ns);
                        // an aliased item is recorded under the key tile it carries in the reference: (row - 1) * ntx + ntx
                        const uint32_t tile = trow * ntx + tlo + q;
                        sets[g][tile] |= 0x100 | m;
                    }
                }
            }
            total_tight += got;
        }
        const uint32_t tile = keys[i] / 1000, em = masks[i];
        uint32_t esub = 0; // exact 8x8-block mask at the emission's sub-block granularity
        if (ts == 16) esub = em;
        else if (ts == 8) esub = em & 1;
        else for (uint32_t by = 0; by < 4; ++by) for (uint32_t bx = 0; bx < 4; ++bx) if ((em >> (by * 4 + bx)) & 1) esub |= 1u << ((by / 2) * 2 + bx / 2);
        if (em) { contributing++; exactbits += __builtin_popcount(esub); }
        auto it = sets[g].find(tile);
        if (it == sets[g].end()) {
            if (em) { if (bad < 5) printf("DROPPED contributing: g=%u tile=%u mask=%x\n", g, tile, em); bad++; }
        } else {
            kept++;
            bits += __builtin_popcount(it->second & 0xF);
            if (esub & ~it->second) { if (badmask < 5) printf("MASK miss: g=%u tile=%u exact=%x got=%x\n", g, tile, esub, it->second & 0xF); badmask++; }
        }
    }
    printf("reference instances %zu | exactly contributing %.4f | kept by tight binning %.4f (tight total %lu) | mask bits per kept %.3f | "
           "exact sub-blocks per contributing %.3f | dropped-but-contributing %lu | mask misses %lu\n",
           keys.size(), (double)contributing / keys.size(), (double)kept / keys.size(), (unsigned long)total_tight,
           (double)bits / (kept ? kept : 1), (double)exactbits / (contributing ? contributing : 1), (unsigned long)bad, (unsigned long)badmask);
    return (bad || badmask) ? 1 : 0;
}
