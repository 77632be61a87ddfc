// Calibration only (never linked into the product): how fast does rocPRIM's tuned radix_sort_pairs sort
// the same workload (u32 keys < 2^23, u32 payloads) on this GPU?  A known-good reference for the
// sort stage's achievable rate (cdna_hip_programming.md §5.4 rule 10).
// build: hipcc --offload-arch=gfx950 -O3 -o rocprim_sort rocprim_sort.cpp
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
    size_t n = argc > 1 ? atoll(argv[1]) : 42500000;
    int bits = argc > 2 ? atoi(argv[2]) : 23;
    std::vector<uint32_t> hk(n), hv(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hk[i] = (uint32_t)(s >> 20) & ((1u << bits) - 1); hv[i] = (uint32_t)i; }
    uint32_t *ki, *ko, *vi, *vo;
    hipMalloc(&ki, n * 4); hipMalloc(&ko, n * 4); hipMalloc(&vi, n * 4); hipMalloc(&vo, n * 4);
    hipMemcpy(ki, hk.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(vi, hv.data(), n * 4, hipMemcpyHostToDevice);
    size_t tmp_bytes = 0; void* tmp = nullptr;
    rocprim::radix_sort_pairs(nullptr, tmp_bytes, ki, ko, vi, vo, n, 0, bits);
    hipMalloc(&tmp, tmp_bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it) rocprim::radix_sort_pairs(tmp, tmp_bytes, ki, ko, vi, vo, n, 0, bits);
    hipEventRecord(a);
    const int R = 10;
    for (int it = 0; it < R; ++it) rocprim::radix_sort_pairs(tmp, tmp_bytes, ki, ko, vi, vo, n, 0, bits);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("rocprim radix_sort_pairs n=%zu bits=%d: %.1f us per sort, %.2f Gpairs/s, tmp=%zu MB\n", n, bits, ms * 1000 / R, n / (ms / R) / 1e6, tmp_bytes >> 20);
    // device copy ceiling
    hipEventRecord(a);
    for (int it = 0; it < R; ++it) hipMemcpyAsync(ko, ki, n * 4, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("d2d copy %zu MB: %.1f us -> %.2f TB/s (read+write)\n", (n * 4) >> 20, ms * 1000 / R, 2.0 * n * 4 / (ms / R * 1e-3) / 1e12);
    return 0;
}
