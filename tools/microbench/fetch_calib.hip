// FETCH_SIZE calibration for the blend's record gathers.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads half
// the bytes of a wide coalesced stream and is uncalibrated for other access widths.  Three kernels over a 2 GiB table of 64-byte
// records (far beyond the 256 MiB Infinity Cache), each touching every byte count exactly once, so the HBM bytes are KNOWN:
//   stream16   : lane i reads 16 bytes at 16*i                      -> table bytes
//   gather_blend: lane i reads record perm[i] the way gs_blend_quad_kernel does: 8 B at +0, 12 B at +16, 16 B at +32
//                 (one 64-byte sector per record, every record once)  -> 64 B per record if sectors are fetched singly
//   gather16   : lane i reads 16 B at +0 of record perm[i]           -> 64 B per record (sector) or 128 B (line)
// Run under rocprofv3 --pmc FETCH_SIZE (tools/microbench/fetch_calib.sh) and compare FETCH_SIZE * 1024 with the known bytes.
// hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void stream16(const uint4* t, uint64_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = t[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// perm: a bijection of the records (multiplicative hash with an odd multiplier modulo a power of two)
__device__ __forceinline__ uint64_t perm(uint64_t i, uint64_t mask) { return (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & mask; }
__global__ void gather_blend(const uint32_t* t, uint64_t nrec, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t* rec = t + perm(i, nrec - 1) * 16;
        const u32x2 a = *reinterpret_cast<const u32x2*>(rec);
        const u32x3 b = *reinterpret_cast<const u32x3*>(rec + 4);
        const u32x4 c = *reinterpret_cast<const u32x4*>(rec + 8);
        acc += a.x ^ a.y ^ b.x ^ b.y ^ b.z ^ c.x ^ c.y ^ c.z ^ c.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void gather16(const uint32_t* t, uint64_t nrec, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += (uint64_t)gridDim.x * blockDim.x) {
        const u32x4 c = *reinterpret_cast<const u32x4*>(t + perm(i, nrec - 1) * 16);
        acc += c.x ^ c.y ^ c.z ^ c.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const uint64_t bytes = 2ull << 30, nrec = bytes / 64;
    void* t; uint32_t* out;
    CHECK(hipMalloc(&t, bytes)); CHECK(hipMalloc((void**)&out, 64));
    CHECK(hipMemset(t, 1, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const uint4*)t, bytes / 16, out);
        hipLaunchKernelGGL(gather_blend, dim3(4096), dim3(256), 0, 0, (const uint32_t*)t, nrec, out);
        hipLaunchKernelGGL(gather16, dim3(4096), dim3(256), 0, 0, (const uint32_t*)t, nrec, out);
    }
    CHECK(hipDeviceSynchronize());
    printf("table %llu bytes, %llu records of 64 B; known bytes per launch: stream16 %llu, gather_blend / gather16 %llu (one 64-byte sector per record)\n",
           (unsigned long long)bytes, (unsigned long long)nrec, (unsigned long long)bytes, (unsigned long long)(nrec * 64));
    return 0;
}
