// How should the projection read the 192-byte SH records of its survivors?  N records, a sorted random subset of them (density p, the
// cull's survivors: ascending indices, ~2.5 records apart at config B) is read once, three ways:
//   per_lane    : lane i reads the 12 float4 of ITS survivor's record (what gs_preprocess_kernel does: every load instruction touches
//                 64 different records)
//   cooperative : the wave reads its 64 survivors' records in 12 instructions of 64 consecutive float4 slots (slot = 12 * survivor +
//                 part): 5.3 whole records per instruction, contiguous 192-byte runs; no transposition back to the owner lanes
//   coop_lds    : cooperative + the transposition through LDS to the owner lane (208-byte pitch), which then reads its 12 float4
//   split 32+192 / merged 256: geometry + SH of a survivor from two arrays (this round's layout) or from ONE 256-byte record, read in one go
// Prints microseconds and TB/s of record bytes for each.  hipcc --offload-arch=gfx950 -O3 sh_gather.hip -o sh_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void per_lane(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, uint32_t n, float* __restrict__ out) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const float4* r = rec + (uint64_t)idx[t] * 12u;
    float4 v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = r[k];
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 12; ++k) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    out[t] = acc;
}

__global__ __launch_bounds__(256) void cooperative(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, uint32_t n, float* __restrict__ out) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t mine = t < n ? idx[t] : idx[n - 1];
    float4 v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const uint32_t slot = (uint32_t)k * 64u + lane; // of the wave's 768 float4
        const uint32_t who = slot / 12u, part = slot % 12u;
        const uint32_t g = (uint32_t)__shfl((int)mine, (int)who, 64);
        v[k] = rec[(uint64_t)g * 12u + part];
    }
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 12; ++k) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    if (t < n) out[t] = acc;
}

__global__ __launch_bounds__(256) void coop_lds(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, uint32_t n, float* __restrict__ out) {
    __shared__ float4 stage[4][64 * 13]; // 208-byte pitch per record: the owner's b128 reads do not collide
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t mine = t < n ? idx[t] : idx[n - 1];
    float4 v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const uint32_t slot = (uint32_t)k * 64u + lane;
        const uint32_t who = slot / 12u, part = slot % 12u;
        const uint32_t g = (uint32_t)__shfl((int)mine, (int)who, 64);
        v[k] = rec[(uint64_t)g * 12u + part];
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const uint32_t slot = (uint32_t)k * 64u + lane;
        stage[w][(slot / 12u) * 13u + slot % 12u] = v[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 12; ++k) { const float4 x = stage[w][lane * 13u + k]; acc += (x.x + x.y) + (x.z + x.w); }
    if (t < n) out[t] = acc;
}

// geometry + SH of a survivor: two arrays (32-byte and 192-byte records, the round-3 layout) or ONE 256-byte record (14 float4 used)
__global__ __launch_bounds__(256) void split_arrays(const float4* __restrict__ geo, const float4* __restrict__ rec, const uint32_t* __restrict__ idx, uint32_t n, float* __restrict__ out) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const uint32_t g = idx[t];
    const float4 a = geo[(uint64_t)g * 2u], b = geo[(uint64_t)g * 2u + 1u];
    const float4* r = rec + (uint64_t)g * 12u;
    float4 v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = r[k];
    float acc = (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
#pragma unroll
    for (int k = 0; k < 12; ++k) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    out[t] = acc;
}
__global__ __launch_bounds__(256) void merged_256(const float4* __restrict__ rec256, const uint32_t* __restrict__ idx, uint32_t n, float* __restrict__ out) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const float4* r = rec256 + (uint64_t)idx[t] * 16u;
    float4 v[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) v[k] = r[k];
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 14; ++k) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    out[t] = acc;
}

int main(int argc, char** argv) {
    const uint32_t N = 6100000;
    const double p = argc > 1 ? atof(argv[1]) : 0.4;
    std::vector<uint32_t> h;
    std::mt19937 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (uint32_t i = 0; i < N; ++i) if (U(rng) < p) h.push_back(i);
    const uint32_t n = (uint32_t)h.size();
    float4* rec; uint32_t* idx; float* out;
    CHECK(hipMalloc((void**)&rec, (size_t)N * 192)); CHECK(hipMalloc((void**)&idx, (size_t)n * 4)); CHECK(hipMalloc((void**)&out, (size_t)n * 4));
    CHECK(hipMemset(rec, 0, (size_t)N * 192));
    CHECK(hipMemcpy(idx, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    void* flush; CHECK(hipMalloc(&flush, 1ull << 30));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid((n + 255) / 256), block(256);
    float4 *geo, *rec256;
    CHECK(hipMalloc((void**)&geo, (size_t)N * 32)); CHECK(hipMalloc((void**)&rec256, (size_t)N * 256));
    CHECK(hipMemset(geo, 0, (size_t)N * 32)); CHECK(hipMemset(rec256, 0, (size_t)N * 256));
    const char* names[5] = {"per_lane", "cooperative", "coop_lds", "split 32+192", "merged 256"};
    for (int which = 0; which < 5; ++which) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipMemsetAsync(flush, rep, 1ull << 30, 0)); // evict the records from the 256 MB Infinity Cache
            CHECK(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL(per_lane, grid, block, 0, 0, rec, idx, n, out);
            else if (which == 1) hipLaunchKernelGGL(cooperative, grid, block, 0, 0, rec, idx, n, out);
            else if (which == 2) hipLaunchKernelGGL(coop_lds, grid, block, 0, 0, rec, idx, n, out);
            else if (which == 3) hipLaunchKernelGGL(split_arrays, grid, block, 0, 0, geo, rec, idx, n, out);
            else hipLaunchKernelGGL(merged_256, grid, block, 0, 0, rec256, idx, n, out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double useful = which >= 3 ? 224.0 : 192.0;
        printf("%-12s density %.2f: %u survivors x %.0f B = %.0f MB in %.1f us -> %.2f TB/s\n", names[which], p, n, useful, n * useful / 1e6, best * 1e3,
               n * useful / (best * 1e-3) / 1e12);
    }
    return 0;
}
