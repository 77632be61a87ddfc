// VALU issue-rate probe for gfx950: how many cycles does a SIMD spend per wave64 v_fma_f32 / v_pk_fma_f32 /
// v_exp_f32 / v_cndmask when 1..8 waves share it?  (hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int KIND> __global__ __launch_bounds__(64) void probe(float* out, int iters, long long* cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, b1 = 0.9999f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
        } else if (KIND == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
        } else if (KIND == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 3) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_cmp_le_f32 vcc, %8, %0\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_le_f32 vcc, %8, %2\n v_cndmask_b32 %3, %3, %9, vcc\n"
                             "v_cmp_le_f32 vcc, %8, %4\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_le_f32 vcc, %8, %6\n v_cndmask_b32 %7, %7, %9, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : "vcc");
        } else if (KIND == 4) { // pk_mul + pk_add mix
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
        }
    }
    if (KIND >= 5) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 c0 = {a0, a1, a2, a3}, c1 = c0, c2 = c0, c3 = c0;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                // 4 independent MFMAs (broadcast A from block k) [+ 12 independent v_fma for KIND 6]
                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4, a5, c0, 4, 3, 0);
                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4, a6, c1, 4, 5, 0);
                c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a4, a7, c2, 4, 7, 0);
                c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a5, a7, c3, 4, 9, 0);
                if (KIND == 6)
                    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                                 "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                                 "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
            }
        }
        a0 += c0.x + c0.y + c0.z + c0.w + c1.x + c2.y + c3.z + c1.w + c2.x + c3.x;
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    CHECK(hipMalloc(&out, 1 << 24)); CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 2000; const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "cmp+cndmask", "pk_mul+pk_add", "mfma4x4x1 (x64)", "4 mfma + 12 fma (x16: 64 mfma, 192 fma)"};
    for (int kind = 0; kind < 7; ++kind)
        for (int wps = 1; wps <= 8; wps *= 2) { // waves per SIMD: grid = 256 CUs * 4 SIMDs * wps single-wave workgroups
            const int grid = 256 * 4 * wps; float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                switch (kind) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(64), 0, 0, out, iters, cyc); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(64), 0, 0, out, iters, cyc); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(64), 0, 0, out, iters, cyc); break;
                case 3: hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(64), 0, 0, out, iters, cyc); break;
                case 4: hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(64), 0, 0, out, iters, cyc); break;
                case 5: hipLaunchKernelGGL(probe<5>, dim3(grid), dim3(64), 0, 0, out, iters, cyc); break;
                case 6: hipLaunchKernelGGL(probe<6>, dim3(grid), dim3(64), 0, 0, out, iters, cyc); break;
                }
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            long long h = 0; CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            const double n_inst = 64.0 * iters; // wave-instructions per wave
            // SIMD time per instruction = elapsed / (instructions issued on one SIMD) assuming an even spread
            printf("%-14s waves/SIMD=%d  elapsed %.3f ms  ns per wave-instr per SIMD %.3f  (cyc@2.4GHz %.2f)  wave0 counter ticks/instr %.2f\n", names[kind], wps, ms,
                   ms * 1e6 / (n_inst * wps), ms * 1e6 / (n_inst * wps) * 2.4, (double)h / n_inst);
        }
    return 0;
}
