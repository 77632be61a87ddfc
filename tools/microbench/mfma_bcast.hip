// Semantics probe: v_mfma_f32_4x4x1_16b_f32 with CBSZ=4 / ABID=g.  Expectation: D[r] in lane l = A[lane 4g + r] * B[lane l] + C.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    const float a = 100.0f + l, b = 1.0f + 0.001f * l;
    f4 c = {0, 0, 0, 0};
    f4 d5 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 5, 0);
    f4 d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { out[l * 8 + r] = d5[r]; out[l * 8 + 4 + r] = d0[r]; }
}
int main() {
    float* d; hipMalloc(&d, 64 * 8 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[64 * 8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad5 = 0, bad0 = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const float b = 1.0f + 0.001f * l;
        if (h[l * 8 + r] != (100.0f + 20 + r) * b) ++bad5;                 // broadcast block 5 -> lanes 20..23
        if (h[l * 8 + 4 + r] != (100.0f + (l & ~3) + r) * b) ++bad0;       // no broadcast: own block's lanes
    }
    printf("cbsz=4 abid=5 mismatches %d; cbsz=0 mismatches %d; lane 9: %g %g %g %g | %g %g %g %g\n", bad5, bad0, h[72], h[73], h[74], h[75], h[76], h[77], h[78], h[79]);
    return 0;
}
