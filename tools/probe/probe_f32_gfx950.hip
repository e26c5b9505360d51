// Dev probe (GPU box): fragment layout of v_mfma_f32_16x16x4_f32 (one wave): which (row, col) does lane l, register r hold?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    // hypothesis for the operands (as for the f64 variant): a = A[i = l & 15][k = l >> 4], b = B[k = l >> 4][j = l & 15]
    // choose A[i][k] = (k == 0) ? i + 1 : 0 and B[k][j] = (k == 0) ? 100 * (j + 1) : 0  ->  D[i][j] = (i + 1) * 100 * (j + 1)
    const float a = (l >> 4) == 0 ? float((l & 15) + 1) : 0.f;
    const float b = (l >> 4) == 0 ? 100.f * float((l & 15) + 1) : 0.f;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const float v = h[l * 4 + r];
            const int j = l & 15;
            const int iA = 4 * (l >> 4) + r;   // hypothesis A: rows 4*(l/16) + r
            const int iB = (l >> 4) + 4 * r;   // hypothesis B: rows (l/16) + 4*r  (the f64 layout)
            okA &= v == (iA + 1) * 100.f * (j + 1);
            okB &= v == (iB + 1) * 100.f * (j + 1);
        }
    printf("layout rows=4*(l>>4)+r: %s ; rows=(l>>4)+4*r: %s\n", okA ? "MATCH" : "no", okB ? "MATCH" : "no");
    printf("lane 0: %g %g %g %g ; lane 16: %g %g %g %g ; lane 17: %g %g\n", h[0], h[1], h[2], h[3], h[64], h[65], h[66], h[67],
           h[68], h[69]);
    return 0;
}
