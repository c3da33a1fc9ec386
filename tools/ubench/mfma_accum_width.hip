// How many bits does v_mfma_f32_16x16x32_f16 keep when it sums products of different magnitude?
// One big product (BIG * 1) and 31 small ones ((4 + 2^-8) * (1 + 2^-10), exact value needs 2^-18 resolution).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float big, float cinit, float* out) {
    // A row (lane m = 0..15 all the same): k-slot 0 = big, others = 4 + 2^-8 ; B col: slot 0 = 1, others 1 + 2^-10
    const int g = threadIdx.x >> 4;
    half8 A, B;
    for (int t = 0; t < 8; ++t) {
        const bool first = (g == 0 && t == 0);
        A[t] = first ? (_Float16)big : (_Float16)4.00390625f;
        B[t] = first ? (_Float16)1.f : (_Float16)1.0009765625f;
    }
    f32x4 c = {cinit, cinit, cinit, cinit};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 64);
    const double small = 4.00390625 * 1.0009765625;
    for (float big : {0.f, 4.f, 64.f, 750.f, 16384.f, 60000.f})
        for (float c0 : {0.f, 100000.f}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, big, c0, d);
            float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
            const double exact = (double)c0 + (double)(float)(_Float16)big + 31 * small;
            printf("big %8g C %8g: mfma %.9g exact %.9g  diff %.3e  (fp32 ulp of result %.3e)\n", big, c0, h, exact, h - exact,
                   std::ldexp(1.0, (int)std::floor(std::log2(std::fabs(exact) + 1e-30)) - 23));
        }
    return 0;
}
