// Does v_mfma_f32_16x16x32_f16 honour subnormal f16 inputs?  And what does the f32 -> f16 conversion give for them?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float a, float b, float* out) {
    half8 A, B;
    for (int t = 0; t < 8; ++t) { A[t] = (_Float16)a; B[t] = (_Float16)b; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)A[0]; out[2] = (float)B[0]; }
    // residual split of a typical small value
    const float v = 0.0731f * a / 3e-5f;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    if (threadIdx.x == 0) { out[3] = v; out[4] = (float)h; out[5] = (float)l; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    const float tests[][2] = {{3e-5f, 1.f}, {1.f, 3e-5f}, {3e-5f, 3e-5f}, {1e-6f, 1000.f}, {6.2e-5f, 1.f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, t[0], t[1], d);
        float h[6]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("a=%g b=%g : mfma sum(32 products) = %g (expected %g); A as f16 = %g, B as f16 = %g | split of %g: hi %g lo %g\n",
               t[0], t[1], h[0], 32.0 * (double)h[1] * (double)h[2], h[1], h[2], h[3], h[4], h[5]);
    }
    return 0;
}
