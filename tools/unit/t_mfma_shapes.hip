// Energy per MAC of the two f16 MFMA shapes under the socket's power cap: register-resident loops of
// v_mfma_f32_16x16x32_f16 and v_mfma_f32_32x32x16_f16 on random (normal) operands, every SIMD of the chip busy with two waves.
//   hipcc --offload-arch=gfx950 -O2 tools/unit/t_mfma_shapes.hip -o build/t_mfma_shapes
//   build/t_mfma_shapes <shape 16|32> <seconds> [zero]       prints TFLOP/s per launch; sample rocm-smi beside it
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <chrono>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k16(const h8* __restrict__ src, float* out, int iters) {
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 256 * i) & 4095]; b[i] = src[(threadIdx.x * 7 + 911 * i) & 4095]; }
    v4f c[16];
    for (int i = 0; i < 16; ++i) c[i] = v4f{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], c[4 * i + j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(const h8* __restrict__ src, float* out, int iters) {
    h8 a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = src[(threadIdx.x + 256 * i) & 4095]; b[i] = src[(threadIdx.x * 7 + 911 * i) & 4095]; }
    v16f c[4];
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 16; ++k) c[i][k] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)   // same MACs per trip as k16: 16 x 8192 = 2 x 4 x 16384
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) c[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], c[2 * i + j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 16;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    const bool zero = argc > 3;
    h8* d; float* o;
    hipMalloc(&d, 4096 * sizeof(h8)); hipMalloc(&o, 2048 * 256 * 4);
    h8* h = (h8*)malloc(4096 * sizeof(h8));
    srand(5);
    for (int i = 0; i < 4096; ++i)
        for (int t = 0; t < 8; ++t) {
            const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = rand() / (double)RAND_MAX;
            h[i][t] = zero ? (_Float16)0.f : (_Float16)(0.05 * sqrt(-2 * log(u1)) * cos(6.2831853 * u2));
        }
    hipMemcpy(d, h, 4096 * sizeof(h8), hipMemcpyHostToDevice);
    const int iters = 20000, blocks = 2048;   // 2 workgroups per CU x 4 waves
    const double flop = 2.0 * blocks * 4 * iters * 16.0 * 8192.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    double best = 0; int n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0);
        if (shape == 16) hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, d, o, iters);
        else hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double tf = flop / (ms * 1e-3) / 1e12;
        if (n++ > 0 && tf > best) best = tf;
        if (n % 8 == 0) { printf("%s %dx: %.1f ms per launch = %.0f TFLOP/s\n", zero ? "zero" : "random", shape, ms, tf); fflush(stdout); }
    }
    printf("best %.0f TFLOP/s\n", best);
    return 0;
}
