// mfma_shape_probe.hip — does the 32x32x16 f16 MFMA sustain more FLOP/s than 16x16x32 at this part's power cap?
// The decoder's FFN and attention kernels run 16x16x32 tiles and are bounded by the socket's power budget, not by issue
// slots; this probe runs both shapes for seconds, at the same accumulator count (64 registers), the
// same rows per wave (64) and — in the LDS variants — the same B-fragment reads per FLOP as the FFN kernel, on random
// operands, and prints TFLOP/s per variant.  tools/mfma_shape_probe.py samples clock and power beside it.
//   build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o /tmp/mfma_shape_probe
//   run:   /tmp/mfma_shape_probe <variant 0..3> <seconds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ h8 rnd8(unsigned& s) {
    h8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u;
        v[i] = (_Float16)(((int)(s >> 8) & 0xffff) * (1.f / 32768.f) - 1.f);
    }
    return v;
}

// V = 0: 16x16x32, 4 x 4 outer product of register fragments      V = 1: 32x32x16, 2 x 2 (twice per iteration)
// V = 2 / 3: the same with the B fragments re-read from LDS every iteration (the FFN kernel's weight stream)
template <int V>
__global__ __launch_bounds__(256, 2) void probe_kernel(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 s_b[8 * 64 * 8];
    unsigned seed = blockIdx.x * 256 + threadIdx.x + 12345u;
    const int lane = threadIdx.x & 63;
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = rnd8(seed);
        b[i] = rnd8(seed);
    }
    if (threadIdx.x < 64) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<h8*>(s_b + (i * 64 + lane) * 8) = rnd8(seed);
    }
    __syncthreads();
    float sum = 0.f;
    if (V == 0 || V == 2) {
        f4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            if (V == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const volatile h8*>(s_b + (((it & 1) * 4 + j) * 64 + lane) * 8);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        f16v acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            if (V == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const volatile h8*>(s_b + (((it & 1) * 4 + j) * 64 + lane) * 8);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)   // two K = 16 steps: the same 32 of K per iteration as the 16x16x32 variant
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * kk + i], b[2 * kk + j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));             \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 2, iters = 20000;
    float* out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&]() {
        switch (variant) {
            case 0: hipLaunchKernelGGL(probe_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
            case 1: hipLaunchKernelGGL(probe_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
            case 2: hipLaunchKernelGGL(probe_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
            default: hipLaunchKernelGGL(probe_kernel<3>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
        }
    };
    launch();
    CK(hipDeviceSynchronize());
    // 16 MFMAs x 16 384 FLOP = 8 MFMAs x 32 768 FLOP per wave and iteration
    const double flop_per_launch = (double)blocks * 4 * iters * 16 * 16384.0;
    double total_ms = 0;
    int launches = 0;
    std::vector<float> per;
    while (total_ms < seconds * 1e3) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 4; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        per.push_back(ms / 4);
        total_ms += ms;
        launches += 4;
    }
    // the last half of the run: the power controller has settled
    double tail = 0;
    const size_t h = per.size() / 2;
    for (size_t i = h; i < per.size(); ++i) tail += per[i];
    tail /= (double)(per.size() - h);
    printf("{\"variant\": %d, \"shape\": \"%s\", \"b_from_lds\": %s, \"launches\": %d, \"ms_per_launch_first\": %.4f, \"ms_per_launch_settled\": %.4f, "
           "\"tflops_first\": %.1f, \"tflops_settled\": %.1f}\n",
           variant, (variant & 1) ? "32x32x16" : "16x16x32", variant >= 2 ? "true" : "false", launches, per[0], tail,
           flop_per_launch / (per[0] * 1e-3) * 1e-12, flop_per_launch / (tail * 1e-3) * 1e-12);
    return 0;
}
