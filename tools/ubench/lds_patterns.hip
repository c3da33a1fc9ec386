// LDS access-pattern microbenchmark (gfx950): cycles per wave-instruction for the layouts the kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int BYTES, bool WRITE>
__global__ __launch_bounds__(256) void k(const int* __restrict__ offs, long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int o = offs[threadIdx.x & 63] + (threadIdx.x >> 6) * 16384;   // each wave its own 16 KiB window
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    f32x4 v[8];
    for (int u = 0; u < 8; ++u) v[u] = f32x4{1, 2, 3, 4};
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
        if (WRITE) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (BYTES == 16) asm volatile("ds_write_b128 %0, %1" ::"v"(o), "v"(v[u]) : "memory");
                if (BYTES == 8) asm volatile("ds_write_b64 %0, %1" ::"v"(o), "v"(f32x2{v[u][0], v[u][1]}) : "memory");
                if (BYTES == 4) asm volatile("ds_write_b32 %0, %1" ::"v"(o), "v"(v[u][0]) : "memory");
                if (BYTES == 2) asm volatile("ds_write_b16 %0, %1" ::"v"(o), "v"(v[u][0]) : "memory");
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (BYTES == 16) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(o) : "memory");
                if (BYTES == 8) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(o) : "memory"); v[u][0] = t[0]; }
                if (BYTES == 4) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(o) : "memory"); v[u][0] = t; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    float a = 0;
    for (int u = 0; u < 8; ++u) a += v[u][0] + v[u][1] + v[u][2] + v[u][3];
    sink[threadIdx.x] = a + reinterpret_cast<float*>(lds)[threadIdx.x];
}

template <int BYTES, bool WRITE>
static void run(const char* name, const std::vector<int>& offs) {
    int* d; long long* o; float* s;
    hipMalloc(&d, 64 * 4); hipMalloc(&o, 8); hipMalloc(&s, 256 * 4);
    hipMemcpy(d, offs.data(), 64 * 4, hipMemcpyHostToDevice);
    long long best = 1LL << 60;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL((k<BYTES, WRITE>), dim3(1), dim3(256), 0, 0, d, o, s);
        long long h; hipMemcpy(&h, o, 8, hipMemcpyDeviceToHost);
        if (h < best) best = h;
    }
    printf("%-58s %s b%-3d : %6.2f clk/inst\n", name, WRITE ? "write" : "read ", BYTES * 8, (double)best / (256 * 8));   // 4 waves x 8 instructions per iteration: divide by 4 for per-CU throughput
    hipFree(d); hipFree(o); hipFree(s);
}

int main() {
    auto gen = [](auto f) { std::vector<int> v(64); for (int l = 0; l < 64; ++l) v[l] = f(l); return v; };
    run<16, false>("contiguous lane*16", gen([](int l) { return l * 16; }));
    for (int T : {16, 48}) {
        for (int S = 16; S <= 400; S += 16) {
            char nm[64];
            snprintf(nm, sizeof nm, "read m*%d + %d*g", S, T);
            run<16, false>(nm, gen([=](int l) { return (l & 15) * S + (l >> 4) * T; }));
        }
    }
    for (int S = 16; S <= 336; S += 16) {
        char nm[64];
        snprintf(nm, sizeof nm, "write b128 lane*%d", S);
        run<16, true>(nm, gen([=](int l) { return l * S; }));
    }
    for (int S = 8; S <= 104; S += 8) {
        char nm[64];
        snprintf(nm, sizeof nm, "write b64 m*%d + 8g", S);
        run<8, true>(nm, gen([=](int l) { return (l & 15) * S + (l >> 4) * 8; }));
    }
    return 0;
}
