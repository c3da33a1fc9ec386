// unit check of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 operands): operand layout, scale semantics, issue rate.
// hipcc --offload-arch=gfx950 -O2 tools/unit/t_mfma_fp8.hip -o build/t_mfma_fp8 && build/t_mfma_fp8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// e4m3fn encode of a value from a small exactly representable set
__host__ __device__ static unsigned char enc(float v) {
    unsigned char s = v < 0 ? 0x80 : 0;
    float a = fabsf(v);
    if (a == 0.f) return s;
    int e = 0;
    while (a >= 2.f) { a *= 0.5f; ++e; }
    while (a < 1.f) { a *= 2.f; --e; }
    int m = (int)((a - 1.f) * 8.f + 0.5f);
    return s | (unsigned char)(((e + 7) << 3) | m);
}

__global__ void k_layout(const unsigned char* A, const unsigned char* B, float* D, int sa, int sb) {
    const int l = threadIdx.x, r = l & 15, kb = l >> 4;
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        unsigned wa = 0, wb = 0;
        for (int j = 0; j < 4; ++j) {
            wa |= (unsigned)A[r * 128 + 32 * kb + 4 * i + j] << (8 * j);
            wb |= (unsigned)B[r * 128 + 32 * kb + 4 * i + j] << (8 * j);
        }
        a[i] = (int)wa; b[i] = (int)wb;
    }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 4; ++i) D[(4 * kb + i) * 16 + r] = c[i];   // D[m = 4g+i][n = l&15]
}

__global__ void k_rate(float* out, long* cyc, int iters) {
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x38383838 - threadIdx.x; }
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 0, 0, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 0, 0, 0, 127, 0, 127);
        c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 0, 0, 0, 127, 0, 127);
        c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 0, 0, 0, 127, 0, 127);
    }
    long t1 = __builtin_readcyclecounter();
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(1.f + threadIdx.x); hb[i] = (_Float16)2.f; }
    v4f d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    long t2 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d3, 0, 0, 0);
    }
    long t3 = __builtin_readcyclecounter();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[1] + d2[2] + d3[3];
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
}

int main() {
    const float vals[] = {0.f, 1.f, -1.f, 2.f, 0.5f, -1.5f, 3.f, -0.25f, 1.75f};
    unsigned char hA[16 * 128], hB[16 * 128];
    float fA[16 * 128], fB[16 * 128];
    srand(3);
    for (int i = 0; i < 16 * 128; ++i) {
        fA[i] = vals[rand() % 9]; fB[i] = vals[rand() % 9];
        hA[i] = enc(fA[i]); hB[i] = enc(fB[i]);
    }
    unsigned char *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int trial = 0; trial < 3; ++trial) {
        const int sa = trial == 0 ? 127 : trial == 1 ? 127 - 3 : 127, sb = trial == 2 ? 127 + 2 : 127;
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, sb);
        float hD[256];
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        const float scale = ldexpf(1.f, (sa - 127) + (sb - 127));
        double emax = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double s = 0;
                for (int k = 0; k < 128; ++k) s += (double)fA[m * 128 + k] * fB[n * 128 + k];
                emax = fmax(emax, fabs(hD[m * 16 + n] - s * scale));
            }
        printf("layout (lane = row, 32 consecutive k per lane group), scale_a %d scale_b %d: max |D - ref| = %g\n", sa, sb, emax);
    }
    float* dout; long* dc;
    hipMalloc(&dout, 64 * 4); hipMalloc(&dc, 16);
    hipLaunchKernelGGL(k_rate, dim3(1), dim3(64), 0, 0, dout, dc, 1000);
    long hc[2];
    hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    printf("cycles per instruction (one wave, 4 independent chains): fp8 16x16x128 scaled %.1f, f16 16x16x32 %.1f\n",
           hc[0] / 4000.0, hc[1] / 4000.0);
    return 0;
}
