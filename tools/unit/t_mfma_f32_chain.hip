// Unit check (gfx950): is v_mfma_f32_16x16x4_f32 bitwise a k-ordered fmaf chain, and is it insensitive to exact-zero
// terms inserted anywhere in the chain?  The shared-footprint token builder (decode.hip, sample_tokens_kernel) relies on both:
// a query's bilinear taps sit at different k positions of the group's pixel list depending on its group mates, and the
// per-lane VALU fallback path must produce the same bits.
//   hipcc --offload-arch=gfx950 -O2 tools/unit/t_mfma_f32_chain.hip -o /tmp/t_chain && /tmp/t_chain
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// A [16][K], B [K][16], C [16][16] row-major;  D = chain over k-blocks of 4
__global__ void mfma_chain_kernel(const float* A, const float* B, const float* C, float* D, int K) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    f32x4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = C[(4 * g + i) * 16 + r];   // D[row = 4g+i][col = r]
    for (int kb = 0; kb < K / 4; ++kb) {
        const float a = A[r * K + 4 * kb + g];        // A[i = l&15][k = l>>4]
        const float b = B[(4 * kb + g) * 16 + r];     // B[k = l>>4][j = l&15]
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = acc[i];
}

static float frand() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }

int main() {
    const int K = 16, trials = 2000;
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 16 * K * 4); hipMalloc(&dB, K * 16 * 4); hipMalloc(&dC, 256 * 4); hipMalloc(&dD, 256 * 4);
    std::vector<float> A(16 * K), B(K * 16), C(256), D(256);
    long bad_chain = 0, bad_pos = 0, total = 0;
    srand(1234);
    for (int t = 0; t < trials; ++t) {
        // every output column j has exactly 4 nonzero weights at random ascending k positions (a bilinear footprint inside a
        // pixel list), the rest exact zeros
        std::vector<int> pos(16 * 4);
        for (auto& v : A) v = frand() * ((t & 1) ? 50.f : 1.f);
        for (auto& v : C) v = frand();
        std::fill(B.begin(), B.end(), 0.f);
        for (int j = 0; j < 16; ++j) {
            int p[4];
            for (;;) {
                for (int q = 0; q < 4; ++q) p[q] = rand() % K;
                bool ok = true;
                for (int a = 0; a < 4; ++a) for (int b = a + 1; b < 4; ++b) ok &= p[a] != p[b];
                if (ok) break;
            }
            for (int a = 0; a < 4; ++a) for (int b = a + 1; b < 4; ++b) if (p[b] < p[a]) { int s = p[a]; p[a] = p[b]; p[b] = s; }
            for (int q = 0; q < 4; ++q) { pos[j * 4 + q] = p[q]; B[p[q] * 16 + j] = fabsf(frand()); }
        }
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_chain_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, K);
        hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float full = C[i * 16 + j], sparse = C[i * 16 + j];
                for (int k = 0; k < K; ++k) full = fmaf(A[i * K + k], B[k * 16 + j], full);          // every term, zeros included
                for (int q = 0; q < 4; ++q) { const int k = pos[j * 4 + q]; sparse = fmaf(A[i * K + k], B[k * 16 + j], sparse); }   // the 4 taps only
                ++total;
                if (memcmp(&full, &D[i * 16 + j], 4)) ++bad_chain;
                if (memcmp(&sparse, &D[i * 16 + j], 4)) ++bad_pos;
            }
    }
    printf("t_mfma_f32_chain: %ld outputs, %ld differ from the full fmaf chain, %ld differ from the 4-tap fmaf chain\n", total, bad_chain, bad_pos);
    return (bad_chain || bad_pos) ? 1 : 0;
}
