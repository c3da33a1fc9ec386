// ldm_ops.h — internal interface of the latent-diffusion U-Net primitives (ldm_ops.hip)
#pragma once
#include "common.h"

// Source 0 of a GroupNorm handed over as the raw split-K partial sums of the convolution that produces it (conv.hip
// ConvLaunch::splits_out): the statistics kernels add them up (+ bias + residual), write the finished tensor to `fin`
// and normalise / tabulate from the finished values.
struct GnPartial {
    const float* part;    // [nsplit][N * HW][C0]
    int nsplit;
    const float* shift;   // [C0] or NULL
    const float* res;     // (N, HW, C0) or NULL
    float* fin;           // (N, HW, C0) finished output
};
int launch_group_norm(const float* x, const float* gamma, const float* beta, const float* film, float* y, float* stats,
                      int N, int HW, int C, int groups, float eps, int silu, hipStream_t stream,
                      const float* x1 = nullptr, int C0 = 0,    // x1: second source, channels [C0, C) of the input
                      long film_ld = 0,                         // floats between the film rows of consecutive images (0: 2C)
                      float* table = nullptr,                   // non-NULL: write the affine table [N][2][C] instead of y (ldm_ops.hip)
                      const GnPartial* part = nullptr);         // non-NULL: source 0 = split-K partials (x is ignored)
int launch_qkv_attention(const float* qkv, float* out, int N, int T, int heads, int ch, int prec, hipStream_t stream);
// ldm_attn.hip: long-sequence attention on the f16 MFMA with fp32-class logits; workspace = pre-split K / V block images
size_t qkv_attention_ws_bytes(int N, int T, int heads, int ch);   // 0: head width not served (use launch_qkv_attention)
int launch_qkv_attention_ws(const float* qkv, float* out, int N, int T, int heads, int ch, void* ws, size_t ws_bytes,
                            hipStream_t stream);
int launch_resample2x(const float* x, float* y, int N, int H, int W, int C, int up, hipStream_t stream);
int launch_small_linear(const float* x, const float* w, const float* b, float* out, int N, int K, int M, int silu_in,
                        hipStream_t stream);
int launch_timestep_embedding(const float* t, float* out, int N, int dim, float max_period, hipStream_t stream);
int launch_add(const float* a, const float* b, float* out, long n, hipStream_t stream);
