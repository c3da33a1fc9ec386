// attn_last.h — the mixing step of the last layer's absorbed token-0 attention (decode.h: launch_attn_last_mix), shared by the
// four-launch form (decode.hip: attn_last_mix_kernel) and the fused kernel (decode_last.hip) so that both produce the same bits.
#pragma once
#include "common.h"

// sum over the 16 lanes of a DPP row (the 16 channel octets of one query): quad xor 1, quad xor 2, half mirror, row mirror
__device__ __forceinline__ float al_row16_allsum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
    return v;
}

// One head for one query, seen by the lane that holds channels c0 .. c0 + 7 of the query's T token rows (xa | xb) and of the
// head's absorbed query vector qt_h = M_h x0 + m_h (qa | qb):
//   score_t = qt_h . x_t / sqrt(32)  (16-lane all-reduce),  p = softmax_t(score),  xbar_h = sum_t p_t x_t  ->  oa | ob.
// Round 6: 290 VALU instructions instead of 540 — the partial dot products as packed fp32 ops, the scores in log2 units
// (the scale carries log2 e) so that the 13 exponentials are one v_exp_f32 each instead of expf's 14-instruction range reduction,
// one select per token (a masked token's score of -1e30 makes its exponential exactly 0).  The kernel that runs this is bound by
// VALU issue (SQ_ACTIVE_INST_VALU = 50 % of its SIMD cycles before the change).
#define S3D_AL_TOKENS 13   // = S3D_N_TOKENS_MAX
__device__ __forceinline__ void al_mix_head(const f32x4 (&xa)[S3D_AL_TOKENS], const f32x4 (&xb)[S3D_AL_TOKENS], const f32x4 qa,
                                            const f32x4 qb, int T, f32x4& oa, f32x4& ob) {
    const float scale = 0.17677669529663687f * 1.4426950408889634f;   // log2(e) / sqrt(32)
    float sc[S3D_AL_TOKENS];
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < S3D_AL_TOKENS; ++t) {
        const f32x4 p = __builtin_elementwise_fma(qb, xb[t], qa * xa[t]);
        float d = (p[0] + p[1]) + (p[2] + p[3]);
        d = al_row16_allsum(d) * scale;
        sc[t] = t < T ? d : -1e30f;
        mx = fmaxf(mx, sc[t]);
    }
    float den = 0.f;
#pragma unroll
    for (int t = 0; t < S3D_AL_TOKENS; ++t) {
        sc[t] = __builtin_amdgcn_exp2f(sc[t] - mx);   // masked tokens: exp2(-1e30 - mx) = 0
        den += sc[t];
    }
    const float inv = 1.f / den;
    oa = zero4();
    ob = zero4();
#pragma unroll
    for (int t = 0; t < S3D_AL_TOKENS; ++t) {
        const float pt = sc[t] * inv;
        const f32x4 p4 = {pt, pt, pt, pt};
        oa = __builtin_elementwise_fma(xa[t], p4, oa);
        ob = __builtin_elementwise_fma(xb[t], p4, ob);
    }
}
