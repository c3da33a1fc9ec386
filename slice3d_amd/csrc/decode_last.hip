// decode_last.hip — the last encoder layer's attention block for token 0 in ONE kernel (round 6).
//
// models.py:83 consumes only token 0 of the last layer, and decode.h describes the absorbed form that makes the layer
// two dense GEMMs around a 13-row mixing step:   qt = M x0 + m  (128 -> 512),   xbar_h = sum_t softmax_t(qt_h . x_t / sqrt 32) x_t,
// u = N xbar + n + x0  (512 -> 128).  Until round 5 these were four launches (tok0_copy, a row GEMM, attn_last_mix_kernel,
// a row GEMM) that wrote and re-read x0, qt (2 KB per query) and xbar (2 KB per query): 1.5 ms per 400 k queries for an
// operator whose only large operand is the 13 token rows (6.7 KB per query, 2.66 GB per step).  Here a workgroup owns NG
// groups of 16 queries at a time and nothing but the token rows (in) and the u rows (out, 512 B per query) touches HBM:
//
//   phase 1  wave w = head w:  qt[:, 128 w .. +127] of the NG x 16 queries on the split-precision MFMA (A = M fragments
//            straight from the packed image in L2, B = the x0 tile split in registers), + m, parked in LDS as fp32;
//   phase 2  round r = group r: wave w mixes queries 4 w .. 4 w + 3 (16 lanes per query, 8 channels per lane, the 13 token
//            rows in registers — the arithmetic of attn_last_mix_kernel, instruction for instruction); xbar_h is split to
//            f16 hi | lo and written OVER the query's qt_h (same 512 bytes; the 16 lanes that read qt_h are the ones that
//            write xbar_h, LDS operations of a wave execute in order) in the B-fragment order of phase 3;
//   phase 3  wave w = output channels 32 w .. 32 w + 31:  u = N xbar + n + x0 (K = 512: 16 k-steps, A = N fragments from
//            L2, B = xbar fragments from LDS), stored as the rows the final FFN kernel normalises in its prologue.
//
// Every accumulator adds its products in the order of the row-GEMM kernels it replaces (k ascending; hi*lo, lo*hi, hi*hi;
// then + bias, then + residual) and the mixing step is the same code: the u rows are BIT-IDENTICAL to the four-launch
// form (tests/test_gpu_parity.py::test_fused_last_layer_is_bit_identical).
//
// LDS: one 2 KB + 16 B region per query (the 16-byte pad spreads the phase-3 fragment reads of the 16 queries over the
// banks); NG = 2: 66 KB per workgroup, two workgroups per CU.  The weight fragments (M, N: 256 KB each as hi | lo) are read
// from L2 once per NG groups: 512 KB per 32 queries = 6.4 GB per 400 k queries at L2 bandwidth beside the 2.66 GB of rows
// from HBM.
#include "decode.h"
#include "attn_last.h"

typedef _Float16 lhalf8 __attribute__((ext_vector_type(8)));

#define AL_QS (2048 + 16)   // bytes per query region
#ifndef AL_D1
#define AL_D1 8       // depth (k-steps) of phase 1's weight-fragment ring
#endif
#ifndef AL_D3
#define AL_D3 4       // depth (k-steps, two output tiles each) of phase 3's ring
#endif

template <bool SINGLE, int NG>
__global__ __launch_bounds__(256, 2) void attn_last_fused_kernel(const float* __restrict__ X, float* __restrict__ U,
                                                                 long groups, int T, const _Float16* __restrict__ wm16,
                                                                 const float* __restrict__ bm,
                                                                 const _Float16* __restrict__ wn16,
                                                                 const float* __restrict__ bn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_q[];   // [NG * 16][AL_QS]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int mq = threadIdx.x >> 4, s16 = threadIdx.x & 15, c0 = s16 * 8;   // phase 2: query of the group, channel octet
    const long n_it = (groups + NG - 1) / NG;
    // Load schedule.  SQ counters say what bounds this kernel: 37 - 53 % of its wave cycles sit in s_waitcnt, the VALU is 22 % and the
    // matrix pipe 15 % busy — and halving the mixing step's VALU count, requesting round 0's rows before phase 1, deepening the
    // weight-fragment rings from 1 to 12 k-steps each left its time at 0.88 ms per 400 k queries (profiles/r06_last_layer.md).  vmcnt
    // retires in order: a wave that has row loads (HBM, ~2 us under load) in flight and then waits for a weight fragment (L2) requested
    // after them waits for the rows as well, so inside ONE wave the three load streams — x0 tile, rows, weight fragments — cannot hide
    // each other; only the CU's other workgroup does (two per CU: the 104 registers of a round's rows and 66 KB of LDS allow no third).
    // Kept from those experiments because they cost nothing: the x0 tile of iteration i + 1 is requested at the top of phase 3 of
    // iteration i, and the weight fragments run through register rings of AL_D1 / AL_D3 k-steps.
    static_assert(NG == 2, "the load schedule below is written for two groups per iteration");
    f32x4 v0[NG][4], v1[NG][4];   // raw x0 tile of the current iteration (lane (m, g): channels 32 u + 8 g .. + 7 of query m)
    auto load_x0 = [&](long it_) {
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) {
            long grp = it_ * NG + ng;
            if (grp >= groups) grp = groups - 1;
            const float* row = X + (grp * T * S3D_GROUP + m) * 128 + 8 * g;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v0[ng][u] = ld4(row + 32 * u);
                v1[ng][u] = ld4(row + 32 * u + 4);
            }
        }
    };
    auto load_rows = [&](long grp, f32x4 (&xa)[S3D_N_TOKENS_MAX], f32x4 (&xb)[S3D_N_TOKENS_MAX]) {
        if (grp >= groups) grp = groups - 1;
        const float* xg = X + (grp * T * S3D_GROUP + mq) * 128 + c0;
#pragma unroll
        for (int t = 0; t < S3D_N_TOKENS_MAX; ++t) {
            const int tc = t < T ? t : T - 1;
            xa[t] = ld4(xg + (long)tc * S3D_GROUP * 128);
            xb[t] = ld4(xg + (long)tc * S3D_GROUP * 128 + 4);
        }
    };
    // the 13-row mixing step of one group: this wave's queries 4 wave .. + 3; xbar_h over qt_h in the B-fragment order of phase 3
    auto mix_round = [&](int r, const f32x4 (&xa)[S3D_N_TOKENS_MAX], const f32x4 (&xb)[S3D_N_TOKENS_MAX]) {
        unsigned char* qreg = s_q + (r * 16 + mq) * AL_QS;
#pragma unroll 1
        for (int h = 0; h < 4; ++h) {
            const f32x4 qa = *reinterpret_cast<const f32x4*>(qreg + (h * 128 + c0) * 4);
            const f32x4 qb = *reinterpret_cast<const f32x4*>(qreg + (h * 128 + c0 + 4) * 4);
            f32x4 oa, ob;
            al_mix_head(xa, xb, qa, qb, T, oa, ob);   // attn_last.h: the arithmetic of attn_last_mix_kernel, instruction for instruction
            // xbar_h channels c0 .. c0 + 7 = the B fragment of lane (m = mq, g = s16 & 3) of k-step 4 h + (s16 >> 2)
            const float x[8] = {oa[0], oa[1], oa[2], oa[3], ob[0], ob[1], ob[2], ob[3]};
            s3d_half8 hh, ll;
            s3d_split8(x, hh, ll);
            unsigned char* d = qreg + (4 * h + (s16 >> 2)) * 128 + (s16 & 3) * 16;
            *reinterpret_cast<s3d_half8*>(d) = hh;
            if (!SINGLE) *reinterpret_cast<s3d_half8*>(d + 64) = ll;
        }
    };
    if ((long)blockIdx.x < n_it) load_x0(blockIdx.x);
#pragma unroll 1
    for (long it = blockIdx.x; it < n_it; it += gridDim.x) {
        // the x0 tile (requested an iteration ago) becomes f16 hi | lo B fragments
        lhalf8 xh[NG][4], xl[NG][4];
#pragma unroll
        for (int ng = 0; ng < NG; ++ng)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float x[8] = {v0[ng][u][0], v0[ng][u][1], v0[ng][u][2], v0[ng][u][3],
                                    v1[ng][u][0], v1[ng][u][1], v1[ng][u][2], v1[ng][u][3]};
                s3d_half8 h, l;
                s3d_split8(x, h, l);
                xh[ng][u] = __builtin_bit_cast(lhalf8, h);
                xl[ng][u] = __builtin_bit_cast(lhalf8, l);
            }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- phase 1: qt = M x0 + m for the NG x 16 queries; this wave: head `wave` ----------------
        {
            S3D_SPLIT_SETTLE();
            // scalar base + one 32-bit lane offset (global_load v, v_off, s[base]): with per-step 64-bit lane addresses hipcc forms all 32
            // of them outside the iteration loop, spills them, and every reload's s_waitcnt vmcnt(0) waits for the prefetched rows
            const char* wp = reinterpret_cast<const char*>(wm16 + (size_t)(8 * wave) * 4 * 1024);
            unsigned loff = (unsigned)lane * 16u;
            asm volatile("" : "+v"(loff));   // (opaque per iteration: the per-step offsets below are not loop invariants to be hoisted and spilled)
            // Weight fragments through a register ring D1 k-steps deep (32 steps: 8 output tiles x 4 k-steps; hi | lo of a step = 8
            // registers).  A step is 6 MFMAs (~100 cycles) and a fragment comes from L2 (600+ cycles under load).
            constexpr int D1 = AL_D1;
            lhalf8 wh[D1], wl[D1];
#pragma unroll
            for (int st = 0; st < D1; ++st) {
                wh[st] = *reinterpret_cast<const lhalf8*>(wp + (loff + (unsigned)st * 2048u));
                if (!SINGLE) wl[st] = *reinterpret_cast<const lhalf8*>(wp + (loff + (unsigned)st * 2048u + 1024u));
            }
            f32x4 acc[NG];
#pragma unroll
            for (int st = 0; st < 32; ++st) {
                const int j = st >> 2, u = st & 3, sl = st % D1;
                __builtin_amdgcn_sched_barrier(0);
                if (u == 0) {
#pragma unroll
                    for (int ng = 0; ng < NG; ++ng) acc[ng] = zero4();
                }
                if (!SINGLE) {
#pragma unroll
                    for (int ng = 0; ng < NG; ++ng)
                        acc[ng] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[sl], xl[ng][u], acc[ng], 0, 0, 0);
#pragma unroll
                    for (int ng = 0; ng < NG; ++ng)
                        acc[ng] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[sl], xh[ng][u], acc[ng], 0, 0, 0);
                }
#pragma unroll
                for (int ng = 0; ng < NG; ++ng)
                    acc[ng] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[sl], xh[ng][u], acc[ng], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (st + D1 < 32) {   // the slot is free: step st + D1
                    wh[sl] = *reinterpret_cast<const lhalf8*>(wp + (loff + (unsigned)(st + D1) * 2048u));
                    if (!SINGLE) wl[sl] = *reinterpret_cast<const lhalf8*>(wp + (loff + (unsigned)(st + D1) * 2048u + 1024u));
                }
                if (u == 3) {
                    const int co = (8 * wave + j) * 16 + 4 * g;
                    const f32x4 sh = ld4(bm + co);
#pragma unroll
                    for (int ng = 0; ng < NG; ++ng)
                        *reinterpret_cast<f32x4*>(s_q + (ng * 16 + m) * AL_QS + co * 4) = acc[ng] + sh;
                }
            }
        }
        __syncthreads();
        // ---------------- phase 2: the 13-row mixing step, round = group ----------------
        f32x4 xa[S3D_N_TOKENS_MAX], xb[S3D_N_TOKENS_MAX];
        load_rows(it * NG, xa, xb);
        mix_round(0, xa, xb);
        __builtin_amdgcn_sched_barrier(0);
        load_rows(it * NG + 1, xa, xb);
        __builtin_amdgcn_sched_barrier(0);
        mix_round(1, xa, xb);
        __syncthreads();
        // ---------------- phase 3: u = N xbar + n + x0; this wave: output channels 32 wave .. + 31 ----------------
        {
            // residual rows of this iteration's outputs = the x0 tile once more (L1 / L2), BEFORE the next iteration's tile replaces v0 / v1
            f32x4 res[NG][2], sh[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) sh[nt] = ld4(bn + (2 * wave + nt) * 16 + 4 * g);
#pragma unroll
            for (int ng = 0; ng < NG; ++ng) {
                long grp = it * NG + ng;
                if (grp >= groups) grp = groups - 1;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    res[ng][nt] = ld4(X + (grp * T * S3D_GROUP + m) * 128 + (2 * wave + nt) * 16 + 4 * g);
            }
            // the next iteration's x0 tile: in flight under this phase's MFMAs (unconditional — the last iteration re-reads its own tile: a
            // conditional load keeps the OLD tile's 64 registers live through all three phases)
            load_x0(it + gridDim.x < n_it ? it + gridDim.x : it);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[NG][2];
#pragma unroll
            for (int ng = 0; ng < NG; ++ng) acc[ng][0] = acc[ng][1] = zero4();
            const char* wp = reinterpret_cast<const char*>(wn16 + (size_t)(2 * wave) * 16 * 1024);   // (scalar base + lane offset: see phase 1)
            unsigned loff = (unsigned)lane * 16u;
            asm volatile("" : "+v"(loff));
            constexpr int D3 = AL_D3;
            lhalf8 wh[D3][2], wl[D3][2];   // [ring slot][nt]: see phase 1
#pragma unroll
            for (int u = 0; u < D3; ++u)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const unsigned fo = loff + (unsigned)(nt * 16 + u) * 2048u;
                    wh[u][nt] = *reinterpret_cast<const lhalf8*>(wp + fo);
                    if (!SINGLE) wl[u][nt] = *reinterpret_cast<const lhalf8*>(wp + (fo + 1024u));
                }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int sl = u % D3;
                __builtin_amdgcn_sched_barrier(0);
                lhalf8 bh[NG], bl[NG];
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) {
                    const unsigned char* f = s_q + (ng * 16 + m) * AL_QS + u * 128 + g * 16;
                    bh[ng] = *reinterpret_cast<const lhalf8*>(f);
                    if (!SINGLE) bl[ng] = *reinterpret_cast<const lhalf8*>(f + 64);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (!SINGLE) {
#pragma unroll
                        for (int ng = 0; ng < NG; ++ng)
                            acc[ng][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[sl][nt], bl[ng], acc[ng][nt], 0, 0, 0);
#pragma unroll
                        for (int ng = 0; ng < NG; ++ng)
                            acc[ng][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[sl][nt], bh[ng], acc[ng][nt], 0, 0, 0);
                    }
#pragma unroll
                    for (int ng = 0; ng < NG; ++ng)
                        acc[ng][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[sl][nt], bh[ng], acc[ng][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (u + D3 < 16) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const unsigned fo = loff + (unsigned)(nt * 16 + u + D3) * 2048u;
                        wh[sl][nt] = *reinterpret_cast<const lhalf8*>(wp + fo);
                        if (!SINGLE) wl[sl][nt] = *reinterpret_cast<const lhalf8*>(wp + (fo + 1024u));
                    }
                }
            }
#pragma unroll
            for (int ng = 0; ng < NG; ++ng) {
                const long grp = it * NG + ng;
                if (grp < groups) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        f32x4 v = acc[ng][nt] + sh[nt];
                        v += res[ng][nt];
                        st4(U + (grp * S3D_GROUP + m) * 128 + (2 * wave + nt) * 16 + 4 * g, v);
                    }
                }
            }
        }
        __syncthreads();   // phase 1 of the next iteration overwrites the regions phase 3 read
    }
}

static std::atomic<unsigned long long> g_al_attr{0};
int launch_attn_last_fused(const float* X, float* U, long groups, int T, const float* wm16, const float* bm,
                           const float* wn16, const float* bn, bool single_pass, hipStream_t stream) {
    if (groups <= 0) return 0;
    S3D_CHECK_ARG(T >= 1 && T <= S3D_N_TOKENS_MAX, "attn_last_fused: T %d", T);
    constexpr int NG = 2;
    const size_t lds = (size_t)NG * 16 * AL_QS;
    TRY_RET(s3d_set_max_lds(g_al_attr, {(const void*)attn_last_fused_kernel<false, NG>, (const void*)attn_last_fused_kernel<true, NG>}, lds));
    const long n_it = (groups + NG - 1) / NG;
    const long cap = 2L * s3d_cu_count();
    const unsigned grid = (unsigned)(n_it < cap ? n_it : cap);
    const _Float16* m16 = reinterpret_cast<const _Float16*>(wm16);
    const _Float16* n16 = reinterpret_cast<const _Float16*>(wn16);
    if (single_pass)
        hipLaunchKernelGGL((attn_last_fused_kernel<true, NG>), dim3(grid), dim3(256), lds, stream, X, U, groups, T, m16, bm, n16, bn);
    else
        hipLaunchKernelGGL((attn_last_fused_kernel<false, NG>), dim3(grid), dim3(256), lds, stream, X, U, groups, T, m16, bm, n16, bn);
    S3D_LAUNCH_CHECK();
    return 0;
}
