// train_attnq.hip — backward of the attention core of one encoder layer, query-major, with Q / K / V RECOMPUTED on chip
// (reference: nn.MultiheadAttention inside nn.TransformerEncoderLayer, models.py:18-19,83, train mode).
//
//     in :  X    [group][T tokens][16 queries][128]   the layer's input rows (what the forward projected)
//           dO   same shape                           gradient w.r.t. the attention output (before out_proj)
//     out:  dQKV [group][T][16][384]                  gradient w.r.t. the in_proj output (q | k | v, head h at 32h)
//
// The forward (decode_attnq.hip, TRAIN) keeps no Q / K / V and no probabilities: this kernel repeats its projections for
// the two queries a wave owns — the same weight-fragment image, the same LDS-DMA ring of quarter-head slots, here with
// three phases per head (q | k | v, all in the swapped form D^T = W X^T: lane (token, g) holds head dims 16j + 4g + i) —
// and runs the 13 x 13 core backward in registers, on the split-precision f16 MFMA like everything else:
//   S^T  = K Q^T            16x16x32, operands = the K / Q registers as they are            -> P (softmax over keys: registers
//                                                                                              i and lane groups g of a column)
//   dPd^T = V dO^T          16x16x32, same operand form (V: swapped form here, dO rows loaded straight into it)
//   dS   = P o (dP - sum_k P dP),  dP = dPd o mask,  Pd = P o mask          (mask = the forward's counter-based dropout draw)
//   dQ^T = K_plain  dS^T    16x16x16: A = K with lane (dim, g) holding keys 4g+i, B = the dS^T registers as they are
//   dK^T = Q_plain  dS      16x16x16: B = dS with lane (key, g) holding queries-tokens 4g+i
//   dV^T = dO_plain Pd      16x16x16
// The "plain" forms (lane = head dim, registers = tokens) and the token-transposed dS / Pd are not recomputed: a D-layout
// register tile used as the A operand IS its transpose, so one 16x16x16 MFMA against an identity B operand transposes a
// tile — on the f16 hi and lo halves separately, whose products with 1.0 are exact (fp32 result = the f16 value, converted
// back without rounding).  Two MFMAs of 4 passes per tile instead of another 24-MFMA projection.
// Results leave in the D layout lane (token, g) x 4 consecutive dims: a tile pair (j = 0, 1) is one 128-byte line of a
// token's dQ / dK / dV head slice after the lane exchange of s3d_full_line_pair.
// Padding rows (tokens T..15 of a tile): as keys they carry P = 0 exactly (masked before the softmax); as query tokens
// their dS / Pd COLUMNS are zeroed before the contractions over the query token.
#include "attnq.h"
#include "train.h"

struct AttnBwdArgs {
    const float* X;
    const float* dO;
    float* dQKV;
    DropCfg d0;
};

typedef _Float16 half2b __attribute__((ext_vector_type(2)));
typedef float float2b __attribute__((ext_vector_type(2)));

// low / high four halfs of an 8-half operand (no instructions: register sub-ranges)
__device__ __forceinline__ half4q lo4(const half8q v) { return __builtin_shufflevector(v, v, 0, 1, 2, 3); }
__device__ __forceinline__ half4q hi4(const half8q v) { return __builtin_shufflevector(v, v, 4, 5, 6, 7); }
// transposed tile of f16 values: D = A * I on the 16-deep MFMA, back to halfs (exact)
__device__ __forceinline__ half4q transpose16(const half4q a, const half4q ident) {
    const f32x4 t = __builtin_amdgcn_mfma_f32_16x16x16f16(a, ident, zero4(), 0, 0, 0);
    const half2b p0 = __builtin_convertvector(float2b{t[0], t[1]}, half2b);
    const half2b p1 = __builtin_convertvector(float2b{t[2], t[3]}, half2b);
    return __builtin_shufflevector(p0, p1, 0, 1, 2, 3);
}

#define AQB_SETTLE()                                \
    __builtin_amdgcn_sched_barrier(0);              \
    asm volatile("s_nop 15" ::: "memory");          \
    __builtin_amdgcn_sched_barrier(0);

// SINGLE: S3D_PREC_F16 training throughput mode (round 6): one f16 MFMA per product of the Q / K / V recomputation (the kernel's
// projection phases, 90 % of its MFMAs).  The 13 x 13 core backward keeps all three products in every mode: dS = P (dP - sum P dP)
// cancels to a small difference of similar numbers (the token rows of one query are alike), and with single-pass dP the step's U-Net
// gradients came out 9x their norm off (profiles/r06_train_f16.md) — with the core exact, 4e-3.
template <bool SINGLE>
__global__ __launch_bounds__(256, 2) void attn_bwd_q_kernel(const AttnBwdArgs a, long groups, int T, const _Float16* wimg,
                                                             const LayerPtrs w) {
    extern __shared__ __attribute__((aligned(16))) _Float16 s_win[];   // ring 4 x 16 KiB, then the in_proj bias
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g = lane >> 4;
    const float scale0 = 0.17677669529663687f;                     // 1 / sqrt(32)
    const float scale = scale0 * 1.4426950408889634f;              // scores in log2 units, as in the forward
    float* s_par = reinterpret_cast<float*>(s_win + 4 * AQ3_SLOT_HALFS);
    for (int i = tid; i < 96; i += 256) st4(s_par + 4 * i, ld4(w.inb + 4 * i));
    __syncthreads();
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_win + lane * 8);
    const unsigned lpar4 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_par + 4 * g);
    // phase ph = 3 h + {0 q, 1 k, 2 v}: the q | k | v slots of the forward's image (its out_proj slots are skipped)
    auto dma_piece = [&](int ph, int buf, int k) {
        const int h = ph / 3, part = ph - 3 * h;
        const _Float16* src0 = wimg + (size_t)h * AQ_WIN_HALFS + part * AQ3_SLOT_HALFS;
        const int i = wave + 4 * k;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + i * 512 + lane * 8),
                                         (__attribute__((address_space(3))) void*)(s_win + buf * AQ3_SLOT_HALFS + i * 512),
                                         16, 0, 0);
    };
    auto dma_phase = [&](int ph, int buf) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dma_piece(ph, buf, k);
    };
    const long items = 2 * groups;
    long ps = 0;   // running phase count: slot = ps & 3
    if ((long)blockIdx.x < items) {
        dma_phase(0, 0);
        dma_phase(1, 1);
        dma_phase(2, 2);
    }
    const bool row_ok = m < T;
    const int mt = row_ok ? m : T - 1;
    // Ring barrier with the exact count of younger vector-memory operations (AQ_BARRIER_N, attnq.h).  Per head a wave issues
    // 4 dO loads (before the q barrier), 3 DMA sets of 4, 12 result stores after the core; per item 16 row loads at the end.
    // Set p is issued during phase p - 3 (the same part of the previous head), so every barrier has 3 x 4 + 12 - 4 + 4 = 24
    // younger operations, 24 + 16 at the first head of an item that is not the workgroup's first (there: 8 + 16 + 4 = 28).
    // T < 9: store halves without active lanes are skipped -> vmcnt(8).
    const bool wide = T >= 9;
    bool wide_h0 = false;
    auto ring_barrier = [&](int h) {
        if (!wide) { AQ_BARRIER(); }
        else if (h == 0 && wide_h0) { AQ_BARRIER_N(40); }
        else { AQ_BARRIER_N(24); }
    };
    // identity B operand of the transposing MFMA: lane (n, g), k-slot 4g + t
    half4q ident;
#pragma unroll
    for (int t = 0; t < 4; ++t) ident[t] = (4 * g + t == m) ? (_Float16)1.f : (_Float16)0.f;

    f32x4 xf[2][4][2];
    auto load_rows = [&](long it) {
        const float* Xn = a.X + (it >> 1) * T * S3D_GROUP * 128;
        const int qn = 8 * (int)(it & 1) + 2 * wave;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float* p = Xn + (mt * S3D_GROUP + qn + r) * 128 + 8 * g;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xf[r][u][0] = ld4(p + 32 * u);
                xf[r][u][1] = ld4(p + 32 * u + 4);
            }
        }
    };
    if ((long)blockIdx.x < items) load_rows(blockIdx.x);

    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const long grp = item >> 1;
        const int q0 = 8 * (int)(item & 1) + 2 * wave;
        half8q xl[2][4], xh[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) split8pk(xf[r][u][0], xf[r][u][1], xh[r][u], xl[r][u]);
        const bool more_items = item + gridDim.x < items;
#pragma unroll 1
        for (int h = 0; h < 4; ++h) {
            // dO rows of this head, straight into the swapped-form layout: lane (token, g), dims 16 j + 4 g + i.  Requested
            // here, consumed after the three projection phases.
            f32x4 dod[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float* p = a.dO + ((grp * T + mt) * S3D_GROUP + q0 + r) * 128 + 32 * h + 4 * g;
                dod[r][0] = ld4(p);
                dod[r][1] = ld4(p + 16);
            }
            f32x4 qd[2][2], kd[2][2], vd[2][2];
            f32x4 bq[2], bk[2], bv[2];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                if (part == 0) {
                    const unsigned lq = lpar4 + (unsigned)h * 128u;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        AQ_READ(bq[j], lq, 64 * j);
                        AQ_READ(bk[j], lq, 512 + 64 * j);
                        AQ_READ(bv[j], lq, 1024 + 64 * j);
                    }
                }
                ring_barrier(h);   // this phase's fragments have landed; the slot of three phases ahead is free
                int nph = 3 * h + part + 3;
                if (nph >= 12) nph -= 12;
                const int nbuf = (int)((ps + 3) & 3);
                const unsigned lwa = lds_ring + (unsigned)(ps & 3) * (AQ3_SLOT_HALFS * 2);
                f32x4 d[2][2], c0[2];
                half8q fh[2][2], fl[2][2];
#define AQB_STEP_READS(B, U)                                         \
    AQ_READ(fh[B][0], lwa, (U) * 2048);                              \
    AQ_READ(fl[B][0], lwa, (U) * 2048 + 1024);                       \
    AQ_READ(fh[B][1], lwa, (4 + (U)) * 2048);                        \
    AQ_READ(fl[B][1], lwa, (4 + (U)) * 2048 + 1024);
#define AQB_STEP_MFMA(B, U)                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                          \
        if (j == 1) {                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                               \
            dma_piece(nph, nbuf, U);                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                    \
        d[0][j] = mfma3q<SINGLE>(fh[B][j], fl[B][j], xh[0][U], xl[0][U], (U) == 0 ? c0[j] : d[0][j]);         \
        d[1][j] = mfma3q<SINGLE>(fh[B][j], fl[B][j], xh[1][U], xl[1][U], (U) == 0 ? c0[j] : d[1][j]);         \
    }                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);
                AQB_STEP_READS(0, 0)
                AQB_STEP_READS(1, 1)
                AQ_WAIT4(4, fh[0][0], fl[0][0], fh[0][1], fl[0][1]);
                if (part == 0)   // the bias reads are older than the fragment reads: landed with this wait
                    asm volatile("" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bk[0]), "+v"(bk[1]), "+v"(bv[0]), "+v"(bv[1]));
#pragma unroll
                for (int j = 0; j < 2; ++j) c0[j] = part == 0 ? bq[j] : part == 1 ? bk[j] : bv[j];
                AQB_STEP_MFMA(0, 0)
                AQB_STEP_READS(0, 2)
                AQ_WAIT4(4, fh[1][0], fl[1][0], fh[1][1], fl[1][1]);
                AQB_STEP_MFMA(1, 1)
                AQB_STEP_READS(1, 3)
                AQ_WAIT4(4, fh[0][0], fl[0][0], fh[0][1], fl[0][1]);
                AQB_STEP_MFMA(0, 2)
                AQ_WAIT4(0, fh[1][0], fl[1][0], fh[1][1], fl[1][1]);
                AQB_STEP_MFMA(1, 3)
#undef AQB_STEP_READS
#undef AQB_STEP_MFMA
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (part == 0) qd[r][j] = d[r][j];
                        if (part == 1) kd[r][j] = d[r][j];
                        if (part == 2) vd[r][j] = d[r][j];
                    }
                ++ps;
            }
            // =============== core backward, one query at a time ===============
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                half8q kh, kl, qsh, qsl, vh, vl, doh, dol;
                half4q quh[2], qul[2];   // unscaled q tiles (the operand of dK)
                split8pk(kd[r][0], kd[r][1], kh, kl);
                split8pk(qd[r][0] * scale, qd[r][1] * scale, qsh, qsl);
                split8pk(vd[r][0], vd[r][1], vh, vl);
                split8pk(dod[r][0], dod[r][1], doh, dol);
                split4pk(qd[r][0], quh[0], qul[0]);
                split4pk(qd[r][1], quh[1], qul[1]);
                AQB_SETTLE()
                const f32x4 s = mfma3q<false>(kh, kl, qsh, qsl, zero4());      // S^T[key 4g+i][token m], log2 units
                const f32x4 dp = mfma3q<false>(vh, vl, doh, dol, zero4());     // dPd^T[key][token]
                // plain forms of K, Q, dO (lane = head dim, registers = tokens 4g+i): tile transposes of the halves
                half4q kph[2], kpl[2], qph[2], qpl[2], dph[2], dpl[2];
                kph[0] = transpose16(lo4(kh), ident); kph[1] = transpose16(hi4(kh), ident);
                kpl[0] = transpose16(lo4(kl), ident); kpl[1] = transpose16(hi4(kl), ident);
                qph[0] = transpose16(quh[0], ident); qph[1] = transpose16(quh[1], ident);
                qpl[0] = transpose16(qul[0], ident); qpl[1] = transpose16(qul[1], ident);
                dph[0] = transpose16(lo4(doh), ident); dph[1] = transpose16(hi4(doh), ident);
                dpl[0] = transpose16(lo4(dol), ident); dpl[1] = transpose16(hi4(dol), ident);
                // softmax over the keys of column m (as the forward)
                float e[4];
                float mx = -1e30f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e[i] = (4 * g + i < T) ? s[i] : -1e30f;
                    mx = fmaxf(mx, e[i]);
                }
                mx = colmax16(mx);
                float den = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e[i] = __builtin_amdgcn_exp2f(e[i] - mx);
                    den += e[i];
                }
                const float inv = __builtin_amdgcn_rcpf(colsum16(den));
                float mk[4] = {1.f, 1.f, 1.f, 1.f};
                if (a.d0.p > 0.f) {
                    const unsigned long long rowq = (unsigned long long)((grp * T + mt) * S3D_GROUP + q0 + r);
                    s3d_drop4(a.d0, (rowq * 4 + (unsigned)h) * 16 + 4 * g, mk);
                }
                float p[4], dpm[4], dot = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    p[i] = e[i] * inv;
                    dpm[i] = dp[i] * mk[i];
                    dot += p[i] * dpm[i];
                }
                dot = colsum16(dot);
                const float colf = row_ok ? 1.f : 0.f;   // a padding query token contributes nothing to dK / dV
                f32x4 ds, pd;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ds[i] = p[i] * (dpm[i] - dot) * (scale0 * colf);
                    pd[i] = p[i] * mk[i] * colf;
                }
                half4q dsh, dsl, pdh, pdl;
                split4pk(ds, dsh, dsl);
                split4pk(pd, pdh, pdl);
                AQB_SETTLE()
                // token-transposed dS and Pd: lane (key, g), registers = query tokens 4g+i
                const half4q dth = transpose16(dsh, ident), dtl = transpose16(dsl, ident);
                const half4q pth = transpose16(pdh, ident), ptl = transpose16(pdl, ident);
                f32x4 dq[2], dk[2], dv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    dq[j] = mfma3h<false>(kph[j], kpl[j], dsh, dsl, zero4());   // dQ^T[dim][token]
                    dk[j] = mfma3h<false>(qph[j], qpl[j], dth, dtl, zero4());   // dK^T[dim][key]
                    dv[j] = mfma3h<false>(dph[j], dpl[j], pth, ptl, zero4());   // dV^T[dim][key]
                }
                // rows leave as full 128-byte lines: lane m < 8 carries dims 0-15, lane m >= 8 dims 16-31 of token m & 7 (+ 8)
                float* ob = a.dQKV + ((grp * T + (m & 7)) * S3D_GROUP + q0 + r) * 384 + 32 * h + 16 * (m >> 3) + 4 * g;
                const bool ok_a = (m & 7) < T, ok_b = (m & 7) + 8 < T;
                f32x4 va, vb;
                s3d_full_line_pair(dq[0], dq[1], m, va, vb);
                if (ok_a) st4(ob, va);
                if (ok_b) st4(ob + 8 * S3D_GROUP * 384, vb);
                s3d_full_line_pair(dk[0], dk[1], m, va, vb);
                if (ok_a) st4(ob + 128, va);
                if (ok_b) st4(ob + 128 + 8 * S3D_GROUP * 384, vb);
                s3d_full_line_pair(dv[0], dv[1], m, va, vb);
                if (ok_a) st4(ob + 256, va);
                if (ok_b) st4(ob + 256 + 8 * S3D_GROUP * 384, vb);
                __builtin_amdgcn_sched_barrier(0);   // one query at a time
            }
        }
        load_rows(more_items ? item + gridDim.x : item);   // unconditional (see decode_attnq.hip)
        __builtin_amdgcn_sched_barrier(0);
        wide_h0 = wide;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last prefetches must not outlive the workgroup's LDS
}

// x (rows x 128), dO (rows x 128) -> dQKV (rows x 384); w.aq16 = the forward's fragment image, w.inb the in_proj bias
int launch_attn_bwd_q(const float* x, const float* d_o, float* dqkv, long groups, int T, const LayerPtrs& w,
                      const DropCfg& d0, hipStream_t stream, bool single) {
    if (groups <= 0) return 0;
    S3D_CHECK_ARG(T >= 2 && T <= 16 && w.aq16 != nullptr && x && d_o && dqkv, "attn_bwd_q: T %d", T);
    const size_t lds = (size_t)(4 * AQ3_SLOT_HALFS) * 2 + 384 * 4;
    static std::atomic<unsigned long long> attr_done{0};
    TRY_RET(s3d_set_max_lds(attr_done, {(const void*)attn_bwd_q_kernel<false>, (const void*)attn_bwd_q_kernel<true>}, lds));
    const long blocks = 2 * groups < 4096 ? 2 * groups : 4096;
    AttnBwdArgs a = {x, d_o, dqkv, d0};
    if (single)
        hipLaunchKernelGGL(attn_bwd_q_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, stream, a, groups, T,
                           reinterpret_cast<const _Float16*>(w.aq16), w);
    else
        hipLaunchKernelGGL(attn_bwd_q_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, stream, a, groups, T,
                           reinterpret_cast<const _Float16*>(w.aq16), w);
    S3D_LAUNCH_CHECK();
    return 0;
}
