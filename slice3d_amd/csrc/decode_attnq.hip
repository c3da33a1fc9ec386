// decode_attnq.hip — self-attention block of one encoder layer, query-major tiling (split precision):
//     X <- LN1(X + out_proj(MHA(X)))        X: [group][T tokens][16 queries][128]
//
// The 13 tokens of ONE query attend only to each other, so a 16-row MFMA tile is made of the T tokens of a
// single query (rows T..15 are padding) instead of 16 queries of one token.  Every wave owns two queries of
// the 16-query group and does everything for them — QKV, the 13x13 attention, out_proj, residual, LayerNorm —
// without exchanging anything with the other waves:
//   Q^T, K^T  swapped form   D^T = W X^T   (f16x3 MFMA): lane (token, g) holds head dims {16j + 4g + i}
//   V         plain form     D   = X W^T   (same fragments, operands exchanged): lane (dim, g) holds tokens 4g+i
//   S^T = K Q^T   f16x3 MFMA 16x16x32: one k-step, k-slot 8g+t <-> dim 16(t>>2)+4g+(t&3) = the Q / K registers as they are
//   softmax over the keys = registers i and lane groups g of one query column (two lane swaps)
//   O^T = V^T P^T f16x3 MFMA 16x16x16: A = the V registers, B = the P registers (k-slot 4g+t <-> key 4g+t); result lane
//                 (token, g) holds dims 4g+i
//   out_proj      f16x3 MFMA: O^T registers are its B operand (k-slot 8g+t <-> dim 16(t>>2) + 4g + (t&3), folded
//                 into the packed W_o columns), accumulated over heads
// No Q/K/V/O ever touches LDS; LDS holds only weight fragments (LDS-DMA ring of quarter-head slots, see the kernel).  13 of 16 tile rows are useful (19 % padding);
// the token-0-pruned last layer keeps the token-major kernel (decode_f16.hip), where pruning skips whole tiles.
#include "attnq.h"

// TRAIN (the training step's forward of the block, models.py:83 in train mode): reads Xin, writes the LayerNorm output to X,
// the pre-LayerNorm sum u = x + dropout1(out_proj(attn) + b) to ta.U (the LayerNorm backward's input) and the attention
// output O (before out_proj; its weight gradient's operand) to ta.O — Q / K / V and the probabilities never leave the
// chip, the backward recomputes them (train_attnq.hip).  Dropout: counter-based masks (dropout.h) of site ta.d0 on the
// probabilities (index ((row * 4 + head) * 16 + key)) and ta.d1 on the block output (index row * 128 + channel) — the
// indices the stored-QKV kernels of train.hip use, so both paths draw the same masks.
struct AttnTrainArgs {
    const float* Xin;   // layer input rows (TRAIN; inference works in place on X)
    float* U;           // pre-LayerNorm rows
    float* O;           // attention output rows (128 = 4 heads x 32)
    DropCfg d0, d1;
};
template <bool SINGLE, bool TRAIN, bool BF = false>   // BF: S3D_PREC_BF16, single pass on the bf16 MFMA (inference); SINGLE: S3D_PREC_F16, one f16 MFMA per projection product (the 13x13 core stays on the fp32 MFMA)
__global__ __launch_bounds__(256, 2) void attn_layer_q_kernel(float* X, long groups, int T, const _Float16* wimg,
                                                               const LayerPtrs w, const AttnTrainArgs ta) {
    static_assert(!BF || (SINGLE && !TRAIN), "the bf16 mode is a single-pass inference mode");
    extern __shared__ __attribute__((aligned(16))) _Float16 s_win[];   // ring 4 x 16 KiB, then the small vectors
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: no waterfall around M0
    const int m = lane & 15, g = lane >> 4;
    const float scale = 0.17677669529663687f * 1.4426950408889634f;   // log2(e) / sqrt(32)
    const _Float16* g_in = wimg;
    const _Float16* g_out = wimg + 4 * AQ_WIN_HALFS;
    // in_proj bias | out_proj bias | LayerNorm1 gamma | beta: 768 floats behind the ring (published by the first barrier)
    float* s_par = reinterpret_cast<float*>(s_win + 4 * AQ3_SLOT_HALFS);
    for (int i = tid; i < 192; i += 256) {
        const float* src = i < 96 ? w.inb + 4 * i : i < 128 ? w.outb + 4 * (i - 96) : i < 160 ? w.ln1g + 4 * (i - 128) : w.ln1b + 4 * (i - 160);
        st4(s_par + 4 * i, ld4(src));
    }
    __syncthreads();   // the first item's accumulators read the out_proj bias before the first ring barrier
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_win + lane * 8);
    const unsigned lpar = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_par + 8 * g);
    const unsigned lpar4 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_par + 4 * g);
    const unsigned lparm = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_par + m);
    // phase ph = 4*h + {0 q, 1 k, 2 v, 3 out_proj}: 16 chunks of 1 KiB, four per wave.  A wave issues its four one at a
    // time BETWEEN the MFMA groups of the phase before (dma_piece(ph, buf, k) after the first half of k-step k): issued
    // back to back at the top of the phase, with the matrix pipe empty, they cost the wave ~60 issue cycles each
    auto dma_piece = [&](int ph, int buf, int k) {
        const int h = ph >> 2, part = ph & 3;
        const _Float16* src0 = part < 3 ? g_in + (size_t)h * AQ_WIN_HALFS + part * AQ3_SLOT_HALFS
                                        : g_out + (size_t)h * AQ_WO_HALFS;
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
    // Ring barrier of phase (h, part) with the exact count of younger vector-memory operations (AQ_BARRIER_N).  Per item a wave
    // issues: 16 DMA sets of 4 (set p during phase p - 3), after the last head 16 row loads + 16 output stores (TRAIN: + 16
    // pre-LayerNorm stores), TRAIN: 4 attention-output stores per head between the v phase and the out_proj barrier.
    //   inference  h > 0: 8 everywhere;  h = 0, parts q / k / v: 8 + 32 = 40 (the epilogue of the previous item is younger)
    //   TRAIN      q / k: 12, v: 8, out_proj: 12;  h = 0: q / k 12 + 48 = 60, v 8 + 48 = 56
    // The first item of a workgroup (prologue: three sets, then the row loads) and T < 9 (store halves without active lanes
    // are skipped) keep vmcnt(8).
    const bool wide = T >= 9;
    bool wide_h0 = false;   // set after the first item
    auto ring_barrier = [&](int h, int part) {
        if (!wide) {
            AQ_BARRIER();
        } else if (h == 0 && part < 3 && wide_h0) {
            if (!TRAIN) { AQ_BARRIER_N(40); }
            else if (part == 2) { AQ_BARRIER_N(56); }
            else { AQ_BARRIER_N(60); }
        } else if (TRAIN && part != 2) {
            AQ_BARRIER_N(12);
        } else {
            AQ_BARRIER();
        }
    };

    // raw fp32 rows of the NEXT item, requested in the epilogue of the current one (the first item's here)
    f32x4 xf[2][4][2];
    auto load_rows = [&](long it) {
        const float* Xn = (TRAIN ? ta.Xin : X) + (it >> 1) * T * S3D_GROUP * 128;
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
        float* Xg = X + grp * T * S3D_GROUP * 128;
        // out_proj accumulators open with residual row + out_proj bias (exact fp32 rows: tile j of the accumulator is
        // columns 32(j>>1) + 8g + 4(j&1) + i, i.e. xf[r][j>>1][j&1]), so the epilogue is LayerNorm only
        f32x4 acc_o[2][8];
        {   // hand-issued reads (a compiler-visible LDS read is guarded with vmcnt(0) against the ring's DMA in flight)
            f32x4 bo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) AQ_READ(bo[j], lpar, (384 + 32 * (j >> 1) + 4 * (j & 1)) * 4);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(bo[0]), "+v"(bo[1]), "+v"(bo[2]), "+v"(bo[3]), "+v"(bo[4]), "+v"(bo[5]), "+v"(bo[6]), "+v"(bo[7]));
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 2; ++r) acc_o[r][j] = TRAIN ? bo[j] : xf[r][j >> 1][j & 1] + bo[j];   // TRAIN: the residual joins after the output dropout
        }
        // rows of the item: both halves in registers for the whole item (the high halves used to be parked in LDS and re-read
        // with every k-step: 2 of 6 fragment reads)
        half8q xl[2][4], xh[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) split8x<BF>(xf[r][u][0], xf[r][u][1], xh[r][u], xl[r][u]);
        const bool more_items = item + gridDim.x < items;
#pragma unroll 1
        for (int h = 0; h < 4; ++h) {
            f32x4 qd[2][2], kd[2][2], vd[2][2];
            f32x4 bq[2], bk[2];
            float bv[2];
            // =============== phases q, k, v: one swapped (q, k) or plain (v) GEMM each ===============
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                if (part == 0) {   // the head's biases (LDS copy) open the accumulations below; read BEFORE the barrier, whose
                                   // LDS fence the compiler knows about (a later read would be waited for with lgkmcnt(0)
                                   // at the first MFMA, i.e. behind the hand-issued fragment reads of two k-steps)
                    const unsigned lq = lpar4 + (unsigned)h * 128u, lv = lparm + (unsigned)h * 128u;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        AQ_READ(bq[j], lq, 64 * j);
                        AQ_READ(bk[j], lq, 512 + 64 * j);
                        AQ_READ32(bv[j], lv, 1024 + 64 * j);
                    }
                }
                ring_barrier(h, part);   // this phase's fragments have landed; the other slot is free
                const int nph = (4 * h + part + 3) & 15, nbuf = (int)((ps + 3) & 3);
                const unsigned lwa = lds_ring + (unsigned)(ps & 3) * (AQ3_SLOT_HALFS * 2);
                f32x4 d[2][2], c0[2];   // c0: the bias opens both row tiles' accumulations (C operand of k-step 0, no copies)
                // fragment pairs (hi | lo) of the two 16-row tiles j and the high halves of the wave's two row tiles for
                // k-step U, all read one k-step ahead (6 reads in flight under the 12 MFMAs of the current step)
                half8q fh[2][2], fl[2][2];
#define AQ_STEP_READS(B, U)                                          \
    AQ_READ(fh[B][0], lwa, (U) * 2048);                              \
    AQ_READ(fl[B][0], lwa, (U) * 2048 + 1024);                       \
    AQ_READ(fh[B][1], lwa, (4 + (U)) * 2048);                        \
    AQ_READ(fl[B][1], lwa, (4 + (U)) * 2048 + 1024);
#define AQ_STEP_MFMA(B, U)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                          \
        if (j == 1) {                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                               \
            dma_piece(nph, nbuf, U);                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                    \
        if (part < 2) { /* D^T = W X^T */                                                                    \
            d[0][j] = mfma3q<SINGLE, BF>(fh[B][j], fl[B][j], xh[0][U], xl[0][U], (U) == 0 ? c0[j] : d[0][j]);    \
            d[1][j] = mfma3q<SINGLE, BF>(fh[B][j], fl[B][j], xh[1][U], xl[1][U], (U) == 0 ? c0[j] : d[1][j]);    \
        } else { /* D = X W^T */                                                                             \
            d[0][j] = mfma3q<SINGLE, BF>(xh[0][U], xl[0][U], fh[B][j], fl[B][j], (U) == 0 ? c0[j] : d[0][j]);    \
            d[1][j] = mfma3q<SINGLE, BF>(xh[1][U], xl[1][U], fh[B][j], fl[B][j], (U) == 0 ? c0[j] : d[1][j]);    \
        }                                                                                                    \
    }                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);
                AQ_STEP_READS(0, 0)
                AQ_STEP_READS(1, 1)
                AQ_WAIT4(4, fh[0][0], fl[0][0], fh[0][1], fl[0][1]);
                if (part == 0)   // the head's bias reads are older than the fragment reads: landed with this wait
                    asm volatile("" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bk[0]), "+v"(bk[1]), "+v"(bv[0]), "+v"(bv[1]));
#pragma unroll
                for (int j = 0; j < 2; ++j) c0[j] = part == 0 ? bq[j] : part == 1 ? bk[j] : f32x4{bv[j], bv[j], bv[j], bv[j]};
                AQ_STEP_MFMA(0, 0)
                AQ_STEP_READS(0, 2)
                AQ_WAIT4(4, fh[1][0], fl[1][0], fh[1][1], fl[1][1]);
                AQ_STEP_MFMA(1, 1)
                AQ_STEP_READS(1, 3)
                AQ_WAIT4(4, fh[0][0], fl[0][0], fh[0][1], fl[0][1]);
                AQ_STEP_MFMA(0, 2)
                AQ_WAIT4(0, fh[1][0], fl[1][0], fh[1][1], fl[1][1]);
                AQ_STEP_MFMA(1, 3)
#undef AQ_STEP_READS
#undef AQ_STEP_MFMA
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
            // =============== attention core, then phase out_proj ===============
            // The 13x13 core on the same split-precision f16 MFMA as the projections: the head's 32 dims are exactly one
            // k-step of v_mfma_f32_16x16x32_f16 (k-slot 8g + t <-> dim 16(t>>2) + 4g + (t&3): the registers as they
            // are), and P V runs on v_mfma_f32_16x16x16_f16 (k-slot 4g + t <-> key 4g + t).  9 MFMAs of 4 passes per
            // query instead of 16 fp32 MFMAs of 8.  Scores in log2 units (the scale carries log2 e), softmax by v_exp_f32.
            // Both queries of the wave go through each stage together (independent chains), and every group of splits
            // is followed by AQ_SETTLE before the MFMAs that read it: the halves are written by 16-bit partial-register
            // asm ops (v_fma_mixlo/hi_f16) whose write -> MFMA-read spacing the compiler does not pad (one wait state
            // measured as too few on gfx950: wrong P V products in some schedules, fixed by the padding alone).
#define AQ_SETTLE()                                 \
    __builtin_amdgcn_sched_barrier(0);              \
    asm volatile("s_nop 15" ::: "memory");          \
    __builtin_amdgcn_sched_barrier(0);
            half8q oh[2], ol[2];
            {
                half8q kh[2], kl[2], qh[2], ql[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    split8x<BF>(kd[r][0], kd[r][1], kh[r], kl[r]);
                    split8x<BF>(qd[r][0] * scale, qd[r][1] * scale, qh[r], ql[r]);
                }
                AQ_SETTLE()
                f32x4 s[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) s[r] = mfma3q<SINGLE, BF>(kh[r], kl[r], qh[r], ql[r], zero4());
                half4q ph[2], pl[2], vh[2][2], vl[2][2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    float e[4];
                    float mx = -1e30f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        e[i] = (4 * g + i < T) ? s[r][i] : -1e30f;
                        mx = fmaxf(mx, e[i]);
                    }
                    mx = colmax16(mx);
                    float den = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        e[i] = __builtin_amdgcn_exp2f(e[i] - mx);   // masked keys: exp2(-1e30) = 0
                        den += e[i];
                    }
                    const float inv = __builtin_amdgcn_rcpf(colsum16(den));   // v_rcp_f32, 1 ulp
                    f32x4 pr = f32x4{e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv};
                    if (TRAIN && ta.d0.p > 0.f) {   // lane (query token m, g): keys 4g .. 4g+3 of probability row (row, head)
                        float mk[4];
                        int mto = mt, go = g;   // opaque copies: otherwise the per-lane index parts are formed in the prologue,
                        asm volatile("" : "+v"(mto), "+v"(go));   // kept live through the item loop and spilled
                        const unsigned long long rowq = (unsigned long long)((grp * T + mto) * S3D_GROUP + q0 + r);
                        s3d_drop4(ta.d0, (rowq * 4 + (unsigned)h) * 16 + 4 * go, mk);
                        pr = f32x4{pr[0] * mk[0], pr[1] * mk[1], pr[2] * mk[2], pr[3] * mk[3]};
                    }
                    split4x<BF>(pr, ph[r], pl[r]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) split4x<BF>(vd[r][j], vh[r][j], vl[r][j]);
                }
                AQ_SETTLE()
                f32x4 od[2][2];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 2; ++j) od[r][j] = mfma3h<SINGLE, BF>(vh[r][j], vl[r][j], ph[r], pl[r], zero4());
#pragma unroll
                for (int r = 0; r < 2; ++r) split8x<BF>(od[r][0], od[r][1], oh[r], ol[r]);
                if (TRAIN) {   // O rows of this head: tiles j = 0, 1 are dims 4g + i and 16 + 4g + i of token m = one 128-byte
                               // line per token after the lane exchange (s3d_full_line_pair)
                    int mo = m, go = g;
                    asm volatile("" : "+v"(mo), "+v"(go));   // (see the dropout draw above)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        f32x4 va, vb;
                        s3d_full_line_pair(od[r][0], od[r][1], mo, va, vb);
                        float* oo = ta.O + ((grp * T + (mo & 7)) * S3D_GROUP + q0 + r) * 128 + 32 * h + 16 * (mo >> 3) + 4 * go;
                        if ((mo & 7) < T) st4(oo, va);
                        if ((mo & 7) + 8 < T) st4(oo + 8 * S3D_GROUP * 128, vb);
                    }
                }
            }
#undef AQ_SETTLE
            ring_barrier(h, 3);   // out_proj fragments have landed; the v slot is free
            // the next phase is the next head's q, or phase 0 of the next item (issued even after the last item: no
            // branch in the MFMA stream; the kernel drains vmcnt before it ends)
            const int oph = (4 * h + 3 + 3) & 15, obuf = (int)((ps + 3) & 3);
            {
                const unsigned lwa = lds_ring + (unsigned)(ps & 3) * (AQ3_SLOT_HALFS * 2);
                half8q wh[2][2], wl[2][2];   // [buffer][tile of the pair]
#define AQ_O_READS(B, G)                                             \
    AQ_READ(wh[B][0], lwa, (2 * (G)) * 2048);                        \
    AQ_READ(wl[B][0], lwa, (2 * (G)) * 2048 + 1024);                 \
    AQ_READ(wh[B][1], lwa, (2 * (G) + 1) * 2048);                    \
    AQ_READ(wl[B][1], lwa, (2 * (G) + 1) * 2048 + 1024);
#define AQ_O_MFMA(B, G)                                                                                      \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                          \
        if (q == 1) {                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                               \
            dma_piece(oph, obuf, G);                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                    \
        _Pragma("unroll") for (int r = 0; r < 2; ++r)                                                        \
            acc_o[r][2 * (G) + q] = mfma3q<SINGLE, BF>(wh[B][q], wl[B][q], oh[r], ol[r], acc_o[r][2 * (G) + q]); \
    }                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);
                AQ_O_READS(0, 0)
                AQ_O_READS(1, 1)
                AQ_WAIT4(4, wh[0][0], wl[0][0], wh[0][1], wl[0][1]);
                AQ_O_MFMA(0, 0)
                AQ_O_READS(0, 2)
                AQ_WAIT4(4, wh[1][0], wl[1][0], wh[1][1], wl[1][1]);
                AQ_O_MFMA(1, 1)
                AQ_O_READS(1, 3)
                AQ_WAIT4(4, wh[0][0], wl[0][0], wh[0][1], wl[0][1]);
                AQ_O_MFMA(0, 2)
                AQ_WAIT4(0, wh[1][0], wl[1][0], wh[1][1], wl[1][1]);
                AQ_O_MFMA(1, 3)
#undef AQ_O_READS
#undef AQ_O_MFMA
            }
            ++ps;
        }
        if (TRAIN) {   // u = x + dropout1(out_proj + bias): tile j, reg i <-> column 32(j>>1) + 8g + 4(j&1) + i of token m
            int mto = mt, go = g;
            asm volatile("" : "+v"(mto), "+v"(go));
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (ta.d1.p > 0.f) {
                        float mk[4];
                        const unsigned long long row = (unsigned long long)((grp * T + mto) * S3D_GROUP + q0 + r);
                        s3d_drop4(ta.d1, row * 128 + 32 * (j >> 1) + 8 * go + 4 * (j & 1), mk);
                        acc_o[r][j] = f32x4{acc_o[r][j][0] * mk[0], acc_o[r][j][1] * mk[1], acc_o[r][j][2] * mk[2], acc_o[r][j][3] * mk[3]};
                    }
                    // residual = f32(hi) + f32(lo) of the row's split (22 bits, as in the FFN kernel's epilogue): keeping the
                    // raw fp32 rows live through the four heads costs 64 registers this kernel does not have
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc_o[r][j][i] += (float)xh[r][j >> 1][4 * (j & 1) + i] + (float)xl[r][j >> 1][4 * (j & 1) + i];
                }
            // pre-LayerNorm rows, in the full-line form of the output stores below — BEFORE the next rows are requested: with
            // the exchange temporaries live next to the 64 prefetch registers and the 64 of gamma / beta, hipcc spilled six of
            // the freshly loaded row registers (a scratch store that has to wait for its load)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                int mo = m, go = g;
                asm volatile("" : "+v"(mo), "+v"(go));
                float* ua = ta.U + ((grp * T + (mo & 7)) * S3D_GROUP + q0 + r) * 128 + 8 * go + 4 * (mo >> 3);
#pragma unroll
                for (int J = 0; J < 4; ++J) {
                    f32x4 va, vb;
                    s3d_full_line_pair(acc_o[r][2 * J], acc_o[r][2 * J + 1], mo, va, vb);
                    if ((mo & 7) < T) st4(ua + 32 * J, va);
                    if ((mo & 7) + 8 < T) st4(ua + 8 * S3D_GROUP * 128 + 32 * J, vb);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next item's rows: requested here, where the q / k / v and fragment registers are free; the LayerNorm
        // below covers most of their latency
        load_rows(more_items ? item + gridDim.x : item);   // unconditional: a conditional load keeps the OLD rows live through the heads
        __builtin_amdgcn_sched_barrier(0);
        // ---- LayerNorm1 (the accumulators already hold out_proj + bias + residual), store.  f32x4 arithmetic: packed
        //      fp32 VALU ops.  gamma / beta are fetched by hand-issued reads, all 16 up front under the statistics of
        //      the first tile (compiler-issued ones are sunk next to each store and waited for one by one) ----
        f32x4 ga[8], be[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            AQ_READ(ga[j], lpar, (512 + 32 * (j >> 1) + 4 * (j & 1)) * 4);
            AQ_READ(be[j], lpar, (640 + 32 * (j >> 1) + 4 * (j & 1)) * 4);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x4 s4 = acc_o[r][0];
#pragma unroll
            for (int j = 1; j < 8; ++j) s4 += acc_o[r][j];
            const float mean = colsum16((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.f / 128.f);
            f32x4 v4 = zero4();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc_o[r][j] -= mean;
                v4 += acc_o[r][j] * acc_o[r][j];
            }
            const float rstd = __builtin_amdgcn_rsqf(colsum16((v4[0] + v4[1]) + (v4[2] + v4[3])) * (1.f / 128.f) + 1e-5f);
            if (r == 0)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(ga[0]), "+v"(ga[1]), "+v"(ga[2]), "+v"(ga[3]), "+v"(ga[4]), "+v"(ga[5]), "+v"(ga[6]), "+v"(ga[7]),
                               "+v"(be[0]), "+v"(be[1]), "+v"(be[2]), "+v"(be[3]), "+v"(be[4]), "+v"(be[5]), "+v"(be[6]), "+v"(be[7]));
            // Full-line stores (s3d_full_line_pair): a lane's tiles j = 2J, 2J + 1 are columns 32J + 8g + {0..3} and {4..7} of
            // token m, so a plain store writes 16-byte pieces at a 32-byte stride.  After the exchange with lane m ^ 8
            // one instruction writes tokens 0-7, the next tokens 8-15, each token's 128-byte line whole (lanes m < 8 the
            // low, lanes m >= 8 the high 16 bytes of every 32).
            float* oa = Xg + ((m & 7) * S3D_GROUP + q0 + r) * 128 + 8 * g + 4 * (m >> 3);
            const bool ok_b = (m & 7) + 8 < T;   // tokens 0-7 always exist (T >= 8 is not required: see ok_a)
            const bool ok_a = (m & 7) < T;
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                const f32x4 r0 = acc_o[r][2 * J] * (ga[2 * J] * rstd) + be[2 * J];
                const f32x4 r1 = acc_o[r][2 * J + 1] * (ga[2 * J + 1] * rstd) + be[2 * J + 1];
                f32x4 va, vb;
                s3d_full_line_pair(r0, r1, m, va, vb);
                if (ok_a) st4(oa + 32 * J, va);
                if (ok_b) st4(oa + 8 * S3D_GROUP * 128 + 32 * J, vb);
            }
            __builtin_amdgcn_sched_barrier(0);   // one row tile at a time
        }
        wide_h0 = wide;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last phase-0 prefetch must not outlive the workgroup's LDS
}

static int launch_attn_q_any(float* X, long groups, int T, const LayerPtrs& w, hipStream_t stream, bool single_pass,
                             const AttnTrainArgs* ta, bool bf16 = false) {
    if (groups <= 0) return 0;
    S3D_CHECK_ARG(T >= 2 && T <= 16 && w.aq16 != nullptr, "attn_q: T %d", T);
    const size_t lds = (size_t)(4 * AQ3_SLOT_HALFS) * 2 + 768 * 4;   // 64 KiB ring + the small vectors
    static std::atomic<unsigned long long> attr_done{0};
    TRY_RET(s3d_set_max_lds(attr_done, {(const void*)attn_layer_q_kernel<false, false>, (const void*)attn_layer_q_kernel<true, false>,
                                        (const void*)attn_layer_q_kernel<false, true>, (const void*)attn_layer_q_kernel<true, true>,
                                        (const void*)attn_layer_q_kernel<true, false, true>}, lds));
    const long blocks = 2 * groups < 4096 ? 2 * groups : 4096;
    const _Float16* img = reinterpret_cast<const _Float16*>(bf16 ? w.aqb16 : w.aq16);
    S3D_CHECK_ARG(!bf16 || (single_pass && !ta && w.aqb16), "attn_q: the bf16 mode is a single-pass inference mode with its own image");
    const AttnTrainArgs none = {};
    if (bf16)
        hipLaunchKernelGGL((attn_layer_q_kernel<true, false, true>), dim3((unsigned)blocks), dim3(256), lds, stream, X, groups, T, img, w, none);
    else if (ta && single_pass)   // the training step's single-pass f16 throughput mode (round 6)
        hipLaunchKernelGGL((attn_layer_q_kernel<true, true>), dim3((unsigned)blocks), dim3(256), lds, stream, X, groups, T, img, w, *ta);
    else if (ta)
        hipLaunchKernelGGL((attn_layer_q_kernel<false, true>), dim3((unsigned)blocks), dim3(256), lds, stream, X, groups, T, img, w, *ta);
    else if (single_pass)
        hipLaunchKernelGGL((attn_layer_q_kernel<true, false>), dim3((unsigned)blocks), dim3(256), lds, stream, X, groups, T, img, w, none);
    else
        hipLaunchKernelGGL((attn_layer_q_kernel<false, false>), dim3((unsigned)blocks), dim3(256), lds, stream, X, groups, T, img, w, none);
    S3D_LAUNCH_CHECK();
    return 0;
}
int launch_attn_layer_q(float* X, long groups, int T, const LayerPtrs& w, hipStream_t stream, bool single_pass, bool bf16) {
    return launch_attn_q_any(X, groups, T, w, stream, single_pass, nullptr, bf16);
}
// training forward of the block (see the kernel): xin -> y = LN1(u), u = xin + dropout1(out_proj(MHA(xin))), o = MHA output
int launch_attn_layer_q_train(const float* xin, float* y, float* u, float* o, long groups, int T, const LayerPtrs& w,
                              const DropCfg& d0, const DropCfg& d1, hipStream_t stream, bool single) {
    S3D_CHECK_ARG(xin && y && u && o, "attn_q train: null buffer");
    AttnTrainArgs ta = {xin, u, o, d0, d1};
    return launch_attn_q_any(y, groups, T, w, stream, single, &ta);
}

// in_proj (384,128) / out_proj (128,128) -> fragment pairs (hi 512 halfs | lo 512 halfs each) of the query-major kernel
//   frag = h*24 + (p*2 + j)*4 + u : row = p*128 + 32h + 16j + (l&15), k = 32u + 8g + t          (p = q,k,v)
//   frag = 96 + h*8 + jc          : row m <-> output channel 32(jc>>1) + 8(m>>2) + 4(jc&1) + (m&3),
//                                   k-slot 8g + t <-> head dim 16(t>>2) + 4g + (t&3)
__global__ void pack_attn_q_f16x3_kernel(const float* __restrict__ win, const float* __restrict__ wout,
                                         _Float16* __restrict__ out, int bf16) {
    const int total = (96 + 32) * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, frag = idx >> 6;
        const int r = lane & 15, g = lane >> 4;
        float v[8];
        if (frag < 96) {
            const int h = frag / 24, f = frag % 24, u = f & 3, j = (f >> 2) & 1, p = f >> 3;
            const float* src = win + (size_t)(p * 128 + 32 * h + 16 * j + r) * 128 + 32 * u + 8 * g;
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = src[t];
        } else {
            const int f = frag - 96, jc = f & 7, h = f >> 3;
            const int n = 32 * (jc >> 1) + 8 * (r >> 2) + 4 * (jc & 1) + (r & 3);
            const float* src = wout + (size_t)n * 128 + 32 * h;
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = src[16 * (t >> 2) + 4 * g + (t & 3)];
        }
        _Float16* dst = out + (size_t)frag * 1024 + lane * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (bf16) {   // bf16 bit patterns (S3D_PREC_BF16), no low halves
                dst[t] = __builtin_bit_cast(_Float16, (unsigned short)(bf16_pair_q(v[t], 0.f) & 0xFFFFu));
                dst[512 + t] = (_Float16)0.f;
                continue;
            }
            const _Float16 hh = (_Float16)v[t];
            dst[t] = hh;
            dst[512 + t] = (_Float16)(v[t] - (float)hh);
        }
    }
}

int launch_pack_attn_q_f16x3(const float* win, const float* wout, float* out, hipStream_t stream, int bf16) {
    hipLaunchKernelGGL(pack_attn_q_f16x3_kernel, dim3(32), dim3(256), 0, stream, win, wout,
                       reinterpret_cast<_Float16*>(out), bf16);
    S3D_LAUNCH_CHECK();
    return 0;
}
