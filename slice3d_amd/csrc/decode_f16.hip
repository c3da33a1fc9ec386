// decode_f16.hip — split-precision ("f16x3") variant of the decoder FFN kernel.
//
// Every fp32 operand x is split as x = hi + lo with hi = f16(x), lo = f16(x - hi)  (22 significant bits),
// and each product is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with fp32
// accumulation: products of two f16 values are exact in fp32, the dropped lo*lo term is < 2^-22 relative,
// so the result is fp32-class (measured against the fp32 path in tests) at 3 MFMAs of 16 cycles per
// 16x16x32 block instead of 8 MFMAs of 32 cycles — 5.3x fewer matrix-pipe cycles than the f32 kernel.
//
// Structure is that of ffn_layer_kernel (decode.hip): 128 rows per 4-wave workgroup, activations live in
// registers as B fragments (now f16 hi/lo pairs), W1/W2 stream through LDS in 32-hidden-unit chunks,
// the hidden tile never leaves registers.  For K=32 MFMAs a lane (m = l&15, g = l>>4) owns the 8
// consecutive channels {32u + 8g .. +7}; the output-channel permutation of W2's rows is chosen so that
// GEMM2's D registers land on exactly those channels again (residual, LayerNorm, 32-byte stores).
#include <stdlib.h>

#include "decode.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float (&x)[8], half8& hi, half8& lo) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const _Float16 h = (_Float16)x[t];
        hi[t] = h;
        lo[t] = (_Float16)(x[t] - (float)h);
    }
}
__device__ __forceinline__ f32x4 mfma3(const half8 ah, const half8 al, const half8 bh, const half8 bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ half8 ldh8(const _Float16* p) { return *reinterpret_cast<const half8*>(p); }
// ---------------------------------------------------------------------------------------------
// BF (S3D_PREC_BF16, round 5): the single-pass mode on the bf16 MFMA — what BASELINE configs[1] literally names.  The 16-bit
// lanes of the operand containers (half8 / half2v) then hold bf16 bit patterns: weights from the bf16 image
// (launch_pack_ffn_f16x3(..., bf16 = 1)), activations rounded by v_cvt_pk_bf16_f32.  Same rate as the f16 instruction, 8
// significand bits instead of 11: a throughput mode further from fp32 than S3D_PREC_F16, never the headline.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <bool BF>
__device__ __forceinline__ f32x4 mfma_hh(const half8 a, const half8 b, const f32x4 c) {
    if (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, a), __builtin_bit_cast(bf8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ unsigned bf16_pair(float a, float b) {   // bf16(a) | bf16(b) << 16, round to nearest even
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf2_t));
}
__device__ __forceinline__ float bf16_lane(const half8 v, int t) {   // the fp32 value of bf16 lane t
    return __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, v[t]) << 16);
}
// rows in BF mode: hi = bf16(x) (the MFMA operand), lo = f16(x - hi) (only the epilogue's residual reads it)
__device__ __forceinline__ void split8_bf(const float (&x)[8], half8& hi, half8& lo) {
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    const u4_t h = {bf16_pair(x[0], x[1]), bf16_pair(x[2], x[3]), bf16_pair(x[4], x[5]), bf16_pair(x[6], x[7])};
    hi = __builtin_bit_cast(half8, h);
#pragma unroll
    for (int t = 0; t < 8; ++t) lo[t] = (_Float16)(x[t] - bf16_lane(hi, t));
}
// global -> LDS copy of one 32 KiB weight chunk by LDS-DMA (1 KiB per wave-instruction, no VGPR staging)
__device__ __forceinline__ void dma_chunk32k(const _Float16* gsrc, _Float16* ldst, int wave, int lane, int nwaves) {
    for (int piece = wave; piece < 32; piece += nwaves)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + piece * 512 + lane * 8),
                                         (__attribute__((address_space(3))) void*)(ldst + piece * 512), 16, 0, 0);
}
__device__ __forceinline__ float quad_sum16(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

#define F16_CHUNK_HALFS 16384   // 32 KiB: W1 hi | W1 lo | W2 hi | W2 lo, 4096 halfs each

// ---------------------------------------------------------------------------------------------
// Software-pipelined FFN.  MODE 0: layer (inference), 1: last layer + fc_out, 2 / 3: TRAINING forward with / without
// dropout (activity bits of the hidden units for the backward, pre-LayerNorm output saved), 4: BACKWARD data path
//     dA = (dY W2) * bit,   dX = gate_scale * dA W1 + Dres
// on the transposed weight image (W2^T chunks GEMM-1-shaped, W1^T chunks GEMM-2-shaped: launch_pack_ffn_f16x3_bwd), the
// same loop with the gate in place of bias + ReLU and no LayerNorm.  128 rows per 4-wave workgroup, two workgroups per
// CU, activations in registers as f16 hi/lo B fragments, W1/W2 stream through LDS in 32-hidden-unit chunks, the hidden
// tile never leaves registers.  For K=32 MFMAs a lane (m = l&15, g = l>>4) owns the 8 consecutive channels
// {32u + 8g .. +7}; the output-channel permutation of W2's rows is chosen so that GEMM2's D registers land on exactly
// those channels again (residual, LayerNorm, 32-byte stores).  GEMM1 runs ONE CHUNK AHEAD of GEMM2:
//     phase A (iteration c):  hn = W1(c+1) x^T + b1(c+1)     [48 MFMAs]   ||   relu + hi/lo split of h(c)  [VALU]
//     phase B              :  acc += W2(c) h(c)              [48 MFMAs]   ||   fragment reads, next chunk's LDS-DMA
// Without the pipelining the ~80 VALU instructions of the split (more with dropout) sit between GEMM1(c) and GEMM2(c),
// which both depend on them: the wave's matrix pipe idles for the whole block and only the SIMD's other wave can fill it
// (MFMA pipe 52-63 % busy, 27 cycles per MFMA against 17 — the round-1 kernel).  Here every MFMA phase has independent
// VALU / LDS work to issue beside it.
// The LDS buffer of iteration c therefore holds W1(c+1) | W2(c): the two 16 KiB halves of a buffer are fetched from
// different chunks of the (unchanged) image.  lin1's bias is the accumulator's initial value.  Fragment reads are
// issued one 12-MFMA group ahead; sched_group_barrier pins the interleave.
// ---------------------------------------------------------------------------------------------
#ifndef PIPE_WAVES
#define PIPE_WAVES 4   // waves per workgroup: 4 = two workgroups per CU, each streaming the weights; 8 = one workgroup and one
                       // weight stream per CU (half the L2 -> LDS traffic): measured identical, 6.15 vs 6.16 ms (tools/ffn_w8.sh)
#endif
#define PIPE_THREADS (PIPE_WAVES * 64)
#ifndef PIPE_R
#define PIPE_R 2     // 16-row tiles per wave: 2 = two workgroups per CU (256 VGPRs), 4 = one (512 VGPRs, half the LDS / L2 traffic)
#endif
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define SG_MFMA 0x008
#define SG_VALU 0x002
#define SG_DSRD 0x100
#define SG_VMEM 0x020

// pieces [p0, p1) (1 KiB each, p1 - p0 == 16) of one chunk image -> the same pieces of an LDS buffer: wave w copies
// 16 / PIPE_WAVES consecutive pieces as LDS-DMA instructions that differ only in their immediate offset (one lane
// address and one M0 value per call instead of a 64-bit add per piece)
__device__ __forceinline__ void dma_pieces(const _Float16* gchunk, _Float16* lbuf, int p0, int p1, int wave, int lane) {
    constexpr int PER = 16 / PIPE_WAVES;
    const int first = p0 + wave * PER;
    const __attribute__((address_space(1))) void* g = (const __attribute__((address_space(1))) void*)(gchunk + first * 512 + lane * 8);
    __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(lbuf + first * 512);
    __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    if (PER > 1) __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
    if (PER > 2) __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
    if (PER > 3) __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
}
__device__ __forceinline__ half8 cat4(half2v a, half2v b, half2v c, half2v d) {
    typedef _Float16 half4v __attribute__((ext_vector_type(4)));
    const half4v ab = __builtin_shufflevector(a, b, 0, 1, 2, 3), cd = __builtin_shufflevector(c, d, 0, 1, 2, 3);
    return __builtin_shufflevector(ab, cd, 0, 1, 2, 3, 4, 5, 6, 7);
}
// hi/lo halves of four values, 1.5 VALU per value (s3d_split2: v_cvt_pk_f16_f32 + v_fma_mix{lo,hi}_f16, bit-identical to
// convert - subtract - convert).  The partial-register results feed MFMAs a whole phase later: no settle needed here.
__device__ __forceinline__ void split4_pk(float a0, float a1, float a2, float a3, half2v& h0, half2v& h1, half2v& l0, half2v& l1) {
    unsigned uh0, ul0, uh1, ul1;
    s3d_split2(a0, a1, uh0, ul0);
    s3d_split2(a2, a3, uh1, ul1);
    h0 = __builtin_bit_cast(half2v, uh0); l0 = __builtin_bit_cast(half2v, ul0);
    h1 = __builtin_bit_cast(half2v, uh1); l1 = __builtin_bit_cast(half2v, ul1);
}
// relu as ONE v_max_f32: fmaxf on an MFMA result costs a canonicalising v_max first, and hipcc folds
// __builtin_amdgcn_fmed3f(v, 0, inf) back into that pair (seen in the ISA: 8 v_max per D tile)
__device__ __forceinline__ float relu1(float v) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}
// relu + split of the 4 pre-activations of one D tile: hi = f16(max(v,0)), lo = f16(max(v,0) - hi)
template <bool SINGLE, bool BF = false>
__device__ __forceinline__ void relu_split4(const f32x4 v, half2v& h0, half2v& h1, half2v& l0, half2v& l1) {
    if (BF) {   // bf16 halves of relu(v); the low halves are not used by the single-pass product
        const float inf = __builtin_inff();
        h0 = __builtin_bit_cast(half2v, bf16_pair(__builtin_amdgcn_fmed3f(v[0], 0.f, inf), __builtin_amdgcn_fmed3f(v[1], 0.f, inf)));
        h1 = __builtin_bit_cast(half2v, bf16_pair(__builtin_amdgcn_fmed3f(v[2], 0.f, inf), __builtin_amdgcn_fmed3f(v[3], 0.f, inf)));
        l0 = h0; l1 = h1;
        return;
    }
    if (SINGLE) {   // the single-pass mode pins 9 VALU slots per MFMA with sched_group_barrier, which does not count inline asm:
                    // with the asm ReLU its groups ran dry and the mode lost 20 % (13.5 vs 16.8 M query-points/s) — builtin here
        const float inf = __builtin_inff();
        split4_pk(__builtin_amdgcn_fmed3f(v[0], 0.f, inf), __builtin_amdgcn_fmed3f(v[1], 0.f, inf),
                  __builtin_amdgcn_fmed3f(v[2], 0.f, inf), __builtin_amdgcn_fmed3f(v[3], 0.f, inf), h0, h1, l0, l1);
        return;
    }
    split4_pk(relu1(v[0]), relu1(v[1]), relu1(v[2]), relu1(v[3]), h0, h1, l0, l1);
}

#define SB() __builtin_amdgcn_sched_barrier(0)
struct FfnTrainArgs {   // MODE 2
    float* Uout;     // pre-LayerNorm output u = x + dropout(FFN(x)), saved for the backward
    DropCfg dh, dq;  // hidden-unit / output dropout
    unsigned* Mout;  // activity bits of the hidden units (post-dropout h > 0), layout: ffn_mask_dword / FFN_MASK_POS (decode.h)
    _Float16* ImgD;  // optional: D^T / R operand images of the INPUT rows for the weight-gradient kernel (decode.h)
    _Float16* ImgR;
};
struct FfnBwdArgs {     // MODE 4
    const float* Dres;     // added to dX (the residual branch's gradient)
    const unsigned* M;     // activity bits written by the forward
    float gate_scale;      // value of a kept unit's dropout factor
    _Float16* ImgD;        // optional: D^T / R operand images of the dY rows (decode.h)
    _Float16* ImgR;
    DropCfg dq;            // p > 0: the rows handed in are du (the LayerNorm backward's output) and dY = du * mask of the FFN's
                           // output dropout is rebuilt here from the counter-based masks (index row * 128 + channel, as the
                           // forward's epilogue draws them) instead of being stored and re-read
};
// per-wave state of the activation stage between GEMM1 and GEMM2
struct FfnActState {
    unsigned mw[PIPE_R];               // MODE 2: bits of the 4-chunk group being produced; MODE 4: of the group being consumed
    unsigned mw_next[PIPE_R];          // MODE 4: next group's dword (requested one group ahead)
    unsigned rm[PIPE_R];               // MODE 2: folded dropout counter of (row, hidden units 4g..4g+3) >> 2, see ffn_act4
    unsigned key;                      // MODE 2: stream key of the hidden-dropout site
    unsigned h, h2;                    // MODE 2: the hash words of the NEXT D tile's four units (drawn one MFMA group ahead)
};
// MODE 2: hash words of D tile (a2, r2) of chunk c — the counter (row*2048 + unit) >> 2 = row*512 | (8c + 4a + g) never
// carries into the row part: its fold is rm ^ (8c + 4a), one xor + add with the key instead of 64-bit adds and the high
// word's multiply — the same hash value s3d_drop4 computes.  Data independent: drawn in the MFMA group BEFORE the one
// that applies it (the odd groups of phase A carry no activation work), so the ~16 dependent VALU instructions (four of
// them quarter-rate multiplies) do not queue up behind the tile's own 24.
__device__ __forceinline__ void ffn_draw(FfnActState& as, int c, int a2, int r2) {
    as.h = s3d_hash32_rounds((as.rm[r2] ^ (unsigned)(8 * c + 4 * a2)) + as.key);
    as.h2 = s3d_drop_remix(as.h);
    asm volatile("" : "+v"(as.h), "+v"(as.h2));   // stay in this scheduling region
}
// D tile (a2, r2) of chunk c: pre-activation -> f16 hi/lo halves of GEMM2's B operand
template <int MODE, bool SINGLE, bool BF = false>
__device__ __forceinline__ void ffn_act4(const f32x4 v, half2v& h0, half2v& h1, half2v& l0, half2v& l1, FfnActState& as,
                                         const FfnTrainArgs& ta, const FfnBwdArgs& ba, int c, int a2, int r2) {
    if (MODE == 2 || MODE == 3) {
        float a[4] = {relu1(v[0]), relu1(v[1]), relu1(v[2]), relu1(v[3])};
        if (MODE == 2) {   // hidden-unit dropout (a template mode, not a run-time branch: the activation must stay in the
                           // MFMA group's basic block to be interleaved with it).  This mode is VALU-issue bound (the
                           // 192 MFMAs of two chunks leave 576 issue slots, 32-bit multiplies take four), so:
                           //  * the hash words were drawn one group ahead (ffn_draw);
                           //  * a kept unit keeps its value here, the factor 1/(1-p) multiplies the finished sums in
                           //    the epilogue (GEMM2 is linear in h);
            const unsigned h = as.h, h2 = as.h2;
            a[0] = (h & 0xFFFFu) >= ta.dh.thresh ? a[0] : 0.f;
            a[1] = (h >> 16) >= ta.dh.thresh ? a[1] : 0.f;
            a[2] = (h2 & 0xFFFFu) >= ta.dh.thresh ? a[2] : 0.f;
            a[3] = (h2 >> 16) >= ta.dh.thresh ? a[3] : 0.f;
        }
        split4_pk(a[0], a[1], a[2], a[3], h0, h1, l0, l1);
        // activity bits from the packed hi halves (a >= 0: the f16 pattern is 0 or >= 1 as an integer; a value so small
        // that its hi half is 0 has lo = 0 too and contributes nothing to the forward): min(u16, 1) of both halves at
        // once, the two pairs merged and deposited with one shift-or each — 4 VALU per tile (FFN_MASK_POS, decode.h)
        {
            unsigned p0, p1;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(p0) : "v"(__builtin_bit_cast(unsigned, h0)), "s"(0x00010001u));
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(p1) : "v"(__builtin_bit_cast(unsigned, h1)), "s"(0x00010001u));
            asm("v_lshl_or_b32 %0, %1, 2, %0" : "+v"(p0) : "v"(p1));
            const unsigned sh = (unsigned)(4 * (c & 3) + a2);
            asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(as.mw[r2]) : "v"(p0), "s"(sh));
        }
    } else if (MODE == 4) {
        // gate = the unit's activity bit: a 1-bit signed field is the AND mask itself (2 VALU per value); a kept unit's
        // dropout factor multiplies the finished sums in the epilogue (GEMM2 is linear in dA)
        const unsigned bits = as.mw[r2] >> (4 * (c & 3) + a2);
        float a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[i] = s3d_gate_bit_imm(v[i], bits, FFN_MASK_POS(i));
        split4_pk(a[0], a[1], a[2], a[3], h0, h1, l0, l1);
    } else {
        relu_split4<SINGLE, BF>(v, h0, h1, l0, l1);
    }
}

// LDS fragment reads and their waits are issued by hand.  hipcc models a pending LDS-DMA as an LDS access of unknown
// order: while the refill of the other buffer is in flight it either degrades every LDS wait to lgkmcnt(0) (waiting
// for the reads it has just issued for the NEXT group) or guards each read with vmcnt(0) (waiting for the refill) —
// measured in the ISA of three variants of this loop.  An asm read is invisible to that bookkeeping; DS_WAIT names the
// registers it releases, so their consumers cannot be scheduled above it.  LDS returns in order: lgkmcnt(2) leaves
// exactly the two reads of the next group outstanding.
#define DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define DS_WAIT2(n, r0, r1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(n))
#define DS_WAIT4(n, r0, r1, r2, r3) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(n))
#define DS_WAIT1(n, r0) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r0) : "n"(n))
#define DS_WAIT3(n, r0, r1, r2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(r0), "+v"(r1), "+v"(r2) : "n"(n))
#define DS_WAIT5(n, r0, r1, r2, r3, r4) \
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4) : "n"(n))

// phase A group K = (u, a): reads of group K+1 (or of phase B's first tile), 6 MFMAs, VALU slice on even K
#define FFN_GROUP_A(K)                                                                                               \
    {                                                                                                                \
        constexpr int u = (K) >> 1, a = (K) & 1, s = (K) & 1;                                                        \
        if (!LAST) {                                                                                                 \
            if ((K) < 7) {                                                                                           \
                DS_READ(fh[s ^ 1], lw, ((((K) + 1) & 1) * 4 + (((K) + 1) >> 1)) * 1024);                             \
                DS_READ(fl[s ^ 1], lw, ((((K) + 1) & 1) * 4 + (((K) + 1) >> 1)) * 1024 + 8192);                      \
            } else {                                                                                                 \
                DS_READ(vh[0], lw, 16384);                                                                           \
                DS_READ(vl[0], lw, 24576);                                                                           \
            }                                                                                                        \
            if ((K) == 0 && MODE == 4) {                                                                             \
                DS_WAIT2(2, fh[0], fl[0]);                                                                           \
                _Pragma("unroll") for (int r = 0; r < PIPE_R; ++r) { hn[0][r] = zero4(); hn[1][r] = zero4(); }        \
            } else if ((K) == 0) {                                                                                   \
                DS_WAIT4(2, bq[0], bq[1], fh[0], fl[0]);                                                             \
                _Pragma("unroll") for (int r = 0; r < PIPE_R; ++r) { hn[0][r] = bq[0]; hn[1][r] = bq[1]; }            \
            } else {                                                                                                 \
                DS_WAIT2(2, fh[s], fl[s]);                                                                           \
            }                                                                                                        \
            if (!SINGLE) {                                                                                           \
                _Pragma("unroll") for (int r = 0; r < PIPE_R; ++r)                                                    \
                    hn[a][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[s], xl[r][u], hn[a][r], 0, 0, 0);           \
                _Pragma("unroll") for (int r = 0; r < PIPE_R; ++r)                                                    \
                    hn[a][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[s], xh[r][u], hn[a][r], 0, 0, 0);           \
            }                                                                                                        \
            _Pragma("unroll") for (int r = 0; r < PIPE_R; ++r)                                                        \
                hn[a][r] = mfma_hh<BF>(fh[s], xh[r][u], hn[a][r]);                                                   \
        } else if ((K) == 7) {                                                                                       \
            DS_READ(vh[0], lw, 16384);                                                                               \
            DS_READ(vl[0], lw, 24576);                                                                               \
        }                                                                                                            \
        if ((K) % (4 / PIPE_R) == 0) {   /* 2 * PIPE_R D tiles over the 8 groups */                                   \
            constexpr int tile = (K) / (4 / PIPE_R), a2 = tile / PIPE_R, r2 = tile % PIPE_R;                         \
            ffn_act4<MODE, SINGLE, BF>(hd[a2][r2], hh2[r2][2 * a2], hh2[r2][2 * a2 + 1], hl2[r2][2 * a2], hl2[r2][2 * a2 + 1], as, ta, ba, c, a2, r2); \
            /* tie the results into the side-effect chain: otherwise the low halves are emitted where they are */   \
            /* first USED (phase B), outside the MFMA cover */                                                       \
            asm volatile("" : "+v"(hl2[r2][2 * a2]), "+v"(hl2[r2][2 * a2 + 1]), "+v"(hh2[r2][2 * a2]),               \
                         "+v"(hh2[r2][2 * a2 + 1]));                                                                 \
        } else if (MODE == 2) {          /* the next tile's dropout words (tile 0 of chunk c + 1 after the last) */   \
            static_assert(PIPE_R == 2, "one activation tile every second group");                                    \
            constexpr int nt = (((K) + 1) / 2) & 3;                                                                   \
            ffn_draw(as, (K) == 7 ? c + 1 : c, nt / PIPE_R, nt % PIPE_R);                                             \
        }                                                                                                            \
        if (!LAST) {                                                                                                 \
            _Pragma("unroll") for (int i = 0; i < (SINGLE ? 1 : 3) * PIPE_R; ++i) {                                  \
                SGB(SG_MFMA, 1);                                                                                     \
                SGB(SG_VALU, SINGLE ? 9 : (MODE == 2 ? 4 : 3));                                                      \
            }                                                                                                        \
        }                                                                                                            \
        SB();                                                                                                        \
    }
// phase B group J = output tile
#define FFN_GROUP_B(J)                                                                                               \
    {                                                                                                                \
        constexpr int s = (J) & 1;                                                                                   \
        if ((J) < 7) {                                                                                               \
            DS_READ(vh[s ^ 1], lw, 16384 + ((J) + 1) * 1024);                                                        \
            DS_READ(vl[s ^ 1], lw, 24576 + ((J) + 1) * 1024);                                                        \
            DS_WAIT2(2, vh[s], vl[s]);                                                                               \
        } else {                                                                                                     \
            DS_WAIT2(0, vh[s], vl[s]);                                                                               \
        }                                                                                                            \
        if (!SINGLE) {                                                                                               \
            _Pragma("unroll") for (int r = 0; r < PIPE_R; ++r)                                                        \
                acc[r][J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[s], hl[r], acc[r][J], 0, 0, 0);                \
            _Pragma("unroll") for (int r = 0; r < PIPE_R; ++r)                                                        \
                acc[r][J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[s], hh[r], acc[r][J], 0, 0, 0);                \
        }                                                                                                            \
        _Pragma("unroll") for (int r = 0; r < PIPE_R; ++r)                                                            \
            acc[r][J] = mfma_hh<BF>(vh[s], hh[r], acc[r][J]);                                                        \
        SB();                                                                                                        \
    }

// One iteration of the pipelined loop (force-inlined twice per trip with hd / hn exchanged, so the hand-over of the
// pre-activations costs no moves).  Every group of 6 MFMAs is its own scheduling region (sched_barrier): the order
// written here IS the issue order.  lw = LDS byte address of this lane's fragment slot in the current weight buffer,
// lb = of its bias quad of the NEXT chunk.
template <int MODE, bool LAST, bool SINGLE, bool BF = false>
__device__ __forceinline__ void ffn_pipe_iter(const unsigned lw, const unsigned lb, const half8 (&xh)[PIPE_R][4],
                                              const half8 (&xl)[PIPE_R][4], f32x4 (&acc)[PIPE_R][8],
                                              const f32x4 (&hd)[2][PIPE_R], f32x4 (&hn)[2][PIPE_R], FfnActState& as,
                                              const FfnTrainArgs& ta, const FfnBwdArgs& ba, const int c) {
    half2v hh2[PIPE_R][4], hl2[PIPE_R][4];
    half8 fh[2], fl[2], vh[2], vl[2];
    f32x4 bq[2];
    if (!LAST) {
        if (MODE != 4) {
            DS_READ(bq[0], lb, 0);
            DS_READ(bq[1], lb, 64);
        }
        DS_READ(fh[0], lw, 0);
        DS_READ(fl[0], lw, 8192);
    }
    SB();
    // ---- phase A: hn = W1(c+1) x^T + b1(c+1)   beside   relu / split of hd (chunk c)
    FFN_GROUP_A(0) FFN_GROUP_A(1) FFN_GROUP_A(2) FFN_GROUP_A(3) FFN_GROUP_A(4) FFN_GROUP_A(5) FFN_GROUP_A(6) FFN_GROUP_A(7)
    // ---- phase B: acc += W2(c) h(c)
    half8 hh[PIPE_R], hl[PIPE_R];
#pragma unroll
    for (int r = 0; r < PIPE_R; ++r) {
        hh[r] = cat4(hh2[r][0], hh2[r][1], hh2[r][2], hh2[r][3]);
        hl[r] = cat4(hl2[r][0], hl2[r][1], hl2[r][2], hl2[r][3]);
    }
    FFN_GROUP_B(0) FFN_GROUP_B(1) FFN_GROUP_B(2) FFN_GROUP_B(3) FFN_GROUP_B(4) FFN_GROUP_B(5) FFN_GROUP_B(6) FFN_GROUP_B(7)
}

// SINGLE: S3D_PREC_F16 — only the hi*hi product of every split (one MFMA per product; not fp32-class)
template <int MODE, bool SINGLE, bool BF = false>
__global__ __launch_bounds__(PIPE_THREADS, (PIPE_R == 2 && PIPE_WAVES == 4) ? 2 : 1) void ffn_layer_f16x3_pipe_kernel(const float* X, float* Yout, long rows,
                                                                   const _Float16* wimg, const LayerPtrs w,
                                                                   const float* fco_w, const float* fco_b,
                                                                   float* sdf_out, float sign, long groups_per_batch,
                                                                   long n_qry, long g_begin, const int* perm,
                                                                   const FfnTrainArgs ta, const FfnBwdArgs ba, const int pre_ln1) {
    constexpr bool FINAL = MODE == 1;
    static_assert(!BF || (SINGLE && MODE <= 1), "the bf16 mode is a single-pass inference mode");
    constexpr int NC = S3D_FFN_NCHUNK;
    // THREE distinct LDS objects: hipcc tags their accesses with alias scopes, so a ds_read of one weight buffer is not
    // guarded (s_waitcnt vmcnt(0)) against the LDS-DMA refill of the OTHER buffer in flight — with one two-buffer array
    // it is, or, when the array is the kernel's only LDS object, every LDS wait degrades to lgkmcnt(0)
    __shared__ __attribute__((aligned(16))) _Float16 s_w0[F16_CHUNK_HALFS], s_w1[F16_CHUNK_HALFS];   // 2 x 32 KiB
    __shared__ __attribute__((aligned(16))) float s_b1[S3D_FFN];                                     // 8 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    const long row0 = ((long)blockIdx.x * PIPE_WAVES + wave) * (PIPE_R * 16);
    // the W1 half (pieces 0-15) of chunk ch -> LDS buffer
    auto dma_w1 = [&](int ch, _Float16* lbuf) {
        dma_pieces(wimg + (size_t)ch * F16_CHUNK_HALFS, lbuf, 0, 16, wave, lane);
    };

    // prologue DMA: W1(0) -> buffer 1 (W1 half);  buffer 0 <- W1(1) | W2(0)
    dma_w1(0, s_w1);
    dma_w1(1, s_w0);
    dma_pieces(wimg, s_w0, 16, 32, wave, lane);
    if (MODE != 4)
        for (int i = threadIdx.x; i < S3D_FFN / 4; i += PIPE_THREADS) st4(s_b1 + 4 * i, ld4(w.b1 + 4 * i));

    half8 xh[PIPE_R][4], xl[PIPE_R][4];
    f32x4 acc[PIPE_R][8];
    FfnActState as;
#pragma unroll
    for (int r = 0; r < PIPE_R; ++r) {
        long row = row0 + r * 16 + m;
        if (row >= rows) row = rows - 1;
        const float* p = X + row * 128 + 8 * g;
        // FINAL with pre_ln1: the rows handed in are the pre-LayerNorm sums u of the last layer's attention block and
        // LayerNorm1 (w.ln1g / w.ln1b) is applied here, on the row's 32 values per lane and a quad reduction — the ln_fwd
        // launch and one round trip of the token-0 rows are gone (same arithmetic as ln_fwd_kernel: two-pass mean / variance)
        float lmean = 0.f, lrstd = 1.f;
        if (FINAL && pre_ln1) {
            float sacc = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 a = ld4(p + 32 * u), b = ld4(p + 32 * u + 4);
                sacc += ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
            }
            lmean = quad_sum16(sacc) * (1.f / 128.f);
            float vacc = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 a = ld4(p + 32 * u), b = ld4(p + 32 * u + 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) vacc += (a[t] - lmean) * (a[t] - lmean) + (b[t] - lmean) * (b[t] - lmean);
            }
            lrstd = 1.f / sqrtf(quad_sum16(vacc) * (1.f / 128.f) + 1e-5f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 a = ld4(p + 32 * u), b = ld4(p + 32 * u + 4);
            float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            if (FINAL && pre_ln1) {
                const f32x4 ga = ld4(w.ln1g + 32 * u + 8 * g), gb = ld4(w.ln1g + 32 * u + 8 * g + 4);
                const f32x4 ba4 = ld4(w.ln1b + 32 * u + 8 * g), bb4 = ld4(w.ln1b + 32 * u + 8 * g + 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    v[t] = (v[t] - lmean) * lrstd * ga[t] + ba4[t];
                    v[4 + t] = (v[4 + t] - lmean) * lrstd * gb[t] + bb4[t];
                }
            }
            if (MODE == 4 && ba.dq.p > 0.f) {
                float mk0[4], mk1[4];
                s3d_drop4(ba.dq, (unsigned long long)row * 128 + 32 * u + 8 * g, mk0);
                s3d_drop4(ba.dq, (unsigned long long)row * 128 + 32 * u + 8 * g + 4, mk1);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    v[t] *= mk0[t];
                    v[4 + t] *= mk1[t];
                }
            }
            if (BF) split8_bf(v, xh[r][u], xl[r][u]);
            else split8(v, xh[r][u], xl[r][u]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[r][j] = zero4();
        as.mw[r] = as.mw_next[r] = 0u;
        as.rm[r] = 0u;
        as.key = 0u;
        if (MODE == 2) {
            as.rm[r] = s3d_hash32_fold((unsigned long long)(row0 + r * 16 + m) * (S3D_FFN / 4)) ^ (unsigned)g;
            as.key = s3d_stream_key(ta.dh.seed, ta.dh.site);
        }
        as.h = as.h2 = 0u;
        if (MODE == 4) {   // activity bits: one dword per 4 chunks, the next group's requested one group ahead
            as.mw[r] = ba.M[ffn_mask_dword(row, 0, g)];
            as.mw_next[r] = ba.M[ffn_mask_dword(row, 1, g)];
        }
    }
    if (MODE == 2) ffn_draw(as, 0, 0, 0);
    dma_publish_barrier();
    const float* sb = s_b1 + 4 * g;     // this lane's bias quad of D tile 0 of chunk 0; tile 1 at +16, chunk c at +32c
    const unsigned lw0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_w0 + lane * 8);
    const unsigned lw1 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_w1 + lane * 8);
    const unsigned lb0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_b1 + 4 * g);
    f32x4 hdA[2][PIPE_R], hdB[2][PIPE_R];   // pre-activations (bias included) of the current / next chunk, D tiles a = 0,1
    {
        const _Float16* sw = s_w1;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < PIPE_R; ++r) hdA[a][r] = MODE == 4 ? zero4() : *reinterpret_cast<const f32x4*>(sb + 16 * a);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const half8 fh = ldh8(sw + ((a * 4 + u) * 64 + lane) * 8);
                if (!SINGLE) {
                    const half8 fl = ldh8(sw + 4096 + ((a * 4 + u) * 64 + lane) * 8);
#pragma unroll
                    for (int r = 0; r < PIPE_R; ++r) hdA[a][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, xl[r][u], hdA[a][r], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < PIPE_R; ++r) hdA[a][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, xh[r][u], hdA[a][r], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < PIPE_R; ++r) hdA[a][r] = mfma_hh<BF>(fh, xh[r][u], hdA[a][r]);
            }
    }
    __syncthreads();   // every wave is done with buffer 1 before the first refill overwrites it

    // between two iterations, after chunk c (c & 3 == 3) closed a 4-chunk group of activity bits:
    //   MODE 2 stores the group's dword, MODE 4 moves on to the next group's and requests the one after
    auto group_done = [&](int c) {
        if ((MODE == 2 || MODE == 3) && ta.Mout) {
#pragma unroll
            for (int r = 0; r < PIPE_R; ++r) {
                const long row = row0 + r * 16 + m;
                if (row < rows) ta.Mout[ffn_mask_dword(row, c >> 2, g)] = as.mw[r];   // 256 contiguous bytes per (tile, group)
                as.mw[r] = 0u;
            }
        }
        if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < PIPE_R; ++r) {
                long row = row0 + r * 16 + m;
                if (row >= rows) row = rows - 1;
                const int nxt = (c >> 2) + 2;
                as.mw[r] = as.mw_next[r];
                as.mw_next[r] = ba.M[ffn_mask_dword(row, nxt < NC / 4 ? nxt : NC / 4 - 1, g)];
            }
        }
    };

    // buffer c & 1 holds W1(c+1) | W2(c); the refill of the other buffer (W1(c+2) | W2(c+1)) is requested at the top of
    // iteration c and published by the barrier at its end.  Two iterations per trip: the buffers are distinct objects
    // (see above) and hd / hn exchange roles without moves.
#pragma unroll 1
    for (int c = 0; c < NC - 2; c += 2) {
        dma_w1(c + 2, s_w1);
        dma_pieces(wimg + (size_t)(c + 1) * F16_CHUNK_HALFS, s_w1, 16, 32, wave, lane);
        SB();
        ffn_pipe_iter<MODE, false, SINGLE, BF>(lw0, lb0 + (c + 1) * S3D_FFN_CHUNK * 4, xh, xl, acc, hdA, hdB, as, ta, ba, c);
        dma_publish_barrier();
        dma_w1(c + 3, s_w0);
        dma_pieces(wimg + (size_t)(c + 2) * F16_CHUNK_HALFS, s_w0, 16, 32, wave, lane);
        SB();
        ffn_pipe_iter<MODE, false, SINGLE, BF>(lw1, lb0 + (c + 2) * S3D_FFN_CHUNK * 4, xh, xl, acc, hdB, hdA, as, ta, ba, c + 1);
        if ((MODE == 2 || MODE == 3 || MODE == 4) && (c & 2)) group_done(c + 1);
        dma_publish_barrier();
    }
    // c = NC-2: buffer 0 holds W1(NC-1) | W2(NC-2); only W2(NC-1) is left to fetch (the W1 half: any valid chunk)
    dma_w1(NC - 1, s_w1);
    dma_pieces(wimg + (size_t)(NC - 1) * F16_CHUNK_HALFS, s_w1, 16, 32, wave, lane);
    SB();
    ffn_pipe_iter<MODE, false, SINGLE, BF>(lw0, lb0 + (NC - 1) * S3D_FFN_CHUNK * 4, xh, xl, acc, hdA, hdB, as, ta, ba, NC - 2);
    dma_publish_barrier();
    ffn_pipe_iter<MODE, true, SINGLE, BF>(lw1, lb0, xh, xl, acc, hdB, hdA, as, ta, ba, NC - 1);
    if (MODE == 2 || MODE == 3) group_done(NC - 1);
    if (MODE >= 2) {   // operand images of this wave's 32 rows for the weight-gradient kernel (layout: decode.h).  Written HERE, after
                       // the loop: in the prologue their 32 stores sat in front of the publishing barrier's vmcnt(0)
                       // (+1 ms per 5.2 M-row call); the row fragments are live to the epilogue anyway
        _Float16* imgd = MODE == 4 ? ba.ImgD : ta.ImgD;
        _Float16* imgr = MODE == 4 ? ba.ImgR : ta.ImgR;
        if (imgd && row0 < rows) {   // (a wave wholly past the end has no block in the images)
            const long blk = row0 >> 5;
            _Float16* rb = imgr + blk * FWR_BLK_HALFS;
            _Float16* db = imgd + blk * FWR_BLK_HALFS;
            const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
            half8 sel[2];   // B operand that picks channels 16c .. 16c+15 of a 32-channel k-step: B[k = 8g + t][n] = (k == 16c + n)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int t = 0; t < 8; ++t) sel[c2][t] = (8 * g + t == 16 * c2 + m) ? (_Float16)1.f : (_Float16)0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                half8 ah[PIPE_R], al[PIPE_R];
#pragma unroll
                for (int r = 0; r < PIPE_R; ++r) {   // rows past the end are zero in both images
                    const bool ok = row0 + r * 16 + m < rows;
                    ah[r] = ok ? xh[r][u] : z8;
                    al[r] = ok ? xl[r][u] : z8;
                    // R image: the fragment as it is — chunk 4u + g of row 16r + m, swizzled by the row
                    const int o = (16 * r + m) * 128 + (((4 * u + g) ^ m) << 3);
                    *reinterpret_cast<half8*>(rb + o) = ah[r];
                    *reinterpret_cast<half8*>(rb + 4096 + o) = al[r];
                }
                // D^T image: x tile (A: row m, k-slot = channel) times the selector -> D layout lane (channel n, g) x rows 4g + i:
                // the slot order of the image (tile 0 rows | tile 1 rows); products with 1.0 are exact
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    static_assert(PIPE_R == 2, "a wave's two row tiles are one 32-row block");
                    const f32x4 h0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], sel[c2], zero4(), 0, 0, 0);
                    const f32x4 h1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], sel[c2], zero4(), 0, 0, 0);
                    const f32x4 l0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[0], sel[c2], zero4(), 0, 0, 0);
                    const f32x4 l1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[1], sel[c2], zero4(), 0, 0, 0);
                    half8 oh, ol;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        oh[i] = (_Float16)h0[i]; oh[4 + i] = (_Float16)h1[i];
                        ol[i] = (_Float16)l0[i]; ol[4 + i] = (_Float16)l1[i];
                    }
                    const int q = 32 * u + 16 * c2 + m;
                    const int o = q * 32 + ((g ^ fwr_dperm(q)) << 3);
                    *reinterpret_cast<half8*>(db + o) = oh;
                    *reinterpret_cast<half8*>(db + 4096 + o) = ol;
                }
            }
        }
    }

    // epilogue: tile j, reg i  <->  column 32*(j>>1) + 8*g + 4*(j&1) + i.
    // The lane's column offset is re-derived from an opaque copy of g: otherwise the ten loop-invariant 64-bit addresses
    // of this epilogue are computed in the prologue and spilled across the loop (20 dwords of scratch per lane, written
    // and read back by every wave: +0.4 GB of HBM writes per launch in the PMC pass)
    int ge = g, me = m;
    asm volatile("" : "+v"(ge), "+v"(me));
    // Stores go out as full 128-byte lines (s3d_full_line_pair, common.h): a lane's tiles 2J, 2J + 1 are 32 contiguous
    // bytes of row m, a plain store writes 16-byte pieces at a 32-byte stride; after the exchange with lane m ^ 8 one
    // instruction covers rows 0-7 of the tile and the next rows 8-15, every row's line whole.
    auto store_pair = [&](float* base, int r, int J, const f32x4 v0, const f32x4 v1) {
        f32x4 va, vb;
        s3d_full_line_pair(v0, v1, me, va, vb);
        const long ra = row0 + r * 16 + (me & 7);
        float* o = base + ra * 128 + 32 * J + 8 * ge + 4 * (me >> 3);
        if (ra < rows) st4(o, va);
        if (ra + 8 < rows) st4(o + 8 * 128, vb);
    };
    if (MODE == 4) {   // dX = dA W1 + Dres
#pragma unroll
        for (int r = 0; r < PIPE_R; ++r) {
            long row = row0 + r * 16 + me;
            if (row >= rows) row = rows - 1;
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                const int col = 32 * J + 8 * ge;
                store_pair(Yout, r, J, acc[r][2 * J] * ba.gate_scale + ld4(ba.Dres + row * 128 + col),
                           acc[r][2 * J + 1] * ba.gate_scale + ld4(ba.Dres + row * 128 + col + 4));
            }
        }
        return;
    }
    // ... and the rows' halves are made opaque here: left alone, the residual f32(hi) + f32(lo) of all 64 values is
    // formed BEFORE the loop (it is loop-invariant), held in registers through it and partly spilled
#pragma unroll
    for (int r = 0; r < PIPE_R; ++r)
        asm volatile("" : "+v"(xh[r][0]), "+v"(xh[r][1]), "+v"(xh[r][2]), "+v"(xh[r][3]), "+v"(xl[r][0]), "+v"(xl[r][1]),
                     "+v"(xl[r][2]), "+v"(xl[r][3]));
#pragma unroll
    for (int r = 0; r < PIPE_R; ++r) {
        const long row = row0 + r * 16 + me;
        f32x4 y[8];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = 32 * (j >> 1) + 8 * ge + 4 * (j & 1);
            const f32x4 b2 = ld4(w.b2 + col);
            float mq4[4] = {1.f, 1.f, 1.f, 1.f};
            if (MODE == 2) s3d_drop4(ta.dq, (unsigned long long)row * 128 + col, mq4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 4 * (j & 1) + i;
                float f = MODE == 2 ? acc[r][j][i] * ta.dh.scale + b2[i] : acc[r][j][i] + b2[i];   // hidden-dropout factor, see ffn_act4
                if (MODE == 2) f *= mq4[i];
                y[j][i] = f + ((BF ? bf16_lane(xh[r][j >> 1], t) : (float)xh[r][j >> 1][t]) + (float)xl[r][j >> 1][t]);
                s += y[j][i];
            }
            if ((MODE == 2 || MODE == 3) && (j & 1)) store_pair(ta.Uout, r, j >> 1, y[j - 1], y[j]);
        }
        const float mean = quad_sum16(s) * (1.f / 128.f);
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = y[j][i] - mean;
                v += d * d;
            }
        const float rstd = 1.f / sqrtf(quad_sum16(v) * (1.f / 128.f) + 1e-5f);
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = 32 * (j >> 1) + 8 * ge + 4 * (j & 1);
            const f32x4 ga = ld4(w.ln2g + col), be = ld4(w.ln2b + col);
#pragma unroll
            for (int i = 0; i < 4; ++i) y[j][i] = (y[j][i] - mean) * rstd * ga[i] + be[i];
            if (FINAL) {
                const f32x4 wo = ld4(fco_w + col);
                dot += y[j][0] * wo[0] + y[j][1] * wo[1] + y[j][2] * wo[2] + y[j][3] * wo[3];
            } else if (j & 1) {
                store_pair(Yout, r, j >> 1, y[j - 1], y[j]);
            }
        }
        if (FINAL) {
            dot = quad_sum16(dot) + fco_b[0];
            if (ge == 0 && row < rows) {
                const long grp = g_begin + row / S3D_GROUP;
                const long b = grp / groups_per_batch;
                const long q = (grp % groups_per_batch) * S3D_GROUP + (row % S3D_GROUP);
                if (q < n_qry) sdf_out[b * n_qry + (perm ? perm[b * n_qry + q] : q)] = sign * dot;
            }
        }
    }
}

#define PIPE_ROWS_PER_WG (PIPE_WAVES * PIPE_R * 16)
int launch_ffn_layer_f16x3(float* X, long rows, const LayerPtrs& w, const float* wimg, const float* fco_w,
                           const float* fco_b, float* sdf_out, float sign, long groups_per_batch, long n_qry,
                           long g_begin, const int* perm, hipStream_t stream, bool single_pass, bool pre_ln1, bool bf16) {
    if (rows <= 0) return 0;
    S3D_CHECK_ARG(!bf16 || single_pass, "ffn: the bf16 mode is a single-pass mode");
    S3D_CHECK_ARG(!pre_ln1 || sdf_out, "ffn: the LayerNorm1 prologue belongs to the final layer's kernel");
    const int PRE_LN_ARG = pre_ln1 ? 1 : 0;
    const _Float16* img = reinterpret_cast<const _Float16*>(wimg);
    const FfnTrainArgs ta = {};
    const FfnBwdArgs ba = {};
    const dim3 grid((unsigned)((rows + PIPE_ROWS_PER_WG - 1) / PIPE_ROWS_PER_WG)), block(PIPE_THREADS);
    if (bf16) {   // wimg: the bf16 image of the layer
        if (sdf_out)
            hipLaunchKernelGGL((ffn_layer_f16x3_pipe_kernel<1, true, true>), grid, block, 0, stream, X, X, rows, img, w, fco_w, fco_b,
                               sdf_out, sign, groups_per_batch, n_qry, g_begin, perm, ta, ba, PRE_LN_ARG);
        else
            hipLaunchKernelGGL((ffn_layer_f16x3_pipe_kernel<0, true, true>), grid, block, 0, stream, X, X, rows, img, w, fco_w, fco_b,
                               sdf_out, sign, groups_per_batch, n_qry, g_begin, perm, ta, ba, PRE_LN_ARG);
    } else if (single_pass) {
        if (sdf_out)
            hipLaunchKernelGGL((ffn_layer_f16x3_pipe_kernel<1, true>), grid, block, 0, stream, X, X, rows, img, w, fco_w, fco_b,
                               sdf_out, sign, groups_per_batch, n_qry, g_begin, perm, ta, ba, PRE_LN_ARG);
        else
            hipLaunchKernelGGL((ffn_layer_f16x3_pipe_kernel<0, true>), grid, block, 0, stream, X, X, rows, img, w, fco_w, fco_b,
                               sdf_out, sign, groups_per_batch, n_qry, g_begin, perm, ta, ba, PRE_LN_ARG);
    } else if (sdf_out) {
        hipLaunchKernelGGL((ffn_layer_f16x3_pipe_kernel<1, false>), grid, block, 0, stream, X, X, rows, img, w, fco_w, fco_b,
                           sdf_out, sign, groups_per_batch, n_qry, g_begin, perm, ta, ba, PRE_LN_ARG);
    } else {
        hipLaunchKernelGGL((ffn_layer_f16x3_pipe_kernel<0, false>), grid, block, 0, stream, X, X, rows, img, w, fco_w, fco_b,
                           sdf_out, sign, groups_per_batch, n_qry, g_begin, perm, ta, ba, PRE_LN_ARG);
    }
    S3D_LAUNCH_CHECK();
    return 0;
}

// training forward: y = LN2(u), u = x + dropout(FFN(x)); y -> Yout, u -> Uout, activity bits -> Mout
int launch_ffn_layer_train_f16x3(const float* Xin, float* Yout, float* Uout, unsigned* Mout, long rows,
                                 const LayerPtrs& w, const DropCfg& drop_hidden, const DropCfg& drop_out,
                                 hipStream_t stream, float* imgd, float* imgr, bool single) {
    if (rows <= 0) return 0;
    S3D_CHECK_ARG(w.wf16 != nullptr, "ffn train f16x3: no packed f16 image");
    S3D_CHECK_ARG((imgd == nullptr) == (imgr == nullptr), "ffn train f16x3: both operand images or none");
    const FfnTrainArgs ta = {Uout, drop_hidden, drop_out, Mout, reinterpret_cast<_Float16*>(imgd), reinterpret_cast<_Float16*>(imgr)};
    const FfnBwdArgs ba = {};
    const dim3 grid((unsigned)((rows + PIPE_ROWS_PER_WG - 1) / PIPE_ROWS_PER_WG)), block(PIPE_THREADS);
    S3D_CHECK_ARG((drop_hidden.p > 0.f) == (drop_out.p > 0.f), "ffn train f16x3: hidden / output dropout must be on or off together");
#define FFN_TRAIN_LAUNCH(MODE, SGL)                                                                                             \
    hipLaunchKernelGGL((ffn_layer_f16x3_pipe_kernel<MODE, SGL>), grid, block, 0, stream, Xin, Yout, rows,                       \
                       reinterpret_cast<const _Float16*>(w.wf16), w, nullptr, nullptr, nullptr, 1.f, 1L, 1L, 0L, nullptr, ta, ba, 0)
    if (drop_hidden.p > 0.f) {
        if (single) FFN_TRAIN_LAUNCH(2, true); else FFN_TRAIN_LAUNCH(2, false);
    } else {
        if (single) FFN_TRAIN_LAUNCH(3, true); else FFN_TRAIN_LAUNCH(3, false);
    }
#undef FFN_TRAIN_LAUNCH
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// FFN backward, data path (MODE 4 of the pipelined kernel):  given dY (gradient w.r.t. the lin2 output) and the
// activity bits M of the hidden units (forward kernel, FfnTrainArgs::Mout),
//   dA = (dY W2) * bit                           [rows][2048], registers only
//   dX = gate_scale * dA W1 + Dres                            [rows][128]
// dY rows live in registers as f16 hi/lo B fragments, the transposed weights (W2^T chunk as GEMM-1-shaped fragments,
// W1^T chunk as GEMM-2-shaped fragments; packed by pack_ffn_f16x3_kernel with swapped strides) stream through LDS.
// ---------------------------------------------------------------------------------------------
int launch_ffn_bwd_dx_f16x3(const float* DY, const float* Dres, const unsigned* M, float* DX, long rows,
                            const float* timg, float gate_scale, hipStream_t stream, float* imgd, float* imgr,
                            const DropCfg* dy_mask, bool single) {
    if (rows <= 0) return 0;
    S3D_CHECK_ARG((imgd == nullptr) == (imgr == nullptr), "ffn bwd dx: both operand images or none");
    const FfnTrainArgs ta = {};
    FfnBwdArgs ba = {Dres, M, gate_scale, reinterpret_cast<_Float16*>(imgd), reinterpret_cast<_Float16*>(imgr), make_drop(0, 0.f, 0)};
    if (dy_mask) ba.dq = *dy_mask;
    const LayerPtrs w = {};
    const dim3 grid((unsigned)((rows + PIPE_ROWS_PER_WG - 1) / PIPE_ROWS_PER_WG)), block(PIPE_THREADS);
    if (single)
        hipLaunchKernelGGL((ffn_layer_f16x3_pipe_kernel<4, true>), grid, block, 0, stream, DY, DX, rows,
                           reinterpret_cast<const _Float16*>(timg), w, nullptr, nullptr, nullptr, 1.f, 1L, 1L, 0L, nullptr, ta,
                           ba, 0);
    else
        hipLaunchKernelGGL((ffn_layer_f16x3_pipe_kernel<4, false>), grid, block, 0, stream, DY, DX, rows,
                           reinterpret_cast<const _Float16*>(timg), w, nullptr, nullptr, nullptr, 1.f, 1L, 1L, 0L, nullptr, ta,
                           ba, 0);
    S3D_LAUNCH_CHECK();
    return 0;
}

// pack lin1 (2048,128) / lin2 (128,2048) into the per-chunk f16 hi/lo fragment image
// Generic strides: element (hidden h, input k) of the GEMM-1-shaped operand is w1[h*s1h + k*s1k], element
// (output n, hidden h) of the GEMM-2-shaped operand is w2[n*s2n + h*s2h].  Forward image: (lin1, 128, 1),
// (lin2, 2048, 1); backward ("transposed") image: (lin2, 1, 2048), (lin1, 1, 128).
__global__ void pack_ffn_f16x3_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                      _Float16* __restrict__ out, int s1h, int s1k, int s2n, int s2h, int bf16) {
    // one thread per (chunk, which, frag, lane): writes 8 hi halfs and 8 lo halfs (bf16: bf16 bit patterns, zero low halves)
    const int total = S3D_FFN_NCHUNK * 2 * 8 * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, frag = (idx >> 6) & 7, which = (idx >> 9) & 1, c = idx >> 10;
        const int r = lane & 15, g = lane >> 4;
        float v[8];
        if (which == 0) {      // W1 fragment (a, u) = (frag>>2, frag&3): row = hidden unit, k = 32u + 8g + t
            const int a = frag >> 2, u = frag & 3;
            const float* p = w1 + (size_t)(32 * c + 16 * a + r) * s1h + (size_t)(32 * u + 8 * g) * s1k;
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = p[(size_t)t * s1k];
        } else {               // W2 fragment j = frag: row = permuted output channel, k-slot t -> hidden unit
            const int j = frag;
            const int n = 32 * (j >> 1) + 8 * (r >> 2) + 4 * (j & 1) + (r & 3);
            const float* p = w2 + (size_t)n * s2n + (size_t)(32 * c) * s2h;
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = p[(size_t)(t < 4 ? 4 * g + t : 16 + 4 * g + (t - 4)) * s2h];
        }
        _Float16* dst = out + (size_t)c * F16_CHUNK_HALFS + which * 8192 + (frag * 64 + lane) * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (bf16) {
                dst[t] = __builtin_bit_cast(_Float16, (unsigned short)(bf16_pair(v[t], 0.f) & 0xFFFFu));
                dst[4096 + t] = (_Float16)0.f;
                continue;
            }
            const _Float16 h = (_Float16)v[t];
            dst[t] = h;
            dst[4096 + t] = (_Float16)(v[t] - (float)h);
        }
    }
}

int launch_pack_ffn_f16x3(const float* w1, const float* w2, float* out, hipStream_t stream, int bf16) {
    hipLaunchKernelGGL(pack_ffn_f16x3_kernel, dim3(256), dim3(256), 0, stream, w1, w2,
                       reinterpret_cast<_Float16*>(out), 128, 1, S3D_FFN, 1, bf16);
    S3D_LAUNCH_CHECK();
    return 0;
}
int launch_pack_ffn_f16x3_bwd(const float* w1, const float* w2, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(pack_ffn_f16x3_kernel, dim3(256), dim3(256), 0, stream, w2, w1,
                       reinterpret_cast<_Float16*>(out), 1, S3D_FFN, 1, 128, 0);
    S3D_LAUNCH_CHECK();
    return 0;
}
