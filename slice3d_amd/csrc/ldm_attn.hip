// ldm_attn.hip — QKVAttentionLegacy (openaimodel.py:353-381) for the long-sequence attention blocks of the gen_slices
// U-Net on the f16 MFMA, fp32-class in BOTH products:
//   qkv [N][T][heads][3][ch] (token-major output of the qkv 1x1 conv)  ->  out [N][T][heads][ch]
//   S^T = K Q^T   operands split THREE ways (hi + mid + lo, lo stored scaled by 2^22 so that it is not subnormal): six f16
//                 MFMAs per product, the four unscaled ones into one accumulator, the two with lo into another.  A two-way
//                 split carries 22 bits per operand; a score of sum|q k| ~ 9 then moves by ~5e-6 and a peaked softmax
//                 hands that to the output (3e-5 on N(0,1) inputs, ldm_ops.hip) — the logits need fp32 operands.
//   O^T = V^T P^T two-way split (three MFMAs): its error is relative to the output.
// K and V are split ONCE by a pre-pass into the LDS image of each 64-key block (K rows of 64 bytes with swizzled quarters,
// V^T rows padded to 160 bytes: conflict-free 16-byte fragment reads), which the main kernel streams with LDS-DMA, double buffered.  A workgroup is four
// waves x 32 queries (two 16-query tiles per wave share every K / V fragment read); online softmax per 64 keys; the S^T
// registers (lane (query, g): keys 16 kt + 4g + i) become the B operand of the second product after one split.
// Head widths up to 32 take one k-step (la_attention_kernel), 48 two (la_attention2_kernel, with a key split across
// workgroups where the grid would not fill the chip); wider heads (256 tokens and fewer) stay on the kernels of ldm_ops.hip.
#include "ldm_ops.h"

typedef _Float16 lh8 __attribute__((ext_vector_type(8)));
typedef _Float16 lh2 __attribute__((ext_vector_type(2)));
typedef float lf2 __attribute__((ext_vector_type(2)));
typedef unsigned lu4 __attribute__((ext_vector_type(4)));

#define LA_KB 64                 // keys per block
#define LA_KLD 32                // halfs per K row (64 B); its four 16-byte quarters are XOR-swizzled by (key >> 1) & 3:
                                 // conflict-free fragment reads without padding (conv.hip), 22 KiB per block image ->
                                 // three workgroups per CU
#define LA_KSWZ(key, q) ((((q) ^ (((key) >> 1) & 3))) * 8)   // half offset of quarter q in the row of `key`
#define LA_VLD 80                // halfs per V^T row: 64 + 16 pad = 160 B
#define LA_K_PART (LA_KB * LA_KLD)
template <int CH>
struct LaGeom {
    static constexpr int DT = (CH + 15) / 16;
    static constexpr bool SPARE = CH % 16 != 0;   // a free channel / row in the 32-wide K rows and the 16-row V^T tiles
    static constexpr int KS = (CH + 31) / 32;     // 32-channel k-steps of S^T: each is its own set of three K parts
    static_assert(!SPARE || KS == 1, "the spare channel lives in the single k-step of the narrow heads");
    static constexpr int V_PART = DT * 16 * LA_VLD;
    static constexpr int V_OFF = 3 * KS * LA_K_PART;
    static constexpr int RAW = V_OFF + 2 * V_PART;
    static constexpr int IMG = (RAW + 511) / 512 * 512;   // halfs per block image: whole 1 KiB DMA pieces
    static constexpr int PIECES = IMG / 512;
};

// x = h + m + l 2^-22 to 33 bits: h, m plain f16 (m is subnormal only for |x| < 0.125, where its 6e-8 granularity is an
// absolute error far below the logits' resolution), l stored scaled by 2^22 so that it is a normal number
__device__ __forceinline__ void la_split3(float x, _Float16& h, _Float16& m, _Float16& l) {
    h = (_Float16)x;
    const float r = x - (float)h;   // exact
    m = (_Float16)r;
    l = (_Float16)((r - (float)m) * 4194304.f);
}

// ---- pre-pass: one workgroup per 64-key block image: zero it (padding, tails), then one thread per (key, 8 channels)
//      writes the three K parts as 16-byte rows and scatters the two V^T parts into the key-slot order P^T is produced in
template <int CH>
__global__ __launch_bounds__(256) void la_pack_kernel(const float* __restrict__ qkv, _Float16* __restrict__ img, int T, int heads,
                                                      int nblk) {
    typedef LaGeom<CH> G;
    const long b = blockIdx.x;
    const int kb = (int)(b % nblk);
    const int hh = (int)((b / nblk) % heads), n = (int)(b / ((long)nblk * heads));
    const int C3 = heads * 3 * CH;
    const float scale = 1.f / sqrtf(sqrtf((float)CH));
    const float* base = qkv + (long)n * T * C3 + hh * 3 * CH;
    _Float16* dst = img + b * G::IMG;
    for (int i = threadIdx.x; i < G::IMG / 8; i += 256) *reinterpret_cast<lu4*>(dst + 8 * i) = lu4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (G::SPARE && threadIdx.x < LA_KB) {
        // spare channel CH of the K rows: 0 for a key, -30000 for a key slot beyond T (q carries 1 there: the score of a
        // missing key comes out of the MFMA as -30000, no masking instructions); spare row CH of V^T: all ones, so that
        // row CH of O^T accumulates sum(p) — the softmax denominator, rescaled with the rest
        if (kb * LA_KB + threadIdx.x >= T) dst[threadIdx.x * LA_KLD + LA_KSWZ(threadIdx.x, CH >> 3) + (CH & 7)] = (_Float16)(-30000.f);
        dst[G::V_OFF + CH * LA_VLD + threadIdx.x] = (_Float16)1.f;
    }
    constexpr int C8 = CH / 8;
    for (int i = threadIdx.x; i < LA_KB * C8; i += 256) {
        const int key = i / C8, c0 = (i % C8) * 8;
        const int kg = kb * LA_KB + key;
        if (kg >= T) continue;
        const float* row = base + (long)kg * C3;
        const f32x4 ka = ld4(row + CH + c0), kc = ld4(row + CH + c0 + 4);
        const f32x4 va = ld4(row + 2 * CH + c0), vc = ld4(row + 2 * CH + c0 + 4);
        lh8 h, m, l;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            _Float16 a, bb, c;
            la_split3((t < 4 ? ka[t] : kc[t - 4]) * scale, a, bb, c);
            h[t] = a;
            m[t] = bb;
            l[t] = c;
        }
        const int ko = (c0 >> 5) * 3 * LA_K_PART + key * LA_KLD + LA_KSWZ(key, (c0 >> 3) & 3);
        *reinterpret_cast<lh8*>(dst + ko) = h;
        *reinterpret_cast<lh8*>(dst + LA_K_PART + ko) = m;
        *reinterpret_cast<lh8*>(dst + 2 * LA_K_PART + ko) = l;
        // key = 16 kt + 4 g' + i  ->  slot 32 (kt >> 1) + 8 g' + 4 (kt & 1) + i   (the order P^T is produced in)
        const int kt = key >> 4, slot = 32 * (kt >> 1) + 8 * ((key >> 2) & 3) + 4 * (kt & 1) + (key & 3);
        _Float16* v0 = dst + G::V_OFF + c0 * LA_VLD + slot;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float x = t < 4 ? va[t] : vc[t - 4];
            const _Float16 xh = (_Float16)x;
            v0[t * LA_VLD] = xh;
            v0[G::V_PART + t * LA_VLD] = (_Float16)(x - (float)xh);
        }
    }
}

// hi/lo split of a pair: v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16 (see decode_attnq.hip)
__device__ __forceinline__ void la_split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(lf2{a, b}, lh2));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
}
__device__ __forceinline__ void la_swap32(float& x, float& y) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
__device__ __forceinline__ void la_swap16(float& x, float& y) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
__device__ __forceinline__ float la_colmax(float v) {   // max over the 4 lane groups g of one query column
    float x = v, y = v;
    la_swap32(x, y);
    x = fmaxf(x, y);
    y = x;
    la_swap16(x, y);
    return fmaxf(x, y);
}
__device__ __forceinline__ float la_colsum(float v) {
    float x = v, y = v;
    la_swap32(x, y);
    x += y;
    y = x;
    la_swap16(x, y);
    return x + y;
}
#define LA_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

// QT: 16-query tiles per wave (2: every fragment read feeds two tiles; 1: twice the workgroups, for grids that would not
// give every SIMD two waves otherwise — batch 1 at 4 096 tokens is 256 workgroups of 128 queries)
template <int CH, int QT>
__global__ __launch_bounds__(256, 3) void la_attention_kernel(const float* __restrict__ qkv, const _Float16* __restrict__ img,
                                                              float* __restrict__ out, int T, int heads, int nblk) {
    typedef LaGeom<CH> G;
    constexpr int DT = G::DT;
    // two distinct LDS objects: a read of one is not guarded against the LDS-DMA refill of the other (decode_f16.hip)
    __shared__ __attribute__((aligned(16))) _Float16 s_b0[G::IMG], s_b1[G::IMG];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    constexpr int QW = 64 * QT;   // queries per workgroup
    const int qblocks = (T + QW - 1) / QW;
    const int qb = blockIdx.x % qblocks;
    const int hh = (blockIdx.x / qblocks) % heads, n = blockIdx.x / (qblocks * heads);
    const int C3 = heads * 3 * CH;
    const float* base = qkv + (long)n * T * C3 + hh * 3 * CH;
    const _Float16* gimg = img + ((long)(n * heads + hh) * nblk) * G::IMG;
    const float qscale = 1.4426950408889634f / sqrtf(sqrtf((float)CH));   // the scores come out in log2 units

    auto dma_block = [&](int kb, _Float16* buf) {
        const _Float16* src = gimg + (long)kb * G::IMG;
#pragma unroll
        for (int i = 0; i < (G::PIECES + 3) / 4; ++i) {
            const int piece = wave + 4 * i;
            if (piece < G::PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 512 + lane * 8),
                                                 (__attribute__((address_space(3))) void*)(buf + piece * 512), 16, 0, 0);
        }
    };
    dma_block(0, s_b0);

    // the wave's 2 x 16 queries: q (scaled) split three ways, k-slot 8g + t <-> channel 8g + t
    lh8 qh[QT], qm[QT], ql[QT];
    int qrow[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = qb * QW + wave * (16 * QT) + qt * 16 + m;
        qrow[qt] = q;
        const int qc = q < T ? q : T - 1;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int c = 8 * g + t;
            const float v = c < CH ? base[(long)qc * C3 + c] * qscale : (G::SPARE && c == CH) ? 1.f : 0.f;
            _Float16 h, md, l;
            la_split3(v, h, md, l);
            qh[qt][t] = h;
            qm[qt][t] = md;
            ql[qt][t] = l;
        }
    }
    f32x4 acc[DT][QT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc[d][qt] = zero4();
    float mx[QT], den[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        mx[qt] = -1e30f;
        den[qt] = 0.f;
    }   // den: this lane's keys only, reduced over g at the end
    dma_publish_barrier();

    // Fragment reads and their counted waits are issued by hand (decode_f16.hip): the compiler's own schedule reads the
    // fragments of two key tiles, waits lgkmcnt(0), multiplies, and only then reads the next two — the LDS latency is
    // exposed five times per block.  Here the block's 12 K fragments are requested up front, the V^T fragments under the
    // S^T products and the softmax.
#define LA_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define LA_WAIT6(n, a, b, c, d, e, f) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(n))
#define LA_WAIT4(n, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(n))
    auto compute = [&](const _Float16* buf, int k0) {
        const bool partial = k0 + LA_KB > T;
        const unsigned lk = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(const_cast<_Float16*>(buf) + m * LA_KLD + LA_KSWZ(m, g));   // key tile kt adds 16 rows: same swizzle
        const unsigned lv = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(const_cast<_Float16*>(buf) + 3 * LA_K_PART + m * LA_VLD + 8 * g);
        lh8 kf[4][3];        // [key tile][hi, mid, lo]
        lh8 vf[2][DT][2];    // [k-step][dim tile][hi, lo]
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int pt = 0; pt < 3; ++pt) LA_RD(kf[kt][pt], lk, (kt * 16 * LA_KLD + pt * LA_K_PART) * 2);
        // ---- S^T for the 4 key tiles x QT query tiles: six products, the unscaled four into one accumulator ----
        f32x4 s[QT][4];
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            if (kp == 0) LA_WAIT6(6, kf[0][0], kf[0][1], kf[0][2], kf[1][0], kf[1][1], kf[1][2]);
            else LA_WAIT6(2 * DT, kf[2][0], kf[2][1], kf[2][2], kf[3][0], kf[3][1], kf[3][2]);   // the V^T reads of k-step 0 stay in flight
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int kt = 2 * kp + k2;
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    f32x4 a2 = LA_MFMA(kf[kt][0], ql[qt], zero4());
                    a2 = LA_MFMA(kf[kt][2], qh[qt], a2);
                    f32x4 a0 = LA_MFMA(kf[kt][1], qm[qt], zero4());
                    a0 = LA_MFMA(kf[kt][0], qm[qt], a0);
                    a0 = LA_MFMA(kf[kt][1], qh[qt], a0);
                    a0 = LA_MFMA(kf[kt][0], qh[qt], a0);
                    s[qt][kt] = a0 + a2 * (1.f / 4194304.f);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // V^T fragments of k-step kp: requested here, consumed after the softmax
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) LA_RD(vf[kp][d][pt], lv, (16 * d * LA_VLD + 32 * kp + pt * G::V_PART) * 2);
        }
        // ---- online softmax over the block's 64 keys; P is produced scaled by 2^14 (exponent bias, removed with 1/den at
        //      the end): small probabilities would otherwise sit in f16's subnormal range and lose their low half ----
        lh8 ph[QT][2], pl[QT][2];   // [query tile][k-step of the P V product]
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (!G::SPARE && partial) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (k0 + kt * 16 + 4 * g + i >= T) s[qt][kt][i] = -1e30f;
            }
            float bmax = fmaxf(fmaxf(s[qt][0][0], s[qt][0][1]), fmaxf(s[qt][0][2], s[qt][0][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt)
                bmax = fmaxf(bmax, fmaxf(fmaxf(s[qt][kt][0], s[qt][kt][1]), fmaxf(s[qt][kt][2], s[qt][kt][3])));
            bmax = la_colmax(bmax);
            const float mnew = fmaxf(mx[qt], bmax);
            const float corr = __builtin_amdgcn_exp2f(mx[qt] - mnew);
            mx[qt] = mnew;
            const float bias = 14.f - mnew;
            float bsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s[qt][kt][i] = __builtin_amdgcn_exp2f(s[qt][kt][i] + bias);   // missing keys: exp2(-30000) = 0
                    if (!G::SPARE) bsum += s[qt][kt][i];
                }
            if (!G::SPARE) den[qt] = den[qt] * corr + bsum;
#pragma unroll
            for (int d = 0; d < DT; ++d) acc[d][qt] *= corr;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                la_split2(s[qt][2 * kk][0], s[qt][2 * kk][1], h0, l0);
                la_split2(s[qt][2 * kk][2], s[qt][2 * kk][3], h1, l1);
                la_split2(s[qt][2 * kk + 1][0], s[qt][2 * kk + 1][1], h2, l2);
                la_split2(s[qt][2 * kk + 1][2], s[qt][2 * kk + 1][3], h3, l3);
                ph[qt][kk] = __builtin_bit_cast(lh8, lu4{h0, h1, h2, h3});
                pl[qt][kk] = __builtin_bit_cast(lh8, lu4{l0, l1, l2, l3});
            }
        }
        // partial-register asm writes -> MFMA reads: pad (decode_attnq.hip, AQ_SETTLE)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (DT == 2) {
                if (kk == 0) LA_WAIT4(4, vf[0][0][0], vf[0][0][1], vf[0][1][0], vf[0][1][1]);
                else LA_WAIT4(0, vf[1][0][0], vf[1][0][1], vf[1][1][0], vf[1][1][1]);
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0][0][0]), "+v"(vf[0][0][1]), "+v"(vf[1][0][0]), "+v"(vf[1][0][1]));
            }
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    f32x4 o = LA_MFMA(vf[kk][d][0], pl[qt][kk], acc[d][qt]);
                    o = LA_MFMA(vf[kk][d][1], ph[qt][kk], o);
                    acc[d][qt] = LA_MFMA(vf[kk][d][0], ph[qt][kk], o);
                }
        }
    };

    for (int kb = 0; kb < nblk; kb += 2) {
        if (kb + 1 < nblk) dma_block(kb + 1, s_b1);
        compute(s_b0, kb * LA_KB);
        dma_publish_barrier();
        if (kb + 1 < nblk) {
            if (kb + 2 < nblk) dma_block(kb + 2, s_b0);
            compute(s_b1, (kb + 1) * LA_KB);
            dma_publish_barrier();
        }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        // den carries the same 2^14 as P.  With a spare V^T row it is row CH of O^T: element CH % 4 of the last tile's
        // accumulator in lane group (CH % 16) / 4; the other groups hold 0 there after the select
        const float dsel = G::SPARE ? (g == (CH % 16) / 4 ? acc[DT - 1][qt][CH % 4] : 0.f) : den[qt];
        const float inv = 1.f / la_colsum(dsel);
        if (qrow[qt] < T) {
            float* o = out + ((long)n * T + qrow[qt]) * (heads * CH) + hh * CH;
#pragma unroll
            for (int d = 0; d < DT; ++d)
                if (d * 16 + 4 * g + 3 < CH) st4(o + d * 16 + 4 * g, acc[d][qt] * inv);   // CH % 4 == 0: the spare row is never stored
        }
    }
}

// ---- head widths 33..64 (the 48-wide heads of the 32 x 32 level: 1 024 tokens at the 64 x 64 latent, 4 096 at 128 x 128):
//      S^T takes two 32-channel k-steps (channels beyond CH are zeros in both operands), O^T has DT = CH / 16 row tiles.
//      The K fragments move through a two-slot ring, one key tile (2 k-steps x 3 parts) ahead of the products; the V^T
//      fragments are requested under the last two key tiles.  78 KB of LDS for the double-buffered block image: two
//      workgroups per CU.  KEY SPLIT: at 1 024 tokens and batch 1 a head has 16 query blocks x 8 heads = 128 workgroups for
//      256 CUs, so gridDim.y workgroups share a query block, each taking a contiguous range of key blocks and leaving its
//      un-normalised O^T, running maximum and denominator in the workspace; la_merge_kernel adds them up
//      (flash-decoding's reduction; one more ~3 us launch instead of half the chip idle for ~30 us).
template <int CH, int QT>
__global__ __launch_bounds__(256, 2) void la_attention2_kernel(const float* __restrict__ qkv, const _Float16* __restrict__ img,
                                                               float* __restrict__ out, float* __restrict__ part_o,
                                                               float* __restrict__ part_md, int T, int heads, int nblk) {
    typedef LaGeom<CH> G;
    constexpr int DT = G::DT, KS = G::KS;
    static_assert(KS == 2 && !G::SPARE && DT * 2 == 6, "built for head widths 48 (and 33..48 padded): six V^T fragments per k-step");
    __shared__ __attribute__((aligned(16))) _Float16 s_b0[G::IMG], s_b1[G::IMG];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    constexpr int QW = 64 * QT;
    const int qblocks = (T + QW - 1) / QW;
    const int qb = blockIdx.x % qblocks;
    const int hh = (blockIdx.x / qblocks) % heads, n = blockIdx.x / (qblocks * heads);
    const int nsplit = gridDim.y, sp = blockIdx.y;
    const int kb_lo = (int)((long)sp * nblk / nsplit), kb_hi = (int)((long)(sp + 1) * nblk / nsplit);
    const int C3 = heads * 3 * CH;
    const float* base = qkv + (long)n * T * C3 + hh * 3 * CH;
    const _Float16* gimg = img + ((long)(n * heads + hh) * nblk) * G::IMG;
    const float qscale = 1.4426950408889634f / sqrtf(sqrtf((float)CH));

    auto dma_block = [&](int kb, _Float16* buf) {
        const _Float16* src = gimg + (long)kb * G::IMG;
#pragma unroll
        for (int i = 0; i < (G::PIECES + 3) / 4; ++i) {
            const int piece = wave + 4 * i;
            if (piece < G::PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 512 + lane * 8),
                                                 (__attribute__((address_space(3))) void*)(buf + piece * 512), 16, 0, 0);
        }
    };
    dma_block(kb_lo, s_b0);

    lh8 qh[QT][KS], qm[QT][KS], ql[QT][KS];
    int qrow[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = qb * QW + wave * (16 * QT) + qt * 16 + m;
        qrow[qt] = q;
        const int qc = q < T ? q : T - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int c = 32 * ks + 8 * g + t;
                const float v = c < CH ? base[(long)qc * C3 + c] * qscale : 0.f;
                _Float16 h, md, l;
                la_split3(v, h, md, l);
                qh[qt][ks][t] = h;
                qm[qt][ks][t] = md;
                ql[qt][ks][t] = l;
            }
    }
    f32x4 acc[DT][QT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc[d][qt] = zero4();
    float mx[QT], den[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        mx[qt] = -1e30f;
        den[qt] = 0.f;
    }
    dma_publish_barrier();

    auto compute = [&](const _Float16* buf, int k0) {
        const bool partial = k0 + LA_KB > T;
        const unsigned lk = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(const_cast<_Float16*>(buf) + m * LA_KLD + LA_KSWZ(m, g));
        const unsigned lv = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(const_cast<_Float16*>(buf) + G::V_OFF + m * LA_VLD + 8 * g);
        lh8 kf[2][KS][3];    // [ring slot][k-step][hi, mid, lo]
        lh8 vf[2][DT][2];    // [k-step of P V][dim tile][hi, lo]
#define LA_RDK(slot, kt)                                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) _Pragma("unroll") for (int pt = 0; pt < 3; ++pt)                \
        LA_RD(kf[slot][ks][pt], lk, ((kt) * 16 * LA_KLD + (3 * ks + pt) * LA_K_PART) * 2)
#define LA_RDV(kk)                                                                                                    \
    _Pragma("unroll") for (int d = 0; d < DT; ++d) _Pragma("unroll") for (int pt = 0; pt < 2; ++pt)                   \
        LA_RD(vf[kk][d][pt], lv, (16 * d * LA_VLD + 32 * (kk) + pt * G::V_PART) * 2)
        LA_RDK(0, 0);
        LA_RDK(1, 1);
        f32x4 s[QT][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int sl = kt & 1;
            // twelve reads in flight at most (the counter holds 15): the oldest six are this key tile's
            LA_WAIT6(6, kf[sl][0][0], kf[sl][0][1], kf[sl][0][2], kf[sl][1][0], kf[sl][1][1], kf[sl][1][2]);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                f32x4 a2 = LA_MFMA(kf[sl][0][0], ql[qt][0], zero4());
                a2 = LA_MFMA(kf[sl][1][0], ql[qt][1], a2);
                a2 = LA_MFMA(kf[sl][0][2], qh[qt][0], a2);
                a2 = LA_MFMA(kf[sl][1][2], qh[qt][1], a2);
                f32x4 a0 = LA_MFMA(kf[sl][0][1], qm[qt][0], zero4());
                a0 = LA_MFMA(kf[sl][1][1], qm[qt][1], a0);
                a0 = LA_MFMA(kf[sl][0][0], qm[qt][0], a0);
                a0 = LA_MFMA(kf[sl][1][0], qm[qt][1], a0);
                a0 = LA_MFMA(kf[sl][0][1], qh[qt][0], a0);
                a0 = LA_MFMA(kf[sl][1][1], qh[qt][1], a0);
                a0 = LA_MFMA(kf[sl][0][0], qh[qt][0], a0);
                a0 = LA_MFMA(kf[sl][1][0], qh[qt][1], a0);
                s[qt][kt] = a0 + a2 * (1.f / 4194304.f);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kt == 0) { LA_RDK(0, 2); }
            else if (kt == 1) { LA_RDK(1, 3); }
            else if (kt == 2) { LA_RDV(0); }
            else { LA_RDV(1); }
        }
        lh8 ph[QT][2], pl[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (partial) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (k0 + kt * 16 + 4 * g + i >= T) s[qt][kt][i] = -1e30f;
            }
            float bmax = fmaxf(fmaxf(s[qt][0][0], s[qt][0][1]), fmaxf(s[qt][0][2], s[qt][0][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt)
                bmax = fmaxf(bmax, fmaxf(fmaxf(s[qt][kt][0], s[qt][kt][1]), fmaxf(s[qt][kt][2], s[qt][kt][3])));
            bmax = la_colmax(bmax);
            const float mnew = fmaxf(mx[qt], bmax);
            const float corr = __builtin_amdgcn_exp2f(mx[qt] - mnew);
            mx[qt] = mnew;
            const float bias = 14.f - mnew;
            float bsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s[qt][kt][i] = __builtin_amdgcn_exp2f(s[qt][kt][i] + bias);
                    bsum += s[qt][kt][i];
                }
            den[qt] = den[qt] * corr + bsum;
#pragma unroll
            for (int d = 0; d < DT; ++d) acc[d][qt] *= corr;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                la_split2(s[qt][2 * kk][0], s[qt][2 * kk][1], h0, l0);
                la_split2(s[qt][2 * kk][2], s[qt][2 * kk][3], h1, l1);
                la_split2(s[qt][2 * kk + 1][0], s[qt][2 * kk + 1][1], h2, l2);
                la_split2(s[qt][2 * kk + 1][2], s[qt][2 * kk + 1][3], h3, l3);
                ph[qt][kk] = __builtin_bit_cast(lh8, lu4{h0, h1, h2, h3});
                pl[qt][kk] = __builtin_bit_cast(lh8, lu4{l0, l1, l2, l3});
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 0) LA_WAIT6(6, vf[0][0][0], vf[0][0][1], vf[0][1][0], vf[0][1][1], vf[0][2][0], vf[0][2][1]);
            else LA_WAIT6(0, vf[1][0][0], vf[1][0][1], vf[1][1][0], vf[1][1][1], vf[1][2][0], vf[1][2][1]);
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    f32x4 o = LA_MFMA(vf[kk][d][0], pl[qt][kk], acc[d][qt]);
                    o = LA_MFMA(vf[kk][d][1], ph[qt][kk], o);
                    acc[d][qt] = LA_MFMA(vf[kk][d][0], ph[qt][kk], o);
                }
        }
#undef LA_RDK
#undef LA_RDV
    };

    for (int kb = kb_lo; kb < kb_hi; kb += 2) {
        if (kb + 1 < kb_hi) dma_block(kb + 1, s_b1);
        compute(s_b0, kb * LA_KB);
        dma_publish_barrier();
        if (kb + 1 < kb_hi) {
            if (kb + 2 < kb_hi) dma_block(kb + 2, s_b0);
            compute(s_b1, (kb + 1) * LA_KB);
            dma_publish_barrier();
        }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float dsum = la_colsum(den[qt]);   // den and acc carry the same 2^14
        if (qrow[qt] >= T) continue;
        if (nsplit == 1) {
            const float inv = 1.f / dsum;
            float* o = out + ((long)n * T + qrow[qt]) * (heads * CH) + hh * CH;
#pragma unroll
            for (int d = 0; d < DT; ++d)
                if (d * 16 + 4 * g + 3 < CH) st4(o + d * 16 + 4 * g, acc[d][qt] * inv);
        } else {
            const long row = ((long)(sp * gridDim.x / (qblocks * heads) + n) * T + qrow[qt]) * heads + hh;   // [split][n][q][head]
            float* o = part_o + row * CH;
#pragma unroll
            for (int d = 0; d < DT; ++d)
                if (d * 16 + 4 * g + 3 < CH) st4(o + d * 16 + 4 * g, acc[d][qt]);
            if (g == 0) {
                part_md[2 * row] = mx[qt];
                part_md[2 * row + 1] = dsum;
            }
        }
    }
}

// out[r][c] = sum_s O_s[r][c] 2^(m_s - M) / sum_s d_s 2^(m_s - M),  M = max_s m_s   (the scores are in log2 units);
// rows = N * T * heads, one thread per four channels
template <int CH>
__global__ __launch_bounds__(256) void la_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_md,
                                                       float* __restrict__ out, long rows, int nsplit) {
    constexpr int C4 = CH / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C4) return;
    const long r = i / C4;
    const int c = (int)(i % C4) * 4;
    float M = -1e30f;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_md[2 * (s * rows + r)]);
    f32x4 o = zero4();
    float d = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float w = __builtin_amdgcn_exp2f(part_md[2 * (s * rows + r)] - M);
        d += w * part_md[2 * (s * rows + r) + 1];
        o += ld4(part_o + (s * rows + r) * CH + c) * w;
    }
    st4(out + r * CH + c, o * (1.f / d));
}

#undef LA_RD
#undef LA_WAIT6
#undef LA_WAIT4

// key splits of the wide-head kernel: enough workgroups for two per CU, at least two key blocks per split (measured at
// 1 x 1 024 x 8 heads: 2 / 4 / 8 splits 27.2 / 27.8 / 32.4 us per call against 44.7 us for the fp32-MFMA kernel;
// profiles/r06_ldm_attn48.md)
static int la_splits(int N, int T, int heads) {
    const int nblk = (T + LA_KB - 1) / LA_KB;
    const long base = (long)N * heads * ((T + 63) / 64);
    int s = 1;
    while (base * s < 512 && 2 * s * 2 <= nblk && s < 8) s *= 2;
    return s;
}
static size_t la_align256(size_t b) { return (b + 255) / 256 * 256; }

size_t qkv_attention_ws_bytes(int N, int T, int heads, int ch) {
    if (ch % 8) return 0;
    const int nblk = (T + LA_KB - 1) / LA_KB;
    if (ch == 48) {
        const int ns = la_splits(N, T, heads);
        size_t b = la_align256((size_t)N * heads * nblk * LaGeom<48>::IMG * sizeof(_Float16));
        if (ns > 1) b += la_align256((size_t)ns * N * T * heads * ch * sizeof(float)) + la_align256((size_t)ns * N * T * heads * 2 * sizeof(float));
        return b;
    }
    if (ch > 32) return 0;
    const size_t img = ch <= 16 ? LaGeom<16>::IMG : LaGeom<32>::IMG;   // DT = 1 or 2
    return (size_t)N * heads * nblk * img * sizeof(_Float16);
}

int launch_qkv_attention_ws(const float* qkv, float* out, int N, int T, int heads, int ch, void* ws, size_t ws_bytes,
                            hipStream_t stream) {
    S3D_CHECK_ARG(N >= 1 && T >= 1 && heads >= 1, "qkv_attention_ws: bad dims");
    S3D_CHECK_ARG(ws && ws_bytes >= qkv_attention_ws_bytes(N, T, heads, ch) && qkv_attention_ws_bytes(N, T, heads, ch) > 0,
                  "qkv_attention_ws: head width %d / workspace %zu", ch, ws_bytes);
    const int nblk = (T + LA_KB - 1) / LA_KB;
    if (ch == 48) {
        const int ns = la_splits(N, T, heads);
        char* p = (char*)ws;
        _Float16* img = (_Float16*)p;
        p += la_align256((size_t)N * heads * nblk * LaGeom<48>::IMG * sizeof(_Float16));
        float* part_o = (float*)p;
        p += la_align256((size_t)ns * N * T * heads * ch * sizeof(float));
        float* part_md = (float*)p;
        hipLaunchKernelGGL((la_pack_kernel<48>), dim3((unsigned)(N * heads * nblk)), dim3(256), 0, stream, qkv, img, T, heads, nblk);
        S3D_LAUNCH_CHECK();
        const int blocks2 = N * heads * ((T + 127) / 128);
        if (blocks2 >= 512)   // two query tiles per wave share every fragment read (ns == 1 here)
            hipLaunchKernelGGL((la_attention2_kernel<48, 2>), dim3((unsigned)blocks2, 1), dim3(256), 0, stream, qkv,
                               (const _Float16*)img, out, part_o, part_md, T, heads, nblk);
        else
            hipLaunchKernelGGL((la_attention2_kernel<48, 1>), dim3((unsigned)(N * heads * ((T + 63) / 64)), ns), dim3(256), 0, stream,
                               qkv, (const _Float16*)img, out, part_o, part_md, T, heads, nblk);
        S3D_LAUNCH_CHECK();
        if (ns > 1) {
            const long rows = (long)N * T * heads;
            hipLaunchKernelGGL((la_merge_kernel<48>), dim3((unsigned)((rows * 12 + 255) / 256)), dim3(256), 0, stream,
                               (const float*)part_o, (const float*)part_md, out, rows, ns);
            S3D_LAUNCH_CHECK();
        }
        return 0;
    }
#define LA_CASE(c)                                                                                                         \
    if (ch == c) {                                                                                                         \
        hipLaunchKernelGGL((la_pack_kernel<c>), dim3((unsigned)(N * heads * nblk)), dim3(256), 0, stream, qkv, (_Float16*)ws, T, \
                           heads, nblk);                                                                                   \
        S3D_LAUNCH_CHECK();                                                                                                \
        const int blocks2 = N * heads * ((T + 127) / 128);                                                                 \
        if (blocks2 >= 512)                                                                                                \
            hipLaunchKernelGGL((la_attention_kernel<c, 2>), dim3(blocks2), dim3(256), 0, stream, qkv, (const _Float16*)ws, \
                               out, T, heads, nblk);                                                                       \
        else                                                                                                               \
            hipLaunchKernelGGL((la_attention_kernel<c, 1>), dim3(N * heads * ((T + 63) / 64)), dim3(256), 0, stream, qkv,  \
                               (const _Float16*)ws, out, T, heads, nblk);                                                  \
        S3D_LAUNCH_CHECK();                                                                                                \
        return 0;                                                                                                          \
    }
    LA_CASE(8) LA_CASE(16) LA_CASE(24) LA_CASE(32)
#undef LA_CASE
    s3d_set_error("qkv_attention_ws: head width %d not built (8, 16, 24, 32, 48)", ch);
    return S3D_E_ARG;
}
