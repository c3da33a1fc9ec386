// ldm_attn.hip — QKVAttentionLegacy (openaimodel.py:353-381) for the long-sequence attention blocks of the gen_slices
// U-Net on the f16 MFMA, fp32-class in BOTH products:
//   qkv [N][T][heads][3][ch] (token-major output of the qkv 1x1 conv)  ->  out [N][T][heads][ch]
//   S^T = K Q^T   operands split THREE ways (hi + mid 2^-11 + lo 2^-22, the lower parts stored scaled so that none is
//                 subnormal): six f16 MFMAs per product into three accumulators, recombined once per tile.  A two-way
//                 split carries 22 bits per operand; a score of sum|q k| ~ 9 then moves by ~5e-6 and a peaked softmax
//                 hands that to the output (3e-5 on N(0,1) inputs, ldm_ops.hip) — the logits need fp32 operands.
//   O^T = V^T P^T two-way split (three MFMAs): its error is relative to the output.
// K and V are split ONCE by a pre-pass into the LDS image of each 64-key block (rows padded to 96 / 160 bytes:
// conflict-free 16-byte fragment reads), which the main kernel streams with LDS-DMA, double buffered.  A workgroup is four
// waves x 32 queries (two 16-query tiles per wave share every K / V fragment read); online softmax per 32 keys; the S^T
// registers (lane (query, g): keys 16 kt + 4g + i) become the B operand of the second product after one split.
// Head widths up to 32 (one k-step); wider heads (1 024 tokens and fewer) stay on the kernels of ldm_ops.hip.
#include "ldm_ops.h"

typedef _Float16 lh8 __attribute__((ext_vector_type(8)));
typedef _Float16 lh2 __attribute__((ext_vector_type(2)));
typedef float lf2 __attribute__((ext_vector_type(2)));
typedef unsigned lu4 __attribute__((ext_vector_type(4)));

#define LA_KB 64                 // keys per block
#define LA_KLD 48                // halfs per K row: 32 + 16 pad = 96 B
#define LA_VLD 80                // halfs per V^T row: 64 + 16 pad = 160 B
#define LA_K_PART (LA_KB * LA_KLD)
template <int CH>
struct LaGeom {
    static constexpr int DT = (CH + 15) / 16;
    static constexpr int V_PART = DT * 16 * LA_VLD;
    static constexpr int RAW = 3 * LA_K_PART + 2 * V_PART;
    static constexpr int IMG = (RAW + 511) / 512 * 512;   // halfs per block image: whole 1 KiB DMA pieces
    static constexpr int PIECES = IMG / 512;
};

__device__ __forceinline__ void la_split3(float x, _Float16& h, _Float16& m, _Float16& l) {
    h = (_Float16)x;
    const float r = (x - (float)h) * 2048.f;   // exact: the remainder of an f16 rounding, scaled by a power of two
    m = (_Float16)r;
    l = (_Float16)((r - (float)m) * 2048.f);
}

// ---- pre-pass: one thread per half of the block images (gather form: padding and tails come out as zeros) ----
template <int CH>
__global__ void la_pack_kernel(const float* __restrict__ qkv, _Float16* __restrict__ img, int T, int heads, int nblk,
                               long total) {
    typedef LaGeom<CH> G;
    const int C3 = heads * 3 * CH;
    const float scale = 1.f / sqrtf(sqrtf((float)CH));
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int e = (int)(idx % G::IMG);
        const long b = idx / G::IMG;
        const int kb = (int)(b % nblk);
        const int hh = (int)((b / nblk) % heads), n = (int)(b / ((long)nblk * heads));
        const float* base = qkv + (long)n * T * C3 + hh * 3 * CH;
        _Float16 v = (_Float16)0.f;
        if (e < 3 * LA_K_PART) {
            const int part = e / LA_K_PART, r = e % LA_K_PART;
            const int key = r / LA_KLD, c = r % LA_KLD;
            const int kg = kb * LA_KB + key;
            if (c < CH && kg < T) {
                _Float16 h, m, l;
                la_split3(base[(long)kg * C3 + CH + c] * scale, h, m, l);
                v = part == 0 ? h : part == 1 ? m : l;
            }
        } else if (e < G::RAW) {
            const int r0 = e - 3 * LA_K_PART;
            const int part = r0 / G::V_PART, r = r0 % G::V_PART;
            const int d = r / LA_VLD, slot = r % LA_VLD;
            if (d < CH && slot < LA_KB) {
                // slot 32 (kt >> 1) + 8 g' + 4 (kt & 1) + i  <-  key 16 kt + 4 g' + i   (the order P^T is produced in)
                const int kk = slot >> 5, g2 = (slot >> 3) & 3, kt = 2 * kk + ((slot >> 2) & 1), i = slot & 3;
                const int kg = kb * LA_KB + 16 * kt + 4 * g2 + i;
                if (kg < T) {
                    const float x = base[(long)kg * C3 + 2 * CH + d];
                    const _Float16 h = (_Float16)x;
                    v = part == 0 ? h : (_Float16)(x - (float)h);
                }
            }
        }
        img[idx] = v;
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

template <int CH>
__global__ __launch_bounds__(256, 2) void la_attention_kernel(const float* __restrict__ qkv, const _Float16* __restrict__ img,
                                                              float* __restrict__ out, int T, int heads, int nblk) {
    typedef LaGeom<CH> G;
    constexpr int DT = G::DT;
    // two distinct LDS objects: a read of one is not guarded against the LDS-DMA refill of the other (decode_f16.hip)
    __shared__ __attribute__((aligned(16))) _Float16 s_b0[G::IMG], s_b1[G::IMG];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int qblocks = (T + 127) / 128;
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
    lh8 qh[2], qm[2], ql[2];
    int qrow[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = qb * 128 + wave * 32 + qt * 16 + m;
        qrow[qt] = q;
        const int qc = q < T ? q : T - 1;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int c = 8 * g + t;
            const float v = c < CH ? base[(long)qc * C3 + c] * qscale : 0.f;
            _Float16 h, md, l;
            la_split3(v, h, md, l);
            qh[qt][t] = h;
            qm[qt][t] = md;
            ql[qt][t] = l;
        }
    }
    f32x4 acc[DT][2];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) acc[d][qt] = zero4();
    float mx[2] = {-1e30f, -1e30f}, den[2] = {0.f, 0.f};   // den: this lane's keys only, reduced over g at the end
    dma_publish_barrier();

    auto compute = [&](const _Float16* buf, int k0) {
        const bool partial = k0 + LA_KB > T;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x4 s[2][2];   // [query tile][key tile of the pair]
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int ro = ((2 * kk + k2) * 16 + m) * LA_KLD + 8 * g;
                const lh8 kh = *reinterpret_cast<const lh8*>(buf + ro);
                const lh8 km = *reinterpret_cast<const lh8*>(buf + LA_K_PART + ro);
                const lh8 kl = *reinterpret_cast<const lh8*>(buf + 2 * LA_K_PART + ro);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    f32x4 a2 = LA_MFMA(kh, ql[qt], zero4());
                    a2 = LA_MFMA(km, qm[qt], a2);
                    a2 = LA_MFMA(kl, qh[qt], a2);
                    f32x4 a1 = LA_MFMA(kh, qm[qt], zero4());
                    a1 = LA_MFMA(km, qh[qt], a1);
                    const f32x4 a0 = LA_MFMA(kh, qh[qt], zero4());
                    s[qt][k2] = a0 + (a1 + a2 * (1.f / 2048.f)) * (1.f / 2048.f);
                }
            }
            lh8 ph[2], pl[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                float e[8];
                float bmax = -1e30f;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = s[qt][k2][i];
                        if (partial && k0 + (2 * kk + k2) * 16 + 4 * g + i >= T) v = -1e30f;
                        e[4 * k2 + i] = v;
                        bmax = fmaxf(bmax, v);
                    }
                bmax = la_colmax(bmax);
                const float mnew = fmaxf(mx[qt], bmax);
                const float corr = __builtin_amdgcn_exp2f(mx[qt] - mnew);
                mx[qt] = mnew;
                float bsum = 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    e[t] = __builtin_amdgcn_exp2f(e[t] - mnew);   // masked keys: exp2(-1e30 - m) = 0
                    bsum += e[t];
                }
                den[qt] = den[qt] * corr + bsum;
#pragma unroll
                for (int d = 0; d < DT; ++d) acc[d][qt] *= corr;
                // probabilities are split after a 2^14 scale (removed with 1/den at the end): small ones would otherwise
                // sit in f16's subnormal range and lose their low half
                unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                la_split2(e[0] * 16384.f, e[1] * 16384.f, h0, l0);
                la_split2(e[2] * 16384.f, e[3] * 16384.f, h1, l1);
                la_split2(e[4] * 16384.f, e[5] * 16384.f, h2, l2);
                la_split2(e[6] * 16384.f, e[7] * 16384.f, h3, l3);
                ph[qt] = __builtin_bit_cast(lh8, lu4{h0, h1, h2, h3});
                pl[qt] = __builtin_bit_cast(lh8, lu4{l0, l1, l2, l3});
            }
            // partial-register asm writes -> MFMA reads: pad (decode_attnq.hip, AQ_SETTLE)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int vo = 3 * LA_K_PART + (16 * d + m) * LA_VLD + 32 * kk + 8 * g;
                const lh8 vh = *reinterpret_cast<const lh8*>(buf + vo);
                const lh8 vl = *reinterpret_cast<const lh8*>(buf + G::V_PART + vo);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    f32x4 o = LA_MFMA(vh, pl[qt], acc[d][qt]);
                    o = LA_MFMA(vl, ph[qt], o);
                    acc[d][qt] = LA_MFMA(vh, ph[qt], o);
                }
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
    for (int qt = 0; qt < 2; ++qt) {
        const float inv = (1.f / 16384.f) / la_colsum(den[qt]);
        if (qrow[qt] < T) {
            float* o = out + ((long)n * T + qrow[qt]) * (heads * CH) + hh * CH;
#pragma unroll
            for (int d = 0; d < DT; ++d)
                if (d * 16 + 4 * g + 3 < CH) st4(o + d * 16 + 4 * g, acc[d][qt] * inv);
        }
    }
}

size_t qkv_attention_ws_bytes(int N, int T, int heads, int ch) {
    if (ch > 32 || ch % 8) return 0;
    const int nblk = (T + LA_KB - 1) / LA_KB;
    const size_t img = ch <= 16 ? LaGeom<16>::IMG : LaGeom<32>::IMG;   // DT = 1 or 2
    return (size_t)N * heads * nblk * img * sizeof(_Float16);
}

int launch_qkv_attention_ws(const float* qkv, float* out, int N, int T, int heads, int ch, void* ws, size_t ws_bytes,
                            hipStream_t stream) {
    S3D_CHECK_ARG(N >= 1 && T >= 1 && heads >= 1, "qkv_attention_ws: bad dims");
    S3D_CHECK_ARG(ws && ws_bytes >= qkv_attention_ws_bytes(N, T, heads, ch) && qkv_attention_ws_bytes(N, T, heads, ch) > 0,
                  "qkv_attention_ws: head width %d / workspace %zu", ch, ws_bytes);
    const int nblk = (T + LA_KB - 1) / LA_KB;
#define LA_CASE(c)                                                                                                         \
    if (ch == c) {                                                                                                         \
        const long total = (long)N * heads * nblk * LaGeom<c>::IMG;                                                        \
        const int pb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);                                     \
        hipLaunchKernelGGL((la_pack_kernel<c>), dim3(pb), dim3(256), 0, stream, qkv, (_Float16*)ws, T, heads, nblk, total); \
        S3D_LAUNCH_CHECK();                                                                                                \
        const int blocks = N * heads * ((T + 127) / 128);                                                                  \
        hipLaunchKernelGGL((la_attention_kernel<c>), dim3(blocks), dim3(256), 0, stream, qkv, (const _Float16*)ws, out, T, \
                           heads, nblk);                                                                                   \
        S3D_LAUNCH_CHECK();                                                                                                \
        return 0;                                                                                                          \
    }
    LA_CASE(8) LA_CASE(16) LA_CASE(24) LA_CASE(32)
#undef LA_CASE
    s3d_set_error("qkv_attention_ws: head width %d not built (8, 16, 24, 32)", ch);
    return S3D_E_ARG;
}
