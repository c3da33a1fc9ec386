// ldm_ops.hip — primitives of the latent-diffusion denoising U-Net (gen_slices/ldm/modules/diffusionmodules/
// openaimodel.py: ResBlock :160-275, AttentionBlock :278-331, QKVAttentionLegacy :353-381, Upsample/Downsample
// :90-158, timestep_embedding util.py) on channels-last fp32 tensors.  Convolutions and the 1x1 qkv / proj
// "conv1d"s run on the conv engine (conv.hip); this file adds GroupNorm(32)+SiLU(+FiLM), the spatial
// self-attention, nearest-2x upsampling, 2x2 average pooling, the small timestep-embedding linears.
#include "ldm_ops.h"
#include "conv.h"

// ---------------------------------------------------------------------------------------------
// GroupNorm(32, C): statistics per (image, group) over HW x (C/32) values in fp32 (as torch's native_group_norm).
// ---------------------------------------------------------------------------------------------
// Stage 1: block (image, group, slice) -> (count, mean, M2) of its slice of the HW pixels (two passes over the slice,
// which stays in L2).  Stage 2 (inside the apply kernel): the slices are merged with Chan's parallel-variance formula.
#define GN_SLICES S3D_GN_SLICES   // (conv.h: the fused-GroupNorm convolution merges the same partial moments)
// Input = channel concatenation of x (C0 channels) and x1 (C - C0 channels; NULL when C0 == C): the th.cat([h, hs.pop()],
// dim=1) in front of the output blocks' first ResBlock (openaimodel.py:750) is never materialised — the GroupNorm groups
// straddle the two tensors, so the concatenation happens in this kernel's loads (its OUTPUT is one tensor).
struct GnSrc {
    const float* x;
    const float* x1;
    int C0;
    // Round 5: source 0 as the raw split-K partial sums of the convolution that produces it (conv.hip: [split][pixel][C0]).
    // The statistics kernels read every element exactly once: they add the splits up in split order (the order of
    // conv_splitk_finish_kernel: the same bits), add the bias and the residual, write the finished tensor to `fin` and go
    // on with the value — the convolution's finish pass (70 launches of ~5 us per denoising step) is gone.
    const float* part;      // NULL: x is a finished tensor
    int nsplit;
    long pstride;           // floats between consecutive splits (= N * HW * C0)
    const float* pshift;    // [C0] bias, or NULL
    const float* pres;      // residual (N, HW, C0), or NULL
    float* fin;             // finished tensor (N, HW, C0)
    __device__ __forceinline__ f32x4 part4(long p, int c) const {   // elements (pixel p, channels c..c+3) of source 0, finished
        const float* q = part + p * C0 + c;
        f32x4 sacc = zero4();
        int k = 0;
        for (; k + 4 <= nsplit; k += 4) {   // four partials in flight, summed as the finish kernel sums them
            const f32x4 v0 = ld4(q + (long)k * pstride), v1 = ld4(q + (long)(k + 1) * pstride), v2 = ld4(q + (long)(k + 2) * pstride),
                        v3 = ld4(q + (long)(k + 3) * pstride);
            sacc += (v0 + v1) + (v2 + v3);
        }
        for (; k < nsplit; ++k) sacc += ld4(q + (long)k * pstride);
        if (pshift) sacc += ld4(pshift + c);
        if (pres) sacc += ld4(pres + p * C0 + c);
        st4(fin + p * C0 + c, sacc);
        return sacc;
    }
    __device__ __forceinline__ float part1(long p, int c) const {
        const float* q = part + p * C0 + c;
        float sacc = 0.f;
        int k = 0;
        for (; k + 4 <= nsplit; k += 4)
            sacc += (q[(long)k * pstride] + q[(long)(k + 1) * pstride]) + (q[(long)(k + 2) * pstride] + q[(long)(k + 3) * pstride]);
        for (; k < nsplit; ++k) sacc += q[(long)k * pstride];
        if (pshift) sacc += pshift[c];
        if (pres) sacc += pres[p * C0 + c];
        fin[p * C0 + c] = sacc;
        return sacc;
    }
    __device__ __forceinline__ float at(long n_hw_p, int c, int C) const {   // element (pixel index over N*HW, channel)
        if (c < C0) return part ? part1(n_hw_p, c) : x[n_hw_p * C0 + c];
        return x1[n_hw_p * (C - C0) + (c - C0)];
    }
};
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnSrc src, int HW, int C, int groups,
                                                       float* __restrict__ part) {
    __shared__ float red[4];
    const int sl = blockIdx.x % GN_SLICES;
    const int gidx = (blockIdx.x / GN_SLICES) % groups, n = blockIdx.x / (GN_SLICES * groups);
    const int cpg = C / groups;
    const int p0 = (int)((long)HW * sl / GN_SLICES), p1 = (int)((long)HW * (sl + 1) / GN_SLICES);
    const long pbase = (long)n * HW + p0;
    const long total = (long)(p1 - p0) * cpg;
    auto block_sum = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    float s = 0.f;
    for (long i = threadIdx.x; i < total; i += 256) s += src.at(pbase + i / cpg, gidx * cpg + (int)(i % cpg), C);
    const float mean = total > 0 ? block_sum(s) / (float)total : 0.f;
    float v = 0.f;
    for (long i = threadIdx.x; i < total; i += 256) {
        const float d = src.at(pbase + i / cpg, gidx * cpg + (int)(i % cpg), C) - mean;
        v += d * d;
    }
    const float m2 = block_sum(v);
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x] = (float)total;
        part[3 * blockIdx.x + 1] = mean;
        part[3 * blockIdx.x + 2] = m2;
    }
}
// Round 4: the same partial moments from ONE coalesced pass.  gn_stats_kernel above walks a group's C/32 channels of every
// pixel (24-byte runs at a C*4-byte stride) twice; the one-launch gn_fused_kernel below reads the same way with only
// N*32 workgroups — 15 / 21 us for a 3 MB map.  Here a workgroup owns a slab of pixels of one image with ALL channels:
// thread = (pixel lane, channel quad), 16-byte loads along the channel axis, shifted sums sum(x - k), sum((x - k)^2) about
// the slab's first pixel k[c] (the cancellation of E[d^2] - E[d]^2 is then of the size (mean - k)^2 / var of a channel
// inside one slab), per-channel totals through LDS adds, then one thread per group folds its C/32 channels into
// (count, mean, M2) — the partial gn_apply_kernel merges with Chan's formula, as before.
__global__ __launch_bounds__(1024) void gn_stats_rows_kernel(const GnSrc src, int HW, int C, int groups,
                                                             float* __restrict__ part) {
    extern __shared__ float s_g[];          // k[C] | sum1[C] | sum2[C] | per-lane partials [lanes][2][C]
    float* s_k = s_g;
    float* s_1 = s_g + C;
    float* s_2 = s_g + 2 * C;
    float* s_p = s_g + 3 * C;
    const int sl = blockIdx.x % GN_SLICES, n = blockIdx.x / GN_SLICES;
    const int p0 = (int)((long)HW * sl / GN_SLICES), p1 = (int)((long)HW * (sl + 1) / GN_SLICES);
    const int cq = C >> 2, lanes = 1024 / cq;            // pixel lanes per pass (C <= 4096)
    const int cqi = threadIdx.x % cq, pr = threadIdx.x / cq;
    const int c = 4 * cqi;
    auto row = [&](int p) {
        const long q = (long)n * HW + p;
        if (c < src.C0) return src.part ? src.part4(q, c) : ld4(src.x + q * src.C0 + c);
        return ld4(src.x1 + q * (C - src.C0) + (c - src.C0));
    };
    if (pr < lanes) {
        f32x4 a1 = zero4(), a2 = zero4();
        if (p1 > p0) {
            const f32x4 k = row(p0);
            int p = p0 + pr;
            for (; p + 3 * lanes < p1; p += 4 * lanes) {   // four rows in flight (the loop is latency-bound otherwise)
                const f32x4 r0 = row(p), r1 = row(p + lanes), r2 = row(p + 2 * lanes), r3 = row(p + 3 * lanes);
                const f32x4 d0 = r0 - k, d1 = r1 - k, d2 = r2 - k, d3 = r3 - k;
                a1 += (d0 + d1) + (d2 + d3);
                a2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            for (; p < p1; p += lanes) {
                const f32x4 d = row(p) - k;
                a1 += d;
                a2 += d * d;
            }
            if (pr == 0) st4(s_k + c, k);
        } else if (pr == 0) {
            st4(s_k + c, zero4());
        }
        st4(s_p + (size_t)(2 * pr) * C + c, a1);
        st4(s_p + (size_t)(2 * pr + 1) * C + c, a2);
    }
    __syncthreads();
    if (pr == 0) {   // the pixel lanes' partials in a fixed order (no atomics: the result is bit-reproducible)
        f32x4 t1 = zero4(), t2 = zero4();
        for (int l = 0; l < lanes; ++l) {
            t1 += ld4(s_p + (size_t)(2 * l) * C + c);
            t2 += ld4(s_p + (size_t)(2 * l + 1) * C + c);
        }
        st4(s_1 + c, t1);
        st4(s_2 + c, t2);
    }
    __syncthreads();
    if ((int)threadIdx.x < groups) {
        const int g = threadIdx.x, cpg = C / groups;
        const float np = (float)(p1 - p0);
        float mean_g = 0.f;
        for (int j = 0; j < cpg; ++j) mean_g += s_k[g * cpg + j] + s_1[g * cpg + j] / fmaxf(np, 1.f);
        mean_g /= (float)cpg;
        float m2 = 0.f;
        for (int j = 0; j < cpg; ++j) {
            const float s1 = s_1[g * cpg + j], mc = s_k[g * cpg + j] + s1 / fmaxf(np, 1.f);
            m2 += (s_2[g * cpg + j] - s1 * s1 / fmaxf(np, 1.f)) + np * (mc - mean_g) * (mc - mean_g);
        }
        float* o = part + 3 * ((size_t)(n * groups + g) * GN_SLICES + sl);
        o[0] = np * (float)cpg;
        o[1] = mean_g;
        o[2] = fmaxf(m2, 0.f);
    }
}
// y = gn(x) * gamma + beta ; optional FiLM (ResBlock use_scale_shift_norm): y = y * (1 + scale[n,c]) + shift[n,c]
// with film = [N][2C] (scale | shift) ; optional SiLU.
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnSrc src, const float* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ film, long film_ld, float* __restrict__ y,
                                                       int N, int HW, int C, int groups, float eps, int silu) {
    // every block first merges the slice moments of all (image, group) pairs (Chan's parallel-variance formula)
    extern __shared__ float s_stats[];   // [N*groups][2] = mean, rstd
    // eight lanes per (image, group): each folds every eighth slice (independent loads), then three pairwise merges
    for (int i0 = 0; i0 < N * groups; i0 += 32) {
        const int i = i0 + (threadIdx.x >> 3), j = threadIdx.x & 7;
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
        if (i < N * groups) {
            float nb[GN_SLICES / 8], mb[GN_SLICES / 8], qb[GN_SLICES / 8];
#pragma unroll
            for (int k = 0; k < GN_SLICES / 8; ++k) {
                const float* q = part + 3 * ((size_t)i * GN_SLICES + j + 8 * k);
                nb[k] = q[0]; mb[k] = q[1]; qb[k] = q[2];
            }
#pragma unroll
            for (int k = 0; k < GN_SLICES / 8; ++k) {
                if (nb[k] == 0.f) continue;
                const float tot = cnt + nb[k], d = mb[k] - mean;
                mean += d * nb[k] / tot;
                m2 += qb[k] + d * d * cnt * nb[k] / tot;
                cnt = tot;
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {   // Chan's merge with lane j ^ o (same order on both sides of a pair: symmetric formula)
            const float cb = __shfl_xor(cnt, o, 64), mbb = __shfl_xor(mean, o, 64), qbb = __shfl_xor(m2, o, 64);
            const float tot = cnt + cb;
            if (tot > 0.f) {
                const float d = mbb - mean;
                const float nm = (cnt * mean + cb * mbb) / tot;
                m2 = m2 + qbb + d * d * cnt * cb / tot;
                mean = nm;
            }
            cnt = tot;
        }
        if (i < N * groups && j == 0) {
            s_stats[2 * i] = mean;
            s_stats[2 * i + 1] = 1.f / sqrtf(m2 / cnt + eps);
        }
    }
    __syncthreads();
    const int c4n = C >> 2, cpg = C / groups;
    const long total = (long)N * HW * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long p = idx / c4n;
        const int n = (int)(p / HW);
        const f32x4 v = c < src.C0 ? ld4(src.x + p * src.C0 + c) : ld4(src.x1 + p * (C - src.C0) + (c - src.C0));   // C0 % 4 == 0
        const f32x4 ga = ld4(gamma + c), be = ld4(beta + c);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = (c + i) / cpg;
            const float m = s_stats[2 * (n * groups + g)], r = s_stats[2 * (n * groups + g) + 1];
            float t = (v[i] - m) * r * ga[i] + be[i];
            if (film) t = t * (1.f + film[(long)n * film_ld + c + i]) + film[(long)n * film_ld + C + c + i];
            if (silu) t = t / (1.f + expf(-t));
            o[i] = t;
        }
        st4(y + p * C + c, o);
    }
}

// Sliced statistics -> the affine table of the fused-GroupNorm convolution: one workgroup per image merges the slab moments of
// every group (eight lanes per group + three pairwise Chan merges, as gn_apply_kernel) and writes A | B of all channels.
__global__ __launch_bounds__(256) void gn_table_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ film,
                                                       long film_ld, float* __restrict__ table, int C, int groups, float eps) {
    __shared__ float s_st[2 * 64];
    const int n = blockIdx.x;
    for (int i0 = 0; i0 < groups; i0 += 32) {
        const int i = i0 + (threadIdx.x >> 3), j = threadIdx.x & 7;
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
        if (i < groups) {
            float nb[GN_SLICES / 8], mb[GN_SLICES / 8], qb[GN_SLICES / 8];
#pragma unroll
            for (int k = 0; k < GN_SLICES / 8; ++k) {
                const float* q = part + 3 * ((size_t)(n * groups + i) * GN_SLICES + j + 8 * k);
                nb[k] = q[0]; mb[k] = q[1]; qb[k] = q[2];
            }
#pragma unroll
            for (int k = 0; k < GN_SLICES / 8; ++k) {
                if (nb[k] == 0.f) continue;
                const float tot = cnt + nb[k], d = mb[k] - mean;
                mean += d * nb[k] / tot;
                m2 += qb[k] + d * d * cnt * nb[k] / tot;
                cnt = tot;
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const float cb = __shfl_xor(cnt, o, 64), mbb = __shfl_xor(mean, o, 64), qbb = __shfl_xor(m2, o, 64);
            const float tot = cnt + cb;
            if (tot > 0.f) {
                const float d = mbb - mean;
                const float nm = (cnt * mean + cb * mbb) / tot;
                m2 = m2 + qbb + d * d * cnt * cb / tot;
                mean = nm;
            }
            cnt = tot;
        }
        if (i < groups && j == 0) {
            s_st[2 * i] = mean;
            s_st[2 * i + 1] = 1.f / sqrtf(m2 / cnt + eps);
        }
    }
    __syncthreads();
    const int cpg = C / groups;
    for (int c = threadIdx.x; c < C; c += 256) {
        const int gi = c / cpg;
        float A = s_st[2 * gi + 1] * gamma[c], B = beta[c] - s_st[2 * gi] * A;
        if (film) {
            const float sc = 1.f + film[(long)n * film_ld + c];
            A *= sc;
            B = B * sc + film[(long)n * film_ld + C + c];
        }
        table[(size_t)n * 2 * C + c] = A;
        table[(size_t)n * 2 * C + C + c] = B;
    }
}

// One-launch variant for the maps of this U-Net (<= 64 x 64): a 1024-thread workgroup owns one (image, group), keeps
// its HW x C/32 values in registers (EPT per thread), reduces mean and centred second moment through LDS and writes
// the normalised values - statistics, apply, FiLM and SiLU in one pass over the data instead of two launches
// (the denoising step has 92 GroupNorms, most of them a few microseconds of launch latency each).
template <int EPT>
__global__ __launch_bounds__(1024) void gn_fused_kernel(const GnSrc src, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ film,
                                                        long film_ld, float* __restrict__ y, int HW, int C, int groups,
                                                        float eps, int silu, float* __restrict__ table) {
    __shared__ float red[16];
    const int gidx = blockIdx.x % groups, n = blockIdx.x / groups;
    const int cpg = C / groups;
    const long total = (long)HW * cpg;
    float* obase = y + (long)n * HW * C + gidx * cpg;
    auto block_sum = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        return t;
    };
    float v[EPT];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const long i = threadIdx.x + 1024L * k;
        v[k] = i < total ? src.at((long)n * HW + i / cpg, gidx * cpg + (int)(i % cpg), C) : 0.f;
        s += v[k];
    }
    const float mean = block_sum(s) / (float)total;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const long i = threadIdx.x + 1024L * k;
        const float d = i < total ? v[k] - mean : 0.f;
        q += d * d;
    }
    const float m2 = block_sum(q);
    if (table) {   // no normalised output: this group's rows of the affine table y = x * A + B (gamma / beta / FiLM folded) that the
                   // consuming convolution applies while it stages its input (ConvGn, conv.h)
        const float rs = 1.f / sqrtf(m2 / (float)total + eps);
        if ((int)threadIdx.x < cpg) {
            const int c = gidx * cpg + threadIdx.x;
            float A = rs * gamma[c], B = beta[c] - mean * A;
            if (film) {
                const float sc = 1.f + film[(long)n * film_ld + c];
                A *= sc;
                B = B * sc + film[(long)n * film_ld + C + c];
            }
            table[(size_t)n * 2 * C + c] = A;
            table[(size_t)n * 2 * C + C + c] = B;
        }
        return;
    }
    const float rstd = 1.f / sqrtf(m2 / (float)total + eps);
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const long i = threadIdx.x + 1024L * k;
        if (i < total) {
            const int c = gidx * cpg + (int)(i % cpg);
            float t = (v[k] - mean) * rstd * gamma[c] + beta[c];
            if (film) t = t * (1.f + film[(long)n * film_ld + c]) + film[(long)n * film_ld + C + c];
            if (silu) t = t / (1.f + expf(-t));
            obase[(i / cpg) * C + i % cpg] = t;
        }
    }
}

// table != NULL (then y is unused): no normalised output — the per-(image, channel) affine table [N][2][C] = A | B of
// y = x * A + B with the statistics, gamma / beta and FiLM folded, which a fused-GroupNorm convolution (ConvGn, conv.h) applies
// while it stages its input
int launch_group_norm(const float* x, const float* gamma, const float* beta, const float* film, float* y, float* stats,
                      int N, int HW, int C, int groups, float eps, int silu, hipStream_t stream, const float* x1,
                      int C0, long film_ld, float* table, const GnPartial* part) {
    if (film_ld <= 0) film_ld = 2L * C;   // rows of a dense (N, 2C) film tensor
    S3D_CHECK_ARG(!table || groups <= 64, "group_norm table: %d groups", groups);
    S3D_CHECK_ARG(!part || (part->part && part->fin && part->nsplit >= 1), "group_norm: bad partial-sum source");
    S3D_CHECK_ARG(C % groups == 0 && C % 4 == 0 && N >= 1 && HW >= 1, "group_norm: C=%d groups=%d", C, groups);
    if (!x1) C0 = C;
    S3D_CHECK_ARG(C0 >= 4 && C0 <= C && C0 % 4 == 0 && (C - C0) % 4 == 0, "group_norm: source split %d | %d", C0, C - C0);
    GnSrc src = {x, x1, C0, nullptr, 0, 0, nullptr, nullptr, nullptr};
    if (part) {
        S3D_CHECK_ARG(C0 % 4 == 0, "group_norm: partial-sum source needs C0 %% 4 == 0");
        src.x = part->fin;      // (what a second pass over the input reads: the finished tensor)
        src.part = part->part; src.nsplit = part->nsplit; src.pstride = (long)N * HW * C0;
        src.pshift = part->shift; src.pres = part->res; src.fin = part->fin;
    }
    {
        const long per_thread = ((long)HW * (C / groups) + 1023) / 1024;
        if (per_thread <= 8 || (per_thread <= 24 && N * groups >= 128)) {   // (with >= 128 workgroups the one-launch form wins again)   // small maps: one launch (a (image, group) slab in the registers of one workgroup); larger ones
                                 // run the two coalesced kernels below — the fused kernel's strided reads took 21 us there
            const dim3 grid((unsigned)(N * groups));
#define GN_CASE(e)                                                                                                     \
    if (per_thread <= e) {                                                                                             \
        hipLaunchKernelGGL((gn_fused_kernel<e>), grid, dim3(1024), 0, stream, src, gamma, beta, film, film_ld, y, HW, C, groups, eps, \
                           silu, table);                                                                               \
        S3D_LAUNCH_CHECK();                                                                                            \
        return 0;                                                                                                      \
    }
            GN_CASE(2) GN_CASE(8) GN_CASE(24)
#undef GN_CASE
        }
    }
    S3D_CHECK_ARG((size_t)N * groups * 2 * sizeof(float) <= 48 * 1024, "group_norm: N*groups %d too large", N * groups);
    if (C <= 2048 && groups <= 1024)   // (LDS: 7 C floats) coalesced one-pass partial moments (all channels of a pixel slab per workgroup)
        hipLaunchKernelGGL(gn_stats_rows_kernel, dim3(N * GN_SLICES), dim3(1024),
                           (size_t)(3 * C + 2 * (1024 / (C / 4)) * C) * sizeof(float), stream, src, HW, C, groups, stats);
    else {
        S3D_CHECK_ARG(!part, "group_norm: partial-sum source with C = %d > 2048", C);
        hipLaunchKernelGGL(gn_stats_kernel, dim3(N * groups * GN_SLICES), dim3(256), 0, stream, src, HW, C, groups, stats);
    }
    S3D_LAUNCH_CHECK();
    if (table) {
        hipLaunchKernelGGL(gn_table_kernel, dim3(N), dim3(256), 0, stream, stats, gamma, beta, film, film_ld, table, C, groups, eps);
        S3D_LAUNCH_CHECK();
        return 0;
    }
    const long total = (long)N * HW * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    src.part = nullptr;   // the statistics kernel has written the finished tensor (src.x == part->fin)
    hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks), dim3(256), (size_t)N * groups * 2 * sizeof(float), stream, src, stats,
                       gamma, beta, film, film_ld, y, N, HW, C, groups, eps, silu);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// QKVAttentionLegacy (openaimodel.py:353-381) on the token-major output of the qkv 1x1 conv:
//   qkv [N][T][heads][3][ch]  ->  out [N][T][heads][ch],  softmax((q s)(k s)^T) v,  s = ch^-1/4, fp32.
// One workgroup = 64 queries of one (image, head); wave = 16 queries.  Keys / values stream through LDS in
// blocks of 64 with an online softmax.  S^T = K Q^T and O^T = V^T P^T run on the fp32 16x16x4 MFMA: the S^T
// registers (lane (query, g): keys 4g+i) are directly the B operand of the second product.
// ---------------------------------------------------------------------------------------------
#define QA_KB 64
// KW = 2: eight waves; waves 4..7 take the odd key blocks of the same 64 queries (own LDS buffers, staged by
// themselves) and the two partial softmax states are merged through LDS at the end.  Used when the grid is too
// small to give every SIMD more than two waves (batch 1: 512 workgroups).
template <int CH, int KW>
__global__ __launch_bounds__(256 * KW) void qkv_attention_kernel(const float* __restrict__ qkv,
                                                                 float* __restrict__ out, int T, int heads) {
    constexpr int LD = CH + 4;               // padded row: 16 query/key lanes hit 16 distinct bank quads
    constexpr int DT = (CH + 15) / 16;       // 16-wide tiles of the head dimension (zero padded)
    constexpr int KS = CH / 4;               // fp32 MFMA k-steps of the q.k contraction
    constexpr int LDV = DT * 16 + 4;
    constexpr int NLD = (QA_KB * (CH / 4) + 255) / 256;   // float4 (K, V) pairs a thread stages per key block
    __shared__ float s_kk[KW][2][QA_KB * LD], s_vv[KW][2][QA_KB * LDV];
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, kh = threadIdx.x >> 8;
    const int tid = threadIdx.x & 255;   // index within the key half
    float(*s_k)[QA_KB * LD] = s_kk[kh];
    float(*s_v)[QA_KB * LDV] = s_vv[kh];
    const int m = lane & 15, g = lane >> 4;
    const int qblocks = (T + 63) / 64;
    const int qb = blockIdx.x % qblocks;
    const int hh = (blockIdx.x / qblocks) % heads, n = blockIdx.x / (qblocks * heads);
    const int C3 = heads * 3 * CH;
    const float* base = qkv + (long)n * T * C3 + hh * 3 * CH;
    const float scale = 1.f / sqrtf(sqrtf((float)CH));
    const int q = qb * 64 + wave * 16 + m;
    const int qc = q < T ? q : T - 1;
    float qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = base[(long)qc * C3 + 4 * s + g] * scale;
    f32x4 acc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) acc[d] = zero4();
    float mx = -1e30f, den = 0.f;

    // key/value blocks are double buffered: the next block's global loads are in flight under this block's MFMAs
    f32x4 pk[NLD], pv[NLD];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i;
            const int key = idx / (CH / 4), c4 = (idx % (CH / 4)) * 4;
            const int kc = (idx < QA_KB * (CH / 4) && k0 + key < T) ? k0 + key : T - 1;
            const float* row = base + (long)kc * C3;
            pk[i] = ld4(row + CH + c4);
            pv[i] = ld4(row + 2 * CH + c4);
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i;
            if (idx < QA_KB * (CH / 4)) {
                const int key = idx / (CH / 4), c4 = (idx % (CH / 4)) * 4;
                float* dk = s_k[buf] + key * LD + c4;
                float* dv = s_v[buf] + key * LDV + c4;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    dk[t] = pk[i][t] * scale;
                    dv[t] = pv[i][t];
                }
            }
        }
    };
    if (CH % 16)   // zero the padded head dims of V once (both buffers)
        for (int i = tid; i < 2 * QA_KB * (DT * 16 - CH); i += 256) {
            const int buf = i / (QA_KB * (DT * 16 - CH)), r = i % (QA_KB * (DT * 16 - CH));
            s_v[buf][(r / (DT * 16 - CH)) * LDV + CH + r % (DT * 16 - CH)] = 0.f;
        }
    // this half's blocks: kh, kh + KW, ...; both halves run the same trip count (blocks past T are fully masked)
    const int nit = ((T + QA_KB - 1) / QA_KB + KW - 1) / KW;
    fetch(kh * QA_KB);
    park(0);
    __syncthreads();
    int buf = 0;
    for (int it = 0; it < nit; ++it, buf ^= 1) {
        const int k0 = (it * KW + kh) * QA_KB;
        if (it + 1 < nit) fetch(k0 + KW * QA_KB);
        const float* sk = s_k[buf];
        const float* sv_ = s_v[buf];
        // S^T[key][query]: lane (query m, g) gets keys kt*16 + 4g + i
        f32x4 sv[4];
        float bmax = -1e30f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 s = zero4();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(sk[(kt * 16 + m) * LD + 4 * ks + g], qf[ks], s, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (k0 + kt * 16 + 4 * g + i >= T) s[i] = -1e30f;
                bmax = fmaxf(bmax, s[i]);
            }
            sv[kt] = s;
        }
        bmax = fmaxf(bmax, __shfl_xor(bmax, 16, 64));
        bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
        const float mnew = fmaxf(mx, bmax);
        const float corr = expf(mx - mnew);
        float bsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = sv[kt][i] > -1e29f ? expf(sv[kt][i] - mnew) : 0.f;
                sv[kt][i] = p;
                bsum += p;
            }
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        den = den * corr + bsum;
        mx = mnew;
        // O^T[d][query] = corr * O^T + sum_key V[key][d] P[query][key]
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            f32x4 o = acc[d] * corr;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(sv_[(kt * 16 + 4 * g + i) * LDV + d * 16 + m], sv[kt][i], o, 0,
                                                            0, 0);
            acc[d] = o;
        }
        if (it + 1 < nit) park(buf ^ 1);
        __syncthreads();
    }
    if constexpr (KW > 1) {   // merge the two key halves' (max, denominator, accumulator) states; s_vv[1] is free now
        float* mrg = &s_vv[1][0][0] + (wave * 64 + lane) * (4 * DT + 2);
        if (kh == 1) {
            mrg[0] = mx;
            mrg[1] = den;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int i = 0; i < 4; ++i) mrg[2 + 4 * d + i] = acc[d][i];
        }
        __syncthreads();
        if (kh == 1) return;
        const float m1 = mrg[0], d1 = mrg[1];
        const float mm = fmaxf(mx, m1);
        const float w0 = expf(mx - mm), w1 = expf(m1 - mm);
        den = den * w0 + d1 * w1;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[d][i] = acc[d][i] * w0 + mrg[2 + 4 * d + i] * w1;
    }
    if (q < T) {
        const float inv = 1.f / den;
        float* o = out + ((long)n * T + q) * (heads * CH) + hh * CH;
#pragma unroll
        for (int d = 0; d < DT; ++d)
            if (d * 16 + 4 * g + 3 < CH) st4(o + d * 16 + 4 * g, acc[d] * inv);
    }
}

int launch_qkv_attention(const float* qkv, float* out, int N, int T, int heads, int ch, int prec, hipStream_t stream) {
    S3D_CHECK_ARG(N >= 1 && T >= 1 && heads >= 1, "qkv_attention: bad dims");
    const int blocks = N * heads * ((T + 63) / 64);
    // batch-1 grids (512 workgroups on 256 CUs) run the eight-wave key-split form; its LDS fits up to 48 channels
    const bool split = blocks <= 1024 && ch <= 48 && T > QA_KB;
#define QA_CASE(c)                                                                                                 \
    if (ch == c) {                                                                                                 \
        if (split && c <= 48)                                                                                      \
            hipLaunchKernelGGL((qkv_attention_kernel<c, (c <= 48 ? 2 : 1)>), dim3(blocks), dim3(c <= 48 ? 512 : 256), 0, \
                               stream, qkv, out, T, heads);                                                        \
        else                                                                                                       \
            hipLaunchKernelGGL((qkv_attention_kernel<c, 1>), dim3(blocks), dim3(256), 0, stream, qkv, out, T, heads); \
        S3D_LAUNCH_CHECK();                                                                                        \
        return 0;                                                                                                  \
    }
    QA_CASE(8) QA_CASE(16) QA_CASE(24) QA_CASE(32) QA_CASE(48) QA_CASE(64) QA_CASE(96)
#undef QA_CASE
    s3d_set_error("qkv_attention: head width %d not built (8,16,24,32,48,64,96)", ch);
    return S3D_E_ARG;
}

// ---------------------------------------------------------------------------------------------
// resampling (openaimodel.py:90-158, use_conv = False): nearest 2x up / 2x2 average pool, channels-last
// ---------------------------------------------------------------------------------------------
__global__ void upsample2x_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
    const int c4n = C >> 2;
    const long total = (long)N * 4 * H * W * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        long p = idx / c4n;
        const int ox = (int)(p % (2 * W));
        p /= 2 * W;
        const int oy = (int)(p % (2 * H)), n = (int)(p / (2 * H));
        st4(y + (((long)n * 2 * H + oy) * 2 * W + ox) * C + c, ld4(x + (((long)n * H + oy / 2) * W + ox / 2) * C + c));
    }
}
__global__ void avgpool2x_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
    const int c4n = C >> 2, Ho = H / 2, Wo = W / 2;
    const long total = (long)N * Ho * Wo * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        long p = idx / c4n;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho), n = (int)(p / Ho);
        const float* b = x + (((long)n * H + 2 * oy) * W + 2 * ox) * C + c;
        st4(y + (((long)n * Ho + oy) * Wo + ox) * C + c,
            ((ld4(b) + ld4(b + C)) + (ld4(b + (long)W * C) + ld4(b + (long)W * C + C))) * 0.25f);
    }
}
int launch_resample2x(const float* x, float* y, int N, int H, int W, int C, int up, hipStream_t stream) {
    S3D_CHECK_ARG(C % 4 == 0 && N >= 1 && H >= 1 && W >= 1 && (up || (H % 2 == 0 && W % 2 == 0)), "resample2x: bad dims");
    const long total = up ? (long)N * 4 * H * W * (C / 4) : (long)N * (H / 2) * (W / 2) * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (up)
        hipLaunchKernelGGL(upsample2x_kernel, dim3(blocks), dim3(256), 0, stream, x, y, N, H, W, C);
    else
        hipLaunchKernelGGL(avgpool2x_kernel, dim3(blocks), dim3(256), 0, stream, x, y, N, H, W, C);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// small dense layers on the timestep embedding: out[n][m] = b[m] + sum_k w[m][k] f(x[n][k]),  f = SiLU or id.
// A weight-streaming GEMV: the step's largest one (768 -> 36 096: the FiLM rows of every ResBlock at once) reads 111 MB
// of weights for 28 MFLOP, so the only figure that matters is bytes in flight.  f(x) of up to four images is staged in
// LDS once per workgroup; a wave owns four output rows at a time and reads them as 16-byte loads (K / 256 x 4 of them
// requested before the first use), so every weight byte is read once for all images.  K % 4 != 0: the scalar form.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void small_linear_scalar_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ b, float* __restrict__ out, int N,
                                                                  int K, int M, int silu_in) {
    const int lane = threadIdx.x & 63;
    const long total = (long)N * M;
    for (long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6); o < total; o += (long)gridDim.x * 4) {
        const int n = (int)(o / M), mo = (int)(o % M);
        float s = 0.f;
        for (int k = lane; k < K; k += 64) {
            float v = x[(long)n * K + k];
            if (silu_in) v = v / (1.f + expf(-v));
            s += w[(long)mo * K + k] * v;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
        if (lane == 0) out[o] = s + (b ? b[mo] : 0.f);
    }
}
#define SL_ROWS 4   // output rows per wave pass
#define SL_NB 4     // images per pass
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ out, int N,
                                                           int K, int M, int silu_in) {
    extern __shared__ __attribute__((aligned(16))) float s_x[];   // [SL_NB][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n0 = 0; n0 < N; n0 += SL_NB) {
        const int nb = N - n0 < SL_NB ? N - n0 : SL_NB;
        if (n0) __syncthreads();
        for (int i = threadIdx.x; i < SL_NB * K; i += 256) {
            float v = i < nb * K ? x[(long)n0 * K + i] : 0.f;
            if (silu_in) v = v / (1.f + expf(-v));
            s_x[i] = v;
        }
        __syncthreads();
        for (int m0 = (blockIdx.x * 4 + wave) * SL_ROWS; m0 < M; m0 += gridDim.x * 4 * SL_ROWS) {
            float acc[SL_ROWS][SL_NB];
#pragma unroll
            for (int r = 0; r < SL_ROWS; ++r)
#pragma unroll
                for (int n = 0; n < SL_NB; ++n) acc[r][n] = 0.f;
            const float* wr[SL_ROWS];
#pragma unroll
            for (int r = 0; r < SL_ROWS; ++r) wr[r] = w + (long)(m0 + r < M ? m0 + r : M - 1) * K;
#pragma unroll 4
            for (int k = lane * 4; k < K; k += 256) {
                f32x4 wv[SL_ROWS];
#pragma unroll
                for (int r = 0; r < SL_ROWS; ++r) wv[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wr[r] + k));
#pragma unroll
                for (int n = 0; n < SL_NB; ++n) {
                    if (n >= nb) break;   // uniform
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(s_x + n * K + k);
#pragma unroll
                    for (int r = 0; r < SL_ROWS; ++r)
                        acc[r][n] += wv[r][0] * xv[0] + wv[r][1] * xv[1] + wv[r][2] * xv[2] + wv[r][3] * xv[3];
                }
            }
#pragma unroll
            for (int r = 0; r < SL_ROWS; ++r)
#pragma unroll
                for (int n = 0; n < SL_NB; ++n) {
                    if (n >= nb) break;
                    float v = acc[r][n];
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
                    if (lane == 0 && m0 + r < M) out[(long)(n0 + n) * M + m0 + r] = v + (b ? b[m0 + r] : 0.f);
                }
        }
    }
}
int launch_small_linear(const float* x, const float* w, const float* b, float* out, int N, int K, int M, int silu_in,
                        hipStream_t stream) {
    S3D_CHECK_ARG(N >= 1 && K >= 1 && M >= 1, "small_linear: bad dims");
    // (below ~4 096 rows the scalar form's one wave per output is the lower-latency one: 768 -> 768 takes 5.4 against 7.1 us)
    if (M >= 4096 && K % 4 == 0 && (size_t)SL_NB * K * sizeof(float) <= 48 * 1024 && ((size_t)w & 15) == 0) {
        const int groups = (M + 4 * SL_ROWS - 1) / (4 * SL_ROWS);
        const int blocks = groups < 8192 ? groups : 8192;
        hipLaunchKernelGGL(small_linear_kernel, dim3(blocks), dim3(256), (size_t)SL_NB * K * sizeof(float), stream, x, w, b, out, N,
                           K, M, silu_in);
    } else {
        const long total = (long)N * M;
        const int blocks = (int)((total + 3) / 4 < 4096 ? (total + 3) / 4 : 4096);
        hipLaunchKernelGGL(small_linear_scalar_kernel, dim3(blocks), dim3(256), 0, stream, x, w, b, out, N, K, M, silu_in);
    }
    S3D_LAUNCH_CHECK();
    return 0;
}

// timestep_embedding (util.py:151-170): [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(max_period) i / half)
__global__ void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int N, int dim,
                                          float max_period) {
    const int half = dim / 2;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < N * half; idx += gridDim.x * blockDim.x) {
        const int n = idx / half, i = idx % half;
        const float freq = expf(-logf(max_period) * (float)i / (float)half);
        const float a = t[n] * freq;
        out[(long)n * dim + i] = cosf(a);
        out[(long)n * dim + half + i] = sinf(a);
        if ((dim & 1) && i == 0) out[(long)n * dim + dim - 1] = 0.f;
    }
}
int launch_timestep_embedding(const float* t, float* out, int N, int dim, float max_period, hipStream_t stream) {
    S3D_CHECK_ARG(N >= 1 && dim >= 2, "timestep_embedding: bad dims");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((N * (dim / 2) + 255) / 256), dim3(256), 0, stream, t, out, N, dim,
                       max_period);
    S3D_LAUNCH_CHECK();
    return 0;
}

// out = a + b (c_fmaps injection openaimodel.py:731-746, elementwise glue)
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        st4(out + 4 * i, ld4(a + 4 * i) + ld4(b + 4 * i));
}
int launch_add(const float* a, const float* b, float* out, long n, hipStream_t stream) {
    S3D_CHECK_ARG(n % 4 == 0 && n > 0, "add: n %ld must be a positive multiple of 4", n);
    const long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, stream, a, b, out, n4);
    S3D_LAUNCH_CHECK();
    return 0;
}
