// train.hip — backward / train-mode kernels (gfx950, fp32).
//
// Together with the forward engines (conv.hip, decode.hip) these implement train_step of the reference
// (reg_slices/train.py:41-53): train-mode BatchNorm (unet_parts.py:17,20 and the VGG16-BN encoder),
// loss gradients (train.py:29-39), the backward of every op of models.py:48-94 and Adam
// (train.py:136, torch.optim.Adam defaults).  Data gradients of convolutions / linears reuse the
// forward implicit-GEMM engine with transposed weight packs; this file adds the weight-gradient GEMM
// (contraction over pixels), reductions, normalisation backward, the attention core and optimiser.
#include "train.h"

// =============================================================================================
// weight gradient:  dW[n][k] = sum_p dY[p][n] * X[pix(p,tap)][c]      (k = tap*Cx + c)
// One MFMA step contracts 4 pixels (k-slot g <-> pixel 4*it+g).  Channel <-> lane mapping is
// interleaved (channel = T*x + j for operand tile j, x = l&15) so each lane loads T consecutive
// channels of ONE pixel with a single 4/8/16-byte load and a wave's load of a pixel quad is
// 4 x (16*T*4) contiguous bytes.
// =============================================================================================
template <int T>
__device__ __forceinline__ void load_vec(const float* p, bool ok, int valid, float (&v)[T]) {
    // valid = number of usable channels starting at p (may be <= 0 or >= T)
    if (ok && valid >= T) {
        if (T == 4) {
            const f32x4 t = ld4(p);
            v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        } else {
#pragma unroll
            for (int j = 0; j < T; ++j) v[j] = p[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < T; ++j) v[j] = (ok && j < valid) ? p[j] : 0.f;
    }
}

template <int TN, int TK>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a, int n_cblk, int taps, long P,
                                                    int steps_per_wave, int Ktot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = lane & 15, g = lane >> 4;
    int bx = blockIdx.x;
    const int cb = bx % n_cblk;
    bx /= n_cblk;
    const int tap = bx % taps;
    const int nb = bx / taps;
    const int split = blockIdx.y * 4 + wave;
    const int KS = a.ks, PAD = a.ks == 3 ? 1 : 0;
    const int stride = a.stride > 1 ? a.stride : 1;
    const int Hin = a.Hin ? a.Hin : a.H, Win = a.Win ? a.Win : a.W;
    const int dyo = tap / KS - PAD, dxo = tap % KS - PAD;
    const int n0 = nb * 16 * TN + TN * x, c0 = cb * 16 * TK + TK * x;
    const int HW = a.H * a.W;

    f32x4 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = zero4();

    const long p_begin = (long)split * steps_per_wave * 4;
#pragma unroll 2
    for (int it = 0; it < steps_per_wave; ++it) {
        const long p = p_begin + (long)it * 4 + g;
        const bool pv = p < P;
        const long pc = pv ? p : 0;
        const int img = (int)(pc / HW);
        const int r = (int)(pc - (long)img * HW);
        const int y = r / a.W, xx = r - y * a.W;
        float av[TN], bv[TK];
        load_vec<TN>(a.dy + pc * a.dy_cstride + a.dy_coff + n0, pv, a.N - n0, av);
        const int yi = y * stride + dyo, xi = xx * stride + dxo;
        const bool ok = pv && yi >= 0 && yi < Hin && xi >= 0 && xi < Win;
        const int ni = (a.x.bmod ? img % a.x.bmod : img) / a.x.bdiv;
        const long off = a.x.sbcast ? (long)ni * a.x.C : ((long)(ni * Hin + (ok ? yi : 0)) * Win + (ok ? xi : 0)) * a.x.C;
        load_vec<TK>(a.x.p + off + a.x_coff + c0, ok, a.Cx - c0, bv);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    // D[row = 4g+reg <-> x_n][col = l&15 <-> x_k]
    float* part = a.partial + (size_t)split * a.N * Ktot;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int n = nb * 16 * TN + TN * (4 * g + reg) + i;
                const int c = cb * 16 * TK + TK * x + j;
                if (n < a.N && c < a.Cx) part[(size_t)n * Ktot + tap * a.Cx + c] = acc[i][j][reg];
            }
}

// Reduction of the split partials into the parameter gradient.  A workgroup owns 64 consecutive elements; its four waves
// sum the splits sg, sg + 4, ... (eight loads in flight each) and meet in LDS in a fixed order — deterministic.  One thread
// per element walking all splits left the in_proj gradient (49 152 elements x 342 splits = 67 MB) to 192 workgroups:
// 360 us, latency-bound; ffn_wgrad_rec's reduction moves the same bytes in 12 us.
template <bool BIAS>   // BIAS: the split-precision linear kernel left nsplit x N bias partials behind the weight partials
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs a, int nsplit, int Ktot) {
    __shared__ float s_p[3][64];
    const long total = (long)a.N * Ktot, all = total + (BIAS ? a.N : 0);
    const int e = threadIdx.x & 63, sg = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * 64; base < all; base += (long)gridDim.x * 64) {
        const long idx = base + e;
        const bool live = idx < all;
        const bool is_bias = BIAS && idx >= total;
        // element `idx` of split sp: weight partials [sp][total], bias partials [sp][N] behind them
        const float* src = is_bias ? a.partial + (size_t)nsplit * total + (idx - total) : a.partial + (live ? idx : 0);
        const size_t stride = is_bias ? (size_t)a.N : (size_t)total;
        float s = 0.f;
        if (live) {
            int sp = sg;
            for (; sp + 28 < nsplit; sp += 32) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(sp + 4 * k) * stride];
                __builtin_amdgcn_sched_barrier(0);   // 8 loads in flight
                s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            for (; sp + 12 < nsplit; sp += 16) {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = src[(size_t)(sp + 4 * k) * stride];
                __builtin_amdgcn_sched_barrier(0);
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
            for (; sp < nsplit; sp += 4) s += src[(size_t)sp * stride];
        }
        if (sg) s_p[sg - 1][e] = s;
        __syncthreads();
        if (sg == 0 && live) {
            s = (s + s_p[0][e]) + (s_p[1][e] + s_p[2][e]);
            if (is_bias) {
                float* o = a.bias_out + (idx - total);
                *o = a.accumulate ? *o + s : s;
            } else {
                const int n = (int)(idx / Ktot), k = (int)(idx - (long)n * Ktot);
                long o = -1;
                if (a.out_kind == S3D_PACK_LINEAR) {
                    o = (long)n * a.ld + k;
                } else if (a.out_kind == S3D_PACK_CONV) {
                    const int tap = k / a.Cx, c = k - tap * a.Cx;
                    if (a.cin_begin + c < a.cin_tot)   // (else: padded input channels, 3 -> 16)
                        o = ((long)n * a.cin_tot + a.cin_begin + c) * (a.ks * a.ks) + tap;
                } else {  // CONVT: n = ci, k = q*ct + co
                    const int q = k / a.ct, co = k - q * a.ct;
                    o = ((long)n * a.ct + co) * 4 + q;
                }
                if (o >= 0) a.out[o] = a.accumulate ? a.out[o] + s : s;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Split-precision weight gradient of a plain linear / 1x1 contraction:  dW[n][c] = sum_p dY[p][n] X[p][c].
// Workgroup tile 128 (n) x 128 (c), 4 waves of 64 x 64; the contraction runs over rows in steps of 32
// (one v_mfma_f32_16x16x32_f16 K).  The MFMA wants 8 consecutive ROWS per lane for a fixed channel, i.e.
// the transposed operands: thread (channel, row group) gathers its 8 rows from global (a wave reads 256
// contiguous bytes per row), splits them into f16 hi/lo and writes two 16-byte words into the
// [channel][row] LDS tile (80-byte channel stride: conflict-free for both the writes and the fragment
// reads).  Next step's global loads are issued before the MFMAs of the current one.
// ---------------------------------------------------------------------------------------------
typedef _Float16 wl_half8 __attribute__((ext_vector_type(8)));
#define WL_LD 40     // (a 96-byte stride makes the b128 fragment reads conflict-free but the stores slower: measured no gain)
#define WZ_LD 40
__global__ __launch_bounds__(256) void wgrad_lin_f16x3_kernel(const WgradArgs a, int n_cblk, long P,
                                                              int steps_per_split) {
    __shared__ __attribute__((aligned(16))) _Float16 s_t[2][2][128 * WL_LD];   // [dy|x][hi|lo], 40 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int ch = tid & 127, rg = tid >> 7;
    // Workgroups are dealt to the 8 XCDs round-robin by linear id.  The tiles of one row split read the same rows of dY / X:
    // XCD x owns the splits x, x + 8, ... and runs all tiles of a split back to back, so the second and third tile find the
    // operand they share in that XCD's L2 (in_proj: three 128-row blocks of dQKV x one X — X came from HBM three times).
    int tile, split;
    {
        const int n_tiles = gridDim.x, n_split = gridDim.y;
        const long L = (long)blockIdx.y * n_tiles + blockIdx.x;
        const long full = (long)(n_split & ~7) * n_tiles;
        if (L < full) {
            const long k = L >> 3;
            tile = (int)(k % n_tiles);
            split = (int)(k / n_tiles) * 8 + (int)(L & 7);
        } else {
            const long r = L - full;
            tile = (int)(r % n_tiles);
            split = (n_split & ~7) + (int)(r / n_tiles);
        }
    }
    const int cb = tile % n_cblk, nb = tile / n_cblk;
    const int n0 = nb * 128, c0 = cb * 128;
    const int wn = wave & 1, wc = wave >> 1;
    const bool n_ok = n0 + ch < a.N, c_ok = c0 + ch < a.Cx;
    const float* dyp = a.dy + a.dy_coff + n0 + (n_ok ? ch : 0);
    const float* xp = a.x.p + a.x_coff + c0 + (c_ok ? ch : 0);
    const long dstride = a.dy_cstride, xstride = a.x.C;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = zero4();

    // raw prefetch registers; loads are unconditional (clamped addresses) so that the compiler keeps all 32 in
    // flight — validity is applied as a 0/1 factor when the values are converted, one step later
    float pd[2][8], px[2][8];
    auto gload = [&](long pbase) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                long row = pbase + 16 * ps + 8 * rg + t;
                row = row < P ? row : P - 1;
                pd[ps][t] = dyp[row * dstride];
                px[ps][t] = xp[row * xstride];
            }
    };
    const long p_begin = (long)split * steps_per_split * 32;
    float csum = 0.f;   // this thread's rows of dY channel ch (bias gradient)
    gload(p_begin);
    for (int it = 0; it < steps_per_split; ++it) {
        const long pb = p_begin + (long)it * 32;
        // hi / lo halves with the 1.5-VALU split (v_cvt_pk_f16_f32 + v_fma_mix: the same bits as convert - subtract - convert,
        // which cost 3 per value here); the validity factor only matters in a step that reaches past the last row or for a
        // padding channel — a uniform test
        const bool tail_step = pb + 32 > P;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            float vd[8], vx[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool in = !tail_step || pb + 16 * ps + 8 * rg + t < P;
                vd[t] = (in && n_ok) ? pd[ps][t] : 0.f;
                vx[t] = (in && c_ok) ? px[ps][t] : 0.f;
                csum += vd[t];
            }
            s3d_half8 hi, lo;
            s3d_split8(vd, hi, lo);
            *reinterpret_cast<s3d_half8*>(&s_t[0][0][ch * WL_LD + 16 * ps + 8 * rg]) = hi;
            *reinterpret_cast<s3d_half8*>(&s_t[0][1][ch * WL_LD + 16 * ps + 8 * rg]) = lo;
            s3d_split8(vx, hi, lo);
            *reinterpret_cast<s3d_half8*>(&s_t[1][0][ch * WL_LD + 16 * ps + 8 * rg]) = hi;
            *reinterpret_cast<s3d_half8*>(&s_t[1][1][ch * WL_LD + 16 * ps + 8 * rg]) = lo;
        }
        __syncthreads();
        if (it + 1 < steps_per_split) gload(p_begin + (long)(it + 1) * 32);
        wl_half8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = (64 * wn + 16 * i + m) * WL_LD + 8 * g;
            ah[i] = *reinterpret_cast<const wl_half8*>(&s_t[0][0][o]);
            al[i] = *reinterpret_cast<const wl_half8*>(&s_t[0][1][o]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = (64 * wc + 16 * j + m) * WL_LD + 8 * g;
            bh[j] = *reinterpret_cast<const wl_half8*>(&s_t[1][0][o]);
            bl[j] = *reinterpret_cast<const wl_half8*>(&s_t[1][1][o]);
        }
        // three product kinds, each swept over the 16 independent accumulators (a.single, the S3D_PREC_F16 training throughput
        // mode: the hi * hi sweep alone — a uniform branch; this kernel is bound by its operand staging, not by its MFMAs)
        if (!a.single) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        __syncthreads();
    }
    // D[row = 4g+reg <-> n][col = l&15 <-> c]
    float* part = a.partial + (size_t)split * a.N * a.Cx;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int n = n0 + 64 * wn + 16 * i + 4 * g + reg;
                const int c = c0 + 64 * wc + 16 * j + m;
                if (n < a.N && c < a.Cx) part[(size_t)n * a.Cx + c] = acc[i][j][reg];
            }
    if (a.bias_out != nullptr && cb == 0) {   // the two row groups of a channel meet in LDS (free after the loop's last barrier)
        float* s_c = reinterpret_cast<float*>(&s_t[0][0][0]);
        s_c[tid] = csum;
        __syncthreads();
        if (tid < 128 && n_ok)
            a.partial[(size_t)gridDim.y * a.N * a.Cx + (size_t)split * a.N + n0 + tid] = s_c[tid] + s_c[tid + 128];
    }
}

__global__ void colsum_final_kernel(const float* __restrict__ partial, int nchunks, int C, float scale,
                                    float* __restrict__ out, int accumulate);

// ---------------------------------------------------------------------------------------------
// FFN weight gradients with recomputation (see train.h).
//
// Operand images (layout: decode.h).  Both contraction operands of a 32-row step are needed by the 16 hidden-block workgroups
// that share the rows, as f16 hi/lo, one of them transposed: splitting and transposing them inside this kernel cost it 39 % of
// its time (profiles/r03_ffn_wgrad_ablation.md).  Round 3 built them with a pass of its own (ffn_rec_images_kernel, 3x the
// tensor's bytes per call, 7.6 ms per step); since round 4 the pipelined FFN kernel writes them for the rows it holds anyway.
// ---------------------------------------------------------------------------------------------
// The contraction.  Workgroup = 128 (q) x 128 (hidden) output tile of hidden block hb, rows split over the grid; 4 waves,
// wave w owns hidden units 32w .. 32w+31 of the block against ALL 128 q.  Per 32-row step (one barrier; the images are
// double buffered in LDS):
//   1. the NEXT step's D^T and R images are requested by LDS-DMA (8 KiB per wave, no registers, no VALU);
//   2. the wave recomputes its 2 hidden tiles x 2 row tiles of Z = R W^T on the MFMA (its 16 W fragments stay in
//      registers for the whole kernel; lin1's bias opens the accumulators) and applies the activity bits.  The D
//      registers ARE the B operand of the contraction (see the D^T image's slot order): Z never goes through LDS;
//   3. out[:, own hidden] += D^T Z: the 8 q-tile A fragments come from LDS one tile ahead, 48 MFMAs.
// The dropout scale multiplies the finished sums.
// ---------------------------------------------------------------------------------------------
// LDS fragment reads and their waits are issued by hand (as in ffn_layer_f16x3_pipe_kernel): with an LDS-DMA refill of
// the other buffer in flight hipcc guards compiler-generated ds_reads of the same object with s_waitcnt vmcnt(0) — the
// wait the double buffer exists to avoid — and splitting the buffers into separate objects (an unrolled two-step loop)
// made it shuffle and spill the accumulators.  LDS returns in order: lgkmcnt(n) leaves exactly the n youngest reads out.
#define FWR_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define FWR_WAIT4(n, r0, r1, r2, r3) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(n))
#define FWR_WAIT2(n, r0, r1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(n))
#define FWR_BUF_BYTES (4 * FWR_BLK_HALFS)   // one buffer: D^T image (16 KiB) | R image (16 KiB)

// one block image (16 KiB) -> LDS: wave w copies the 4 KiB at offset 4096 w as four 1-KiB LDS-DMA instructions that
// differ only in their immediate offset (one lane address, one M0 value per image)
__device__ __forceinline__ void fwr_dma(const _Float16* gblk, _Float16* lbuf, int wave, int lane) {
    const __attribute__((address_space(1))) void* g = (const __attribute__((address_space(1))) void*)(gblk + wave * 2048 + lane * 8);
    __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(lbuf + wave * 2048);
    __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
}

// SINGLE (round 6, the training step's single-pass f16 throughput mode, S3D_PREC_F16): only the hi * hi product of every split
// — the low halves of the images are neither read from LDS nor multiplied; not fp32-class, never the headline mode.
template <int COLSUM, bool SINGLE>
__global__ __launch_bounds__(256, 2) void ffn_wgrad_rec_kernel(const FfnWgradArgs a, int steps_per_split, int nsplit) {
    __shared__ __attribute__((aligned(16))) _Float16 s_t[2][2][FWR_BLK_HALFS];   // [buffer][D^T | R] hi|lo images, 64 KiB
    __shared__ __attribute__((aligned(16))) unsigned s_m[2][128];                 // [buffer][g'][row] mask dwords of hb
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g = lane >> 4;
    // 1-D grid of 16 hidden blocks x nsplit row ranges.  Workgroups are dealt to the 8 XCDs round-robin; remap so
    // that the 16 hidden blocks of one row range run on ONE XCD back to back: they read the same images, which
    // then come from that XCD's L2 once instead of from the fabric 16 times.
    int hb, split;
    {
        const int L = blockIdx.x, nb = S3D_FFN / 128;
        if (nsplit % 8 == 0) {
            const int xcd = L & 7, k = L >> 3;
            hb = k % nb;
            split = (k / nb) * 8 + xcd;
        } else {
            hb = L % nb;
            split = L / nb;
        }
    }
    const long P = a.P;
    const long blk0 = (long)split * steps_per_split;                 // first 32-row block of this range
    const long nblk_all = (P + 31) / 32;
    const int steps = (int)(nblk_all - blk0 < steps_per_split ? nblk_all - blk0 : steps_per_split);   // >= 1
    const _Float16* dimg = reinterpret_cast<const _Float16*>(a.Dimg) + blk0 * FWR_BLK_HALFS;
    const _Float16* rimg = reinterpret_cast<const _Float16*>(a.Rimg) + blk0 * FWR_BLK_HALFS;

    // this wave's W fragments: hidden tiles 2*wave + e of the block, K = 128 = 4 x 32
    wl_half8 wh[2][4], wlo[2][4];
    float bv[2];
    {
        const _Float16* img = reinterpret_cast<const _Float16*>(a.wimg);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int tile = hb * 8 + 2 * wave + e;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t o = ((size_t)(tile * 4 + u) * 64 + lane) * 8;
                wh[e][u] = *reinterpret_cast<const wl_half8*>(img + o);
                wlo[e][u] = SINGLE ? wh[e][u] : *reinterpret_cast<const wl_half8*>(img + (size_t)S3D_FFN * 128 + o);
            }
            bv[e] = a.bias ? a.bias[tile * 16 + m] : 0.f;
        }
    }
    f32x4 acc[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) acc[i][e] = zero4();
    float csum[2] = {0.f, 0.f};
    const int bitpos0 = 4 * wave + FFN_MASK_POS(m & 3);   // + e: the D tile (decode.h)

    // lane byte addresses inside buffer 0: A fragment of q tile 0 (tile i at +1024 i, lo half at +8192), R fragment of K
    // step u, row tile 0 (row tile 1 at +4096, lo half at +8192; the R image follows the D^T image)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&s_t[0][0][0];
    const unsigned a_addr0 = lds0 + 2u * (unsigned)(m * 32 + ((g ^ fwr_dperm(m)) << 3));
    unsigned x_addr0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x_addr0[u] = lds0 + 2u * FWR_BLK_HALFS + 2u * (unsigned)(m * 128 + (((4 * u + g) ^ m) << 3));

    // the step's mask dwords: thread (g' = tid >> 5, row = tid & 31) of the first 128 threads; rows past the end carry 0
    const unsigned* mbase = a.mask + blk0 * 32 * 64;                                           // uniform
    const unsigned moff = (unsigned)ffn_mask_dword(tid & 31, hb, (tid >> 5) & 3);               // lane offset (layout: decode.h)
    const int rows_left = (int)(P - blk0 * 32 < (long)steps_per_split * 32 ? P - blk0 * 32 : (long)steps_per_split * 32);
    auto mask_of = [&](int step) -> unsigned {
        return step * 32 + (tid & 31) < rows_left ? (mbase + (size_t)step * 32 * 64)[moff] : 0u;
    };
    fwr_dma(dimg, s_t[0][0], wave, lane);
    fwr_dma(rimg, s_t[0][1], wave, lane);
    if (tid < 128) s_m[0][tid] = mask_of(0);
    dma_publish_barrier();
#pragma unroll 1
    for (int it = 0; it < steps; ++it) {
        const int cur = it & 1;
        const bool more = it + 1 < steps;
        unsigned pm = 0;
        if (more) {   // ---- 1. next step's images by LDS-DMA into the other buffer ----
            fwr_dma(dimg + (size_t)(it + 1) * FWR_BLK_HALFS, s_t[cur ^ 1][0], wave, lane);
            fwr_dma(rimg + (size_t)(it + 1) * FWR_BLK_HALFS, s_t[cur ^ 1][1], wave, lane);
            pm = mask_of(it + 1);
        }
        const unsigned boff = (unsigned)cur * FWR_BUF_BYTES;
        const unsigned aa = a_addr0 + boff;
        __builtin_amdgcn_sched_barrier(0);
        // ---- 2. recompute Z tiles (rt, e): D[row 4g+i of tile rt][hidden m of tile e]; the fragments of K step u+1
        //      are requested before the MFMAs of step u ----
        f32x4 z[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int e = 0; e < 2; ++e) z[rt][e] = f32x4{bv[e], bv[e], bv[e], bv[e]};
        wl_half8 ah[2], al[2];
        {
            wl_half8 xh[2][2], xl[2][2];
            {
                const unsigned xa = x_addr0[0] + boff;
                if (SINGLE) {
                    FWR_READ(xh[0][0], xa, 0); FWR_READ(xh[0][1], xa, 4096);
                } else {
                    FWR_READ(xh[0][0], xa, 0); FWR_READ(xl[0][0], xa, 8192); FWR_READ(xh[0][1], xa, 4096); FWR_READ(xl[0][1], xa, 12288);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cs = u & 1, ns = cs ^ 1;
                if (SINGLE) {   // (half the reads: the counted waits leave the next step's two / the A fragment's one outstanding)
                    if (u < 3) {
                        const unsigned xa = x_addr0[u + 1] + boff;
                        FWR_READ(xh[ns][0], xa, 0); FWR_READ(xh[ns][1], xa, 4096);
                        FWR_WAIT2(2, xh[cs][0], xh[cs][1]);
                    } else {
                        FWR_READ(ah[0], aa, 0);
                        FWR_WAIT2(1, xh[cs][0], xh[cs][1]);
                    }
                } else if (u < 3) {
                    const unsigned xa = x_addr0[u + 1] + boff;
                    FWR_READ(xh[ns][0], xa, 0); FWR_READ(xl[ns][0], xa, 8192); FWR_READ(xh[ns][1], xa, 4096); FWR_READ(xl[ns][1], xa, 12288);
                    FWR_WAIT4(4, xh[cs][0], xl[cs][0], xh[cs][1], xl[cs][1]);
                } else {
                    FWR_READ(ah[0], aa, 0); FWR_READ(al[0], aa, 8192);
                    FWR_WAIT4(2, xh[cs][0], xl[cs][0], xh[cs][1], xl[cs][1]);
                }
                if (!SINGLE) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int e = 0; e < 2; ++e) z[rt][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[cs][rt], wlo[e][u], z[rt][e], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int e = 0; e < 2; ++e) z[rt][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[cs][rt], wh[e][u], z[rt][e], 0, 0, 0);
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int e = 0; e < 2; ++e) z[rt][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[cs][rt], wh[e][u], z[rt][e], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // activity bit, split: hidden unit (block-local) 32*wave + 16e + m -> bit 4*wave + e + FFN_MASK_POS(m & 3) of dword
        // g' = m >> 2 of the row; this lane's rows are 4g..4g+3 (tile 0) and 16+4g.. (tile 1)
        wl_half8 zh[2], zl[2];
        {
            const uint4 mw0 = *reinterpret_cast<const uint4*>(&s_m[cur][(m >> 2) * 32 + 4 * g]);
            const uint4 mw1 = *reinterpret_cast<const uint4*>(&s_m[cur][(m >> 2) * 32 + 16 + 4 * g]);
            const unsigned mw[8] = {mw0.x, mw0.y, mw0.z, mw0.w, mw1.x, mw1.y, mw1.z, mw1.w};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    v[t] = s3d_gate_bit(z[t >> 2][e][t & 3], mw[t], bitpos0 + e);
                    if (COLSUM) csum[e] += v[t];
                }
                s3d_split8(v, zh[e], zl[e]);
            }
        }
        S3D_SPLIT_SETTLE();   // partial-register split results feed the MFMAs below straight from registers
        // ---- 3. out[q tile i][own hidden tile e] += D^T Z, the A fragments one tile ahead ----
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cs = i & 1, ns = cs ^ 1;
            if (SINGLE) {
                if (i < 7) {
                    FWR_READ(ah[ns], aa, (i + 1) * 1024);
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ah[cs]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[cs]));
                }
            } else if (i < 7) {
                FWR_READ(ah[ns], aa, (i + 1) * 1024); FWR_READ(al[ns], aa, (i + 1) * 1024 + 8192);
                FWR_WAIT2(2, ah[cs], al[cs]);
            } else {
                FWR_WAIT2(0, ah[cs], al[cs]);
            }
            if (!SINGLE) {
#pragma unroll
                for (int e = 0; e < 2; ++e) acc[i][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cs], zl[e], acc[i][e], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 2; ++e) acc[i][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cs], zh[e], acc[i][e], 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) acc[i][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cs], zh[e], acc[i][e], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tid < 128) s_m[cur ^ 1][tid] = pm;
        dma_publish_barrier();   // the other buffer's images have landed; everyone is done with this one
    }
    // partial[split][q][hidden]: D[row = q 4g+reg][col = hidden m]
    float* part = a.partial + (size_t)split * 128 * S3D_FFN;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int q = 16 * i + 4 * g + reg;
                const int hid = hb * 128 + 32 * wave + 16 * e + m;
                part[(size_t)q * S3D_FFN + hid] = acc[i][e][reg] * a.scale;
            }
    if (COLSUM) {
        float* cp = a.partial + (size_t)nsplit * 128 * S3D_FFN + (size_t)split * S3D_FFN;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float v = csum[e] * a.scale;
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0) cp[hb * 128 + (2 * wave + e) * 16 + m] = v;
        }
    }
}

__global__ void ffn_wgrad_rec_reduce_kernel(const float* __restrict__ partial, int nsplit, float* __restrict__ out,
                                            int transpose_out, int accumulate) {
    const long total = 128L * S3D_FFN;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        int sp = 0;
        for (; sp + 8 <= nsplit; sp += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partial[(size_t)(sp + k) * total + idx];
            __builtin_amdgcn_sched_barrier(0);
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; sp < nsplit; ++sp) s += partial[(size_t)sp * total + idx];
        const long q = idx / S3D_FFN, hid = idx - q * S3D_FFN;
        const long o = transpose_out ? hid * 128 + q : idx;
        out[o] = accumulate ? out[o] + s : s;
    }
}

int launch_ffn_wgrad_rec(const FfnWgradArgs& a, hipStream_t stream) {
    if (a.P <= 0) return 0;
    S3D_CHECK_ARG(a.Dimg && a.Rimg && a.wimg && a.mask && a.out && a.partial, "ffn_wgrad_rec: null argument");
    const long total_steps = (a.P + 31) / 32;
    long splits = 64;                                          // 16 hidden blocks x 64 = 1024 workgroups
    const long max_by_steps = (total_steps + 7) / 8;
    if (splits > max_by_steps) splits = max_by_steps;
    const long cap = (long)(a.partial_floats / ((size_t)129 * S3D_FFN));
    if (splits > cap) splits = cap;
    if (splits < 1) {
        s3d_set_error("ffn_wgrad_rec: partial workspace too small");
        return S3D_E_WORKSPACE;
    }
    const int spw = (int)((total_steps + splits - 1) / splits);
    splits = (total_steps + spw - 1) / spw;
    dim3 grid((unsigned)((S3D_FFN / 128) * splits));
    if (a.single) {
        if (a.bias_out)
            hipLaunchKernelGGL((ffn_wgrad_rec_kernel<1, true>), grid, dim3(256), 0, stream, a, spw, (int)splits);
        else
            hipLaunchKernelGGL((ffn_wgrad_rec_kernel<0, true>), grid, dim3(256), 0, stream, a, spw, (int)splits);
    } else if (a.bias_out)
        hipLaunchKernelGGL((ffn_wgrad_rec_kernel<1, false>), grid, dim3(256), 0, stream, a, spw, (int)splits);
    else
        hipLaunchKernelGGL((ffn_wgrad_rec_kernel<0, false>), grid, dim3(256), 0, stream, a, spw, (int)splits);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(ffn_wgrad_rec_reduce_kernel, dim3(1024), dim3(256), 0, stream, a.partial, (int)splits, a.out,
                       a.transpose_out, a.accumulate);
    S3D_LAUNCH_CHECK();
    if (a.bias_out) {
        hipLaunchKernelGGL(colsum_final_kernel, dim3(S3D_FFN / 64), dim3(1024), 0, stream,
                           a.partial + (size_t)splits * 128 * S3D_FFN, (int)splits, S3D_FFN, 1.f, a.bias_out,
                           a.accumulate);
        S3D_LAUNCH_CHECK();
    }
    return 0;
}

// element (hid, k) = w[hid*sh + k*sk]; lane (hl = l & 15, g = l >> 4) of fragment (tile, u) holds k = 32u + 8g .. +7
__global__ void pack_ffn_rec_f16x3_kernel(const float* __restrict__ w, int sh, int sk, _Float16* __restrict__ out) {
    const int total = (S3D_FFN / 16) * 4 * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, u = (idx >> 6) & 3, tile = idx >> 8;
        const int hid = tile * 16 + (lane & 15), k0 = 32 * u + 8 * (lane >> 4);
        _Float16* dst = out + (size_t)idx * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float v = w[(size_t)hid * sh + (size_t)(k0 + t) * sk];
            const _Float16 h = (_Float16)v;
            dst[t] = h;
            dst[(size_t)S3D_FFN * 128 + t] = (_Float16)(v - (float)h);
        }
    }
}
int launch_pack_ffn_rec_f16x3(const float* w, int sh, int sk, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(pack_ffn_rec_f16x3_kernel, dim3(128), dim3(256), 0, stream, w, sh, sk,
                       reinterpret_cast<_Float16*>(out));
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Split-precision weight gradient of a 3x3 convolution (stride 1, padding 1):
//   dW[n][ky][kx][c] = sum over pixels p of dY[p][n] * X[p + (ky-1, kx-1)][c]
// Workgroup = 64 (n) x 64 (c) x all 9 taps, 4 waves of 32 x 32 x 9 (36 accumulator tiles each); the contraction
// runs over 4-row x 8-column pixel tiles, one v_mfma_f32_16x16x32_f16 K per tile (k-slot g <-> tile row, the
// lane's 8 halfs <-> the 8 columns).  Per tile the workgroup stages, as f16 hi/lo and channel-major,
//   dY^T [n][4 rows][8 cols]                 and
//   X^T  [c][6 halo rows][col -1 | 8 cols | col 8]   (interior 16-byte aligned at half 8, neighbours at 7 and 16)
// once, and all nine taps read from it: the row shift picks another halo row, the column shift is made in
// registers from the aligned word and its two neighbour dwords (v_alignbyte).  dY / X are read once per tile
// instead of once per tap; next tile's global loads are in flight during the MFMAs.
// ---------------------------------------------------------------------------------------------
#define W3_DLD 48     // halfs per n row of the dY^T tile (4 x 8 + 16 pad: 96 B = 32 mod 64, conflict-free b128 reads)
#define W3_XROW 24    // halfs per halo row
#define W3_XLD 144    // halfs per channel of the X^T tile (6 x 24: 288 B = 32 mod 64)
typedef int w3_int4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wl_half8 w3_frag(int d0, int d1, int d2, int d3) {
    w3_int4 v = {d0, d1, d2, d3};
    return __builtin_bit_cast(wl_half8, v);
}
// WNT x WCT accumulator tile pairs per wave per tap, WVN x WVC waves: block = 16*WNT*WVN (n) x 16*WCT*WVC (c).
// <2,2,2,2>: 64 x 64 (channel counts that are multiples of 64); <2,1,2,2>: 64 x 32; <1,1,4,1>: 64 x 16 (the 3 -> 16
// padded first layer, which is bound by reading dY: the per-tap kernel read it nine times).  N = 32 uses half a block.
template <int WNT, int WCT, int WVN, int WVC>
__global__ __launch_bounds__(256, 2) void wgrad_conv3_f16x3_kernel(const WgradArgs a, int n_cblk, long n_tiles,
                                                                   int tiles_per_split, int Ktot) {
    constexpr int NB = 16 * WNT * WVN, CB = 16 * WCT * WVC;
    static_assert(WVN * WVC == 4 && NB == 64, "4 waves, 64 dY channels per block");
    constexpr int XU = CB * 6;                        // (channel, halo row) staging units
    constexpr int XK = (XU + 255) / 256;              // units per thread
    __shared__ __attribute__((aligned(16))) _Float16 s_d[2][NB * W3_DLD];   // dY^T hi|lo
    __shared__ __attribute__((aligned(16))) _Float16 s_x[2][CB * W3_XLD];   // X^T  hi|lo
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int ch = tid & 63, slot = tid >> 6;   // dY staging role: channel, tile row
    const int cb = blockIdx.x % n_cblk, nb = blockIdx.x / n_cblk;
    const int n0 = nb * NB, c0 = cb * CB;
    const int wn = wave % WVN, wc = wave / WVN;
    const int H = a.H, W = a.W;
    const int tiles_x = W >> 3, tpi = tiles_x * (H >> 2);
    const bool nvalid = n0 + ch < a.N;   // N = 32 layers run as a half-empty 64-row block
    const float* dyp = a.dy + a.dy_coff + n0 + (nvalid ? ch : 0);
    const long dstride = a.dy_cstride, xstride = a.x.C;
    // X staging role k: unit u = tid + 256 k -> (channel, halo row); threads past the last unit shadow unit 0
    int xc[XK], xr[XK];
    bool xact[XK];
#pragma unroll
    for (int k = 0; k < XK; ++k) {
        const int u = tid + 256 * k;
        xact[k] = u < XU;
        xc[k] = xact[k] ? u % CB : 0;
        xr[k] = xact[k] ? u / CB : 0;
    }
    const float* xp = a.x.p + a.x_coff + c0;

    f32x4 acc[9][WNT][WCT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < WNT; ++i)
#pragma unroll
            for (int j = 0; j < WCT; ++j) acc[t][i][j] = zero4();

    // raw prefetch registers (unconditional clamped loads; validity becomes a 0/1 factor at conversion time)
    float pd[8], px[XK][10];
    unsigned vm = 0;   // bit 0 tile valid, 1 + k halo row of unit k inside the image, 3 left column valid, 4 right column valid
    const long t_begin = (long)blockIdx.y * tiles_per_split;
    long t_end = t_begin + tiles_per_split;
    if (t_end > n_tiles) t_end = n_tiles;
    auto gload = [&](long t) {
        const bool tv = t < t_end;
        const long tc = tv ? t : t_end - 1;
        const int img = (int)(tc / tpi);
        const int r = (int)(tc - (long)img * tpi);
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int ni = (a.x.bmod ? img % a.x.bmod : img) / a.x.bdiv;
        {
            const float* p = dyp + ((long)(img * H + ty * 4 + slot) * W + tx * 8) * dstride;
#pragma unroll
            for (int j = 0; j < 8; ++j) pd[j] = p[j * dstride];
        }
        const bool xl = tx > 0, xrt = tx + 1 < tiles_x;
        vm = (tv ? 1u : 0u) | (xl ? 8u : 0u) | (xrt ? 16u : 0u);
#pragma unroll
        for (int k = 0; k < XK; ++k) {
            const int y = ty * 4 - 1 + xr[k];
            const bool yok = y >= 0 && y < H;
            vm |= yok ? (2u << k) : 0u;
            const int yc = min(max(y, 0), H - 1);
            const float* p = xp + xc[k] + (long)(ni * H + yc) * W * xstride;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int x = min(max(tx * 8 - 1 + j, 0), W - 1);
                px[k][j] = p[x * xstride];
            }
        }
    };
    if (t_begin < t_end) gload(t_begin);
    for (long t = t_begin; t < t_end; ++t) {
        // ---- stage the tile ----
        {
            const float tvf = (vm & 1u) ? 1.f : 0.f;
            const float dvf = nvalid ? tvf : 0.f;
            wl_half8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = pd[j] * dvf;
                const _Float16 h = (_Float16)v;
                hi[j] = h;
                lo[j] = (_Float16)(v - (float)h);
            }
            *reinterpret_cast<wl_half8*>(&s_d[0][ch * W3_DLD + 8 * slot]) = hi;
            *reinterpret_cast<wl_half8*>(&s_d[1][ch * W3_DLD + 8 * slot]) = lo;
#pragma unroll
            for (int k = 0; k < XK; ++k) {
                const float rf = (vm & (2u << k)) ? tvf : 0.f;
                const float lf = (vm & 8u) ? rf : 0.f, rtf = (vm & 16u) ? rf : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = px[k][j + 1] * rf;
                    const _Float16 h = (_Float16)v;
                    hi[j] = h;
                    lo[j] = (_Float16)(v - (float)h);
                }
                const float vl = px[k][0] * lf, vr = px[k][9] * rtf;
                const _Float16 hl = (_Float16)vl, hr = (_Float16)vr;
                if (xact[k]) {
                    const int o = xc[k] * W3_XLD + xr[k] * W3_XROW;
                    *reinterpret_cast<wl_half8*>(&s_x[0][o + 8]) = hi;
                    *reinterpret_cast<wl_half8*>(&s_x[1][o + 8]) = lo;
                    s_x[0][o + 7] = hl;
                    s_x[1][o + 7] = (_Float16)(vl - (float)hl);
                    s_x[0][o + 16] = hr;
                    s_x[1][o + 16] = (_Float16)(vr - (float)hr);
                }
            }
        }
        __syncthreads();
        if (t + 1 < t_end) gload(t + 1);
        // ---- 9 taps x (WNT x WCT) tiles ----
        wl_half8 ah[WNT], al[WNT];
#pragma unroll
        for (int i = 0; i < WNT; ++i) {
            const int o = (16 * (wn * WNT + i) + m) * W3_DLD + 8 * g;
            ah[i] = *reinterpret_cast<const wl_half8*>(&s_d[0][o]);
            al[i] = *reinterpret_cast<const wl_half8*>(&s_d[1][o]);
        }
#pragma unroll
        for (int dyi = 0; dyi < 3; ++dyi) {
#pragma unroll
            for (int j = 0; j < WCT; ++j) {
                const int o = (16 * (wc * WCT + j) + m) * W3_XLD + (g + dyi) * W3_XROW;
                wl_half8 bh[3], bl[3];
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    const w3_int4 I = *reinterpret_cast<const w3_int4*>(&s_x[hl][o + 8]);
                    const int Pw = *reinterpret_cast<const int*>(&s_x[hl][o + 6]);
                    const int Nw = *reinterpret_cast<const int*>(&s_x[hl][o + 16]);
                    const int s01 = __builtin_amdgcn_alignbyte(I[1], I[0], 2);
                    const int s12 = __builtin_amdgcn_alignbyte(I[2], I[1], 2);
                    const int s23 = __builtin_amdgcn_alignbyte(I[3], I[2], 2);
                    const wl_half8 f0 = w3_frag(__builtin_amdgcn_alignbyte(I[0], Pw, 2), s01, s12, s23);
                    const wl_half8 f1 = __builtin_bit_cast(wl_half8, I);
                    const wl_half8 f2 = w3_frag(s01, s12, s23, __builtin_amdgcn_alignbyte(Nw, I[3], 2));
                    if (hl == 0) {
                        bh[0] = f0; bh[1] = f1; bh[2] = f2;
                    } else {
                        bl[0] = f0; bl[1] = f1; bl[2] = f2;
                    }
                }
#pragma unroll
                for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
                    for (int i = 0; i < WNT; ++i) {
                        f32x4 c = acc[dyi * 3 + dxi][i][j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[dxi], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[dxi], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[dxi], c, 0, 0, 0);
                        acc[dyi * 3 + dxi][i][j] = c;
                    }
            }
        }
        __syncthreads();
    }
    // D[row = 4g+reg <-> n][col = m <-> c]  ->  partial[split][n][tap*Cx + c]
    float* part = a.partial + (size_t)blockIdx.y * a.N * Ktot;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int i = 0; i < WNT; ++i)
#pragma unroll
            for (int j = 0; j < WCT; ++j)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int n = n0 + 16 * (wn * WNT + i) + 4 * g + reg;
                    const int c = c0 + 16 * (wc * WCT + j) + m;
                    if (n < a.N) part[(size_t)n * Ktot + tap * a.Cx + c] = acc[tap][i][j][reg];
                }
}

static bool wgrad_conv3_eligible(const WgradArgs& a) {
    return a.prec == S3D_PREC_F16X3 && a.ks == 3 && a.stride <= 1 && !a.x.sbcast && a.x.bdiv >= 1 &&
           (a.Hin == 0 || a.Hin == a.H) && (a.Win == 0 || a.Win == a.W) && (a.N % 64 == 0 || a.N == 32) &&
           (a.Cx % 64 == 0 || a.Cx == 32 || a.Cx == 16) && a.H % 4 == 0 && a.W % 8 == 0 &&
           a.out_kind == S3D_PACK_CONV;
}

static int launch_wgrad_conv3_f16x3(const WgradArgs& a, hipStream_t stream) {
    const bool narrow = a.Cx == 16, half = a.Cx == 32;
    const int n_nblk = (a.N + 63) / 64, n_cblk = narrow ? 1 : (half ? 1 : a.Cx / 64), Ktot = 9 * a.Cx;
    const long n_tiles = (long)a.Nimg * (a.H / 4) * (a.W / 8);
    const long blocks = (long)n_nblk * n_cblk;
    constexpr long want_wgs = 1024;
    long splits = (want_wgs + blocks - 1) / blocks;            // aim at >= 1024 workgroups (4 per CU): more only grows the partial-sum reduction
    const long max_by_tiles = (n_tiles + 7) / 8;               // >= 8 tiles per workgroup
    if (splits > max_by_tiles) splits = max_by_tiles;
    const long cap = (long)(a.partial_floats / ((size_t)a.N * Ktot));
    if (splits > cap) splits = cap;
    if (splits < 1) {
        s3d_set_error("wgrad: partial workspace too small for %d x %d x 9", a.N, a.Cx);
        return S3D_E_WORKSPACE;
    }
    const int tps = (int)((n_tiles + splits - 1) / splits);
    splits = (n_tiles + tps - 1) / tps;
    const dim3 grid((unsigned)blocks, (unsigned)splits);
    if (narrow)
        hipLaunchKernelGGL((wgrad_conv3_f16x3_kernel<1, 1, 4, 1>), grid, dim3(256), 0, stream, a, n_cblk, n_tiles, tps,
                           Ktot);
    else if (half)   // the U-Net's last stage: 32-channel inputs (and 32 outputs, a half-empty n block)
        hipLaunchKernelGGL((wgrad_conv3_f16x3_kernel<2, 1, 2, 2>), grid, dim3(256), 0, stream, a, n_cblk, n_tiles, tps,
                           Ktot);
    else
        hipLaunchKernelGGL((wgrad_conv3_f16x3_kernel<2, 2, 2, 2>), grid, dim3(256), 0, stream, a, n_cblk, n_tiles, tps,
                           Ktot);
    S3D_LAUNCH_CHECK();
    const long total = (long)a.N * Ktot;
    const int rb = (int)((total + 63) / 64 < 8192 ? (total + 63) / 64 : 8192);
    hipLaunchKernelGGL(wgrad_reduce_kernel<false>, dim3(rb), dim3(256), 0, stream, a, (int)splits, Ktot);
    S3D_LAUNCH_CHECK();
    return 0;
}

static bool wgrad_lin_eligible(const WgradArgs& a, long P) {
    return a.prec == S3D_PREC_F16X3 && a.ks == 1 && a.stride <= 1 && !a.x.sbcast && !a.x.bmod && a.x.bdiv == 1 &&
           (a.Hin == 0 || a.Hin == a.H) && (a.Win == 0 || a.Win == a.W) && a.N >= 64 && a.Cx >= 64 && P >= 1024;
}

static int launch_wgrad_lin_f16x3(const WgradArgs& a, long P, hipStream_t stream) {
    const int n_nblk = (a.N + 127) / 128, n_cblk = (a.Cx + 127) / 128;
    const long total_steps = (P + 31) / 32;
    const long tiles = (long)n_nblk * n_cblk;
    long splits = (1024 + tiles - 1) / tiles;                 // aim at >= 1024 workgroups
    const long max_by_steps = (total_steps + 7) / 8;          // >= 8 K-steps per workgroup
    if (splits > max_by_steps) splits = max_by_steps;
    const long cap = (long)(a.partial_floats / ((size_t)a.N * a.Cx + (a.bias_out ? a.N : 0)));
    if (splits > cap) splits = cap;
    if (splits < 1) {
        s3d_set_error("wgrad: partial workspace too small for %d x %d", a.N, a.Cx);
        return S3D_E_WORKSPACE;
    }
    const int spw = (int)((total_steps + splits - 1) / splits);
    splits = (total_steps + spw - 1) / spw;
    hipLaunchKernelGGL(wgrad_lin_f16x3_kernel, dim3((unsigned)(n_nblk * n_cblk), (unsigned)splits), dim3(256), 0, stream,
                       a, n_cblk, P, spw);
    S3D_LAUNCH_CHECK();
    const long total = (long)a.N * a.Cx + (a.bias_out ? a.N : 0);
    const int blocks = (int)((total + 63) / 64 < 8192 ? (total + 63) / 64 : 8192);
    if (a.bias_out)
        hipLaunchKernelGGL(wgrad_reduce_kernel<true>, dim3(blocks), dim3(256), 0, stream, a, (int)splits, a.Cx);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<false>, dim3(blocks), dim3(256), 0, stream, a, (int)splits, a.Cx);
    S3D_LAUNCH_CHECK();
    return 0;
}

static void wgrad_plan(long P, int N, int Cx, int taps, int& TN, int& TK, int& n_nblk, int& n_cblk, int& splits_y,
                       int& steps_per_wave) {
    TN = N >= 64 ? 4 : N >= 32 ? 2 : 1;
    TK = Cx >= 64 ? 4 : Cx >= 32 ? 2 : 1;
    n_nblk = (N + 16 * TN - 1) / (16 * TN);
    n_cblk = (Cx + 16 * TK - 1) / (16 * TK);
    const long total_steps = (P + 3) / 4;
    const long tiles = (long)n_nblk * n_cblk * taps;
    long sy = (2048 + tiles - 1) / tiles;                       // aim at >= 1024 workgroups (4 per CU): more only grows the partial-sum reduction
    const long max_by_steps = (total_steps + 4 * 16 - 1) / (4 * 16);  // >= 16 MFMA steps per wave
    if (sy > max_by_steps) sy = max_by_steps;
    const long cap = (64L << 20) / ((long)N * Cx * taps * 4);   // partial buffer <= 64 Mi floats
    if (sy > cap) sy = cap;
    if (sy < 1) sy = 1;
    splits_y = (int)sy;
    steps_per_wave = (int)((total_steps + sy * 4 - 1) / (sy * 4));
}

size_t wgrad_partial_floats(long P, int N, int Ktot) {
    // upper bound used by workspace planners: plan with the whole K as one segment
    int TN, TK, nn, nc, sy, spw;
    wgrad_plan(P, N, Ktot, 1, TN, TK, nn, nc, sy, spw);
    return (size_t)sy * 4 * N * Ktot;
}

int launch_wgrad(const WgradArgs& a, hipStream_t stream) {
    S3D_CHECK_ARG(a.ks >= 1 && a.ks <= 3 && a.N > 0 && a.Cx > 0, "wgrad: bad dims");
    S3D_CHECK_ARG(a.dy_cstride % 4 == 0 && a.dy_coff % 4 == 0 && a.x.C % 4 == 0 && a.x_coff % 4 == 0,
                  "wgrad: channel strides/offsets must be multiples of 4");
    const long P = (long)a.Nimg * a.H * a.W;
    if (wgrad_lin_eligible(a, P)) return launch_wgrad_lin_f16x3(a, P, stream);
    if (a.bias_out != nullptr) {
        S3D_CHECK_ARG(a.bias_partial != nullptr, "wgrad: bias_out needs bias_partial");
        const int rc = launch_colsum(a.dy, P, a.dy_cstride, a.dy_coff, a.N, a.bias_out, a.accumulate, a.bias_partial, stream);
        if (rc) return rc;
    }
    if (wgrad_conv3_eligible(a)) return launch_wgrad_conv3_f16x3(a, stream);
    const int taps = a.ks * a.ks;
    int TN, TK, n_nblk, n_cblk, sy, spw;
    wgrad_plan(P, a.N, a.Cx, taps, TN, TK, n_nblk, n_cblk, sy, spw);
    const int Ktot = taps * a.Cx;
    const size_t need = (size_t)sy * 4 * a.N * Ktot;
    if (need > a.partial_floats) {
        s3d_set_error("wgrad: partial workspace %zu < %zu floats", a.partial_floats, need);
        return S3D_E_WORKSPACE;
    }
    dim3 grid((unsigned)(n_nblk * taps * n_cblk), (unsigned)sy), block(256);
#define WG_CASE(tn, tk) \
    if (TN == tn && TK == tk) hipLaunchKernelGGL((wgrad_kernel<tn, tk>), grid, block, 0, stream, a, n_cblk, taps, P, spw, Ktot)
    WG_CASE(4, 4); WG_CASE(4, 2); WG_CASE(4, 1); WG_CASE(2, 4); WG_CASE(2, 2); WG_CASE(2, 1);
    WG_CASE(1, 4); WG_CASE(1, 2); WG_CASE(1, 1);
#undef WG_CASE
    S3D_LAUNCH_CHECK();
    const long total = (long)a.N * Ktot;
    const int blocks = (int)((total + 63) / 64 < 8192 ? (total + 63) / 64 : 8192);
    hipLaunchKernelGGL(wgrad_reduce_kernel<false>, dim3(blocks), dim3(256), 0, stream, a, sy * 4, Ktot);
    S3D_LAUNCH_CHECK();
    return 0;
}

__device__ __forceinline__ f32x4 bn_relu4(const f32x4 z, const f32x4 mu, const f32x4 rs, const f32x4 ga,
                                          const f32x4 be) {
    f32x4 y = (z - mu) * rs * ga + be;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = fmaxf(y[i], 0.f);
    return y;
}

// Pilot value of the one-pass BatchNorm statistics: the mean of 16 rows at Fibonacci-hashed positions (not a single
// row: row 0 is an image corner, whose zero-padded conv output can sit many sigmas from the channel mean).  Every
// workgroup and the finalize kernel add the same rows in the same order, so they all hold the same bits.
#define BN_PILOT_ROWS 16
__device__ __forceinline__ long bn_pilot_row(int j, long P) {
    const unsigned long long h = (unsigned)(j * 2654435769u + 1327217885u);
    return (long)((h * (unsigned long long)P) >> 32);
}
__device__ __forceinline__ f32x4 bn_pilot4(const float* __restrict__ col, long P, int cstride) {
    f32x4 k = zero4();
#pragma unroll 4
    for (int j = 0; j < BN_PILOT_ROWS; ++j) k += ld4(col + bn_pilot_row(j, P) * cstride);
    return k * (1.f / BN_PILOT_ROWS);
}

// =============================================================================================
// column reductions (deterministic two-stage).  mode 0: sum x ; 1: sum (x-m[c])^2 ;
// 2: two sums at once: sum g and sum g*(z-m[c])*r[c]   (BN backward; g in `in`, z in `in2`)
// 3: as 2 with g = relu'(bn(z)) * dy recomputed from the BN parameters (dy in `in`): no g tensor in memory
// 4: two sums at once about a pilot value: sum (x-k[c]) and sum (x-k[c])^2 with k = bn_pilot4 of `in` (one-pass
//    BatchNorm statistics: the shift keeps the E[d^2] - E[d]^2 cancellation at the size of (mean-k)^2 / var)
// =============================================================================================
#define CS_CHUNKS CS_CHUNKS_MAX
template <int MODE>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, const float* __restrict__ in2,
                                                     long P, int cstride, int coff, int C,
                                                     const float* __restrict__ m, const float* __restrict__ r,
                                                     float* __restrict__ partial, long rows_per_chunk,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta) {
    // thread = (row lane rl, column quad cq): narrow matrices (C = 64..128) still use all 256 threads
    __shared__ f32x4 red[256], red2[MODE >= 2 ? 256 : 1];
    const int c4n = C >> 2;
    const int tpr = c4n < 256 ? c4n : 256;
    const int rl_n = 256 / tpr;
    const int rl = threadIdx.x / tpr, ct = threadIdx.x - rl * tpr;
    const long r0 = (long)blockIdx.x * rows_per_chunk;
    long r1 = r0 + rows_per_chunk;
    if (r1 > P) r1 = P;
    for (int cq0 = 0; cq0 < c4n; cq0 += tpr) {
        const int cq = cq0 + ct;
        const bool active = rl < rl_n && cq < c4n;
        const int c = (cq < c4n ? cq : 0) * 4;
        f32x4 s = zero4(), s2 = zero4();
        f32x4 mv = zero4(), rv = zero4();
        f32x4 gav = zero4(), bev = zero4();
        if (MODE >= 1 && MODE <= 3) mv = ld4(m + c);
        if (MODE == 2 || MODE == 3) rv = ld4(r + c);
        if (MODE == 4) mv = bn_pilot4(in + coff + c, P, cstride);
        if (MODE == 3) {
            gav = ld4(gamma + c);
            bev = ld4(beta + c);
        }
        auto masked = [&](f32x4 d, const f32x4 zz) {   // MODE 3: dy -> g
            const f32x4 y = bn_relu4(zz, mv, rv, gav, bev);
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = y[i] > 0.f ? d[i] : 0.f;
            return d;
        };
        if (active) {
            const float* src = in + coff + c;
            if (MODE == 0) {   // 8 independent row streams: 8 loads in flight per thread
                f32x4 t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = zero4();
                long p = r0 + rl;
                for (; p + 7L * rl_n < r1; p += 8L * rl_n) {
                    f32x4 v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = ld4(src + (p + (long)k * rl_n) * cstride);
                    __builtin_amdgcn_sched_barrier(0);   // all 8 loads issued before the first add waits
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] += v[k];
                }
                for (; p < r1; p += rl_n) t[0] += ld4(src + p * cstride);
                s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
            } else {
                const float* src2 = (MODE == 2 || MODE == 3) ? in2 + coff + c : src;
                long p = r0 + rl;
                for (; p + 3L * rl_n < r1; p += 4L * rl_n) {
                    f32x4 v[4], z[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = ld4(src + (p + (long)k * rl_n) * cstride);
                        if (MODE == 2 || MODE == 3) z[k] = ld4(src2 + (p + (long)k * rl_n) * cstride);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (MODE == 1) {
                            const f32x4 d = v[k] - mv;
                            s += d * d;
                        } else if (MODE == 4) {
                            const f32x4 d = v[k] - mv;
                            s += d;
                            s2 += d * d;
                        } else {
                            if (MODE == 3) v[k] = masked(v[k], z[k]);
                            s += v[k];
                            s2 += v[k] * ((z[k] - mv) * rv);
                        }
                    }
                }
                for (; p < r1; p += rl_n) {
                    f32x4 v = ld4(src + p * cstride);
                    if (MODE == 1) {
                        const f32x4 d = v - mv;
                        s += d * d;
                    } else if (MODE == 4) {
                        const f32x4 d = v - mv;
                        s += d;
                        s2 += d * d;
                    } else {
                        const f32x4 z = ld4(src2 + p * cstride);
                        if (MODE == 3) v = masked(v, z);
                        s += v;
                        s2 += v * ((z - mv) * rv);
                    }
                }
            }
        }
        red[threadIdx.x] = s;
        if (MODE >= 2) red2[threadIdx.x] = s2;
        __syncthreads();
        if (rl == 0 && cq < c4n) {
            for (int k = 1; k < rl_n; ++k) {
                s += red[k * tpr + ct];
                if (MODE >= 2) s2 += red2[k * tpr + ct];
            }
            st4(partial + ((size_t)blockIdx.x * C + c), s);
            if (MODE >= 2) st4(partial + ((size_t)(gridDim.x + blockIdx.x) * C + c), s2);
        }
        __syncthreads();
    }
}

// final: out[c] (+)= scale * sum over chunks.  Block = 64 columns x 16 chunk slices (coalesced 256-byte
// rows, 16-way split of the chunk loop with 8 loads in flight per thread, LDS combine): deterministic.
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, int nchunks, int C,
                                                            float scale, float* __restrict__ out, int accumulate) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (c < C) {
        int k = sl;
        for (; k + 7 * 16 < nchunks; k += 8 * 16) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = partial[(size_t)(k + 16 * j) * C + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += v[j];
        }
        for (; k < nchunks; k += 16) s[0] += partial[(size_t)k * C + c];
    }
    red[sl][cl] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (sl == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][cl];
        t *= scale;
        out[c] = accumulate ? out[c] + t : t;
    }
}

static int colsum_chunks(long P, long& rows_per_chunk) {
    long n = P < CS_CHUNKS ? P : CS_CHUNKS;
    if (n < 1) n = 1;
    rows_per_chunk = (P + n - 1) / n;
    return (int)((P + rows_per_chunk - 1) / rows_per_chunk);
}

int launch_colsum(const float* in, long P, int cstride, int coff, int C, float* out, int accumulate, float* partial,
                  hipStream_t stream) {
    S3D_CHECK_ARG(C % 4 == 0 && cstride % 4 == 0 && coff % 4 == 0, "colsum: C/stride must be multiples of 4");
    if (P <= 0) return 0;
    long rpc;
    const int n = colsum_chunks(P, rpc);
    hipLaunchKernelGGL((colsum_kernel<0>), dim3(n), dim3(256), 0, stream, in, nullptr, P, cstride, coff, C, nullptr,
                       nullptr, partial, rpc, nullptr, nullptr);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, stream, partial, n, C, 1.f, out,
                       accumulate);
    S3D_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// BatchNorm2d train mode
// =============================================================================================
// one-pass statistics: m1 = E[x - k] (in `mean`), m2 = E[(x - k)^2] (in `rstd`), k = the pilot value of colsum mode 4
__global__ void bn_finalize_shift_kernel(const float* __restrict__ z, long P, int C, float* __restrict__ mean,
                                         float* __restrict__ rstd, float* __restrict__ running_mean,
                                         float* __restrict__ running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float k = 0.f;
    for (int j = 0; j < BN_PILOT_ROWS; ++j) k += z[bn_pilot_row(j, P) * C + c];
    k *= 1.f / BN_PILOT_ROWS;
    const float m1 = mean[c], m2 = rstd[c];
    const float mu = k + m1;
    const float var = fmaxf(m2 - m1 * m1, 0.f);
    mean[c] = mu;
    rstd[c] = 1.f / sqrtf(var + 1e-5f);
    if (running_mean) {
        running_mean[c] = 0.9f * running_mean[c] + 0.1f * mu;
        const float unb = P > 1 ? var * (float)P / (float)(P - 1) : var;
        running_var[c] = 0.9f * running_var[c] + 0.1f * unb;
    }
}

// cross-rank statistics, merged the way torch.nn.SyncBatchNorm merges them (no E[x^2] - mu^2 cancellation for channels
// with |mean| >> std): stage 1 all-reduces the per-rank means; stage 2 all-reduces var_r + (mu_r - mu)^2 (Chan's
// formula for ranks of equal element count P).  buf: [0,C) sum of means | [C,2C) stage-2 terms | [2C,3C) mu_r, var_r packed
__global__ void bn_sync_pack_kernel(const float* __restrict__ z, long P, int C, const float* __restrict__ m1v,
                                    const float* __restrict__ m2v, float* __restrict__ buf, float* __restrict__ keep) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float k = 0.f;
    for (int j = 0; j < BN_PILOT_ROWS; ++j) k += z[bn_pilot_row(j, P) * C + c];
    k *= 1.f / BN_PILOT_ROWS;
    const float m1 = m1v[c], m2 = m2v[c];
    const float mu = k + m1, var = fmaxf(m2 - m1 * m1, 0.f);
    buf[c] = mu;
    keep[c] = mu;        // this rank's mean and variance stay in the caller's mean / rstd slots until stage 2
    keep[C + c] = var;
}
__global__ void bn_sync_stage2_kernel(float* __restrict__ buf, const float* __restrict__ keep, int C, int world) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mu = buf[c] * (1.f / (float)world), d = keep[c] - mu;
    buf[C + c] = keep[C + c] + d * d;
}
// ... and, after the second all-reduce (sums over `world` ranks of equal element count P): the global statistics
__global__ void bn_sync_finalize_kernel(const float* __restrict__ buf, long P, int C, int world,
                                        float* __restrict__ mean, float* __restrict__ rstd,
                                        float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float inv = 1.f / (float)world;
    const float mu = buf[c] * inv;
    const float var = fmaxf(buf[C + c] * inv, 0.f);
    mean[c] = mu;
    rstd[c] = 1.f / sqrtf(var + 1e-5f);
    if (running_mean) {
        const float n = (float)P * (float)world;
        running_mean[c] = 0.9f * running_mean[c] + 0.1f * mu;
        running_var[c] = 0.9f * running_var[c] + 0.1f * (n > 1.f ? var * n / (n - 1.f) : var);
    }
}
__global__ void pack2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ buf, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        buf[c] = a[c];
        buf[C + c] = b[c];
    }
}
static int bn_sync_all_reduce(const BnSync* sync, int C, hipStream_t stream, int off = 0, int n = 0) {
    S3D_CHECK_ARG(sync->all_reduce_sum && sync->scratch && sync->world_size >= 1 && 4 * C <= 2048,
                  "sync_bn: bad descriptor (C = %d)", C);
    const int rc = sync->all_reduce_sum(sync->user, sync->scratch + off, n ? (long)n : 2L * C, (void*)stream);
    if (rc != 0) {
        s3d_set_error("sync_bn: the all-reduce callback returned %d", rc);
        return S3D_E_ARG;
    }
    return 0;
}

int launch_bn_stats(const float* z, long P, int C, float* mean, float* rstd, float* running_mean,
                    float* running_var, float* partial, hipStream_t stream, const BnSync* sync) {
    S3D_CHECK_ARG(C % 4 == 0 && P > 0, "bn_stats: bad dims");
    long rpc;
    const int n = colsum_chunks(P, rpc);
    if (sync) {
        hipLaunchKernelGGL((colsum_kernel<4>), dim3(n), dim3(256), 0, stream, z, nullptr, P, C, 0, C, nullptr,
                           nullptr, partial, rpc, nullptr, nullptr);
        S3D_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, stream, partial, n, C,
                           1.f / (float)P, mean, 0);
        hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, stream, partial + (size_t)n * C,
                           n, C, 1.f / (float)P, rstd, 0);
        hipLaunchKernelGGL(bn_sync_pack_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, z, P, C, mean, rstd,
                           sync->scratch, sync->scratch + 2 * C);
        S3D_LAUNCH_CHECK();
        TRY_RET(bn_sync_all_reduce(sync, C, stream, 0, C));          // stage 1: sum of the per-rank means
        hipLaunchKernelGGL(bn_sync_stage2_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sync->scratch,
                           sync->scratch + 2 * C, C, sync->world_size);
        S3D_LAUNCH_CHECK();
        TRY_RET(bn_sync_all_reduce(sync, C, stream, C, C));          // stage 2: sum of var_r + (mu_r - mu)^2
        hipLaunchKernelGGL(bn_sync_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sync->scratch, P, C,
                           sync->world_size, mean, rstd, running_mean, running_var);
        S3D_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL((colsum_kernel<4>), dim3(n), dim3(256), 0, stream, z, nullptr, P, C, 0, C, nullptr,
                       nullptr, partial, rpc, nullptr, nullptr);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, stream, partial, n, C,
                       1.f / (float)P, mean, 0);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, stream, partial + (size_t)n * C,
                       n, C, 1.f / (float)P, rstd, 0);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_shift_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, z, P, C, mean, rstd,
                       running_mean, running_var);
    S3D_LAUNCH_CHECK();
    return 0;
}

template <bool POOL>
__global__ void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ y, int n, int h, int w, int c) {
    const int c4 = c >> 2;
    const int ho = POOL ? h >> 1 : h, wo = POOL ? w >> 1 : w;
    const long total = (long)n * ho * wo * c4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % c4) * 4;
        const f32x4 mu = ld4(mean + cc), rs = ld4(rstd + cc), ga = ld4(gamma + cc), be = ld4(beta + cc);
        if (!POOL) {
            st4(y + idx * 4, bn_relu4(ld4(z + idx * 4), mu, rs, ga, be));
        } else {
            long r = idx / c4;
            const int x = (int)(r % wo);
            r /= wo;
            const int yy = (int)(r % ho);
            const int ni = (int)(r / ho);
            f32x4 best;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 v = bn_relu4(ld4(z + ((long)(ni * h + 2 * yy + (t >> 1)) * w + 2 * x + (t & 1)) * c + cc),
                                         mu, rs, ga, be);
#pragma unroll
                for (int i = 0; i < 4; ++i) best[i] = t == 0 ? v[i] : fmaxf(best[i], v[i]);
            }
            st4(y + idx * 4, best);
        }
    }
}

int launch_bn_apply(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    float* y, int n, int h, int w, int c, int pool, hipStream_t stream) {
    S3D_CHECK_ARG(c % 4 == 0 && (!pool || (h % 2 == 0 && w % 2 == 0)), "bn_apply: bad dims");
    const long total = (long)n * (pool ? h / 2 : h) * (pool ? w / 2 : w) * (c / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (pool)
        hipLaunchKernelGGL((bn_apply_kernel<true>), dim3(blocks), dim3(256), 0, stream, z, mean, rstd, gamma, beta, y,
                           n, h, w, c);
    else
        hipLaunchKernelGGL((bn_apply_kernel<false>), dim3(blocks), dim3(256), 0, stream, z, mean, rstd, gamma, beta, y,
                           n, h, w, c);
    S3D_LAUNCH_CHECK();
    return 0;
}

// BN -> ReLU -> 2x2 max-pool backward without a stored g: the pooled gradient belongs to the first maximum of its
// window (if that maximum is positive).  Both passes walk the POOLED grid, read the window's four z values and
// redo the forward's comparison.
struct PoolPick {
    f32x4 g[4];   // gradient routed to each of the four window positions
};
__device__ __forceinline__ PoolPick pool_route(const f32x4 (&zz)[4], const f32x4 d, const f32x4 mu, const f32x4 rs,
                                               const f32x4 ga, const f32x4 be) {
    f32x4 v[4], best;
    int arg[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        v[t] = bn_relu4(zz[t], mu, rs, ga, be);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (t == 0 || v[t][i] > best[i]) {
                best[i] = v[t][i];
                arg[i] = t;
            }
    }
    PoolPick o;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) o.g[t][i] = (arg[i] == t && best[i] > 0.f) ? d[i] : 0.f;
    return o;
}

// partial[chunk][c] = sum g, partial[nchunks + chunk][c] = sum g * xhat  over the chunk's pooled pixels
__global__ __launch_bounds__(256) void bn_bwd_pool_sums_kernel(const float* __restrict__ z,
                                                               const float* __restrict__ dy, int n, int h, int w,
                                                               int C, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               float* __restrict__ partial, long rows_per_chunk) {
    __shared__ f32x4 red[256], red2[256];
    const int c4n = C >> 2;
    const int tpr = c4n < 256 ? c4n : 256;
    const int rl_n = 256 / tpr;
    const int rl = threadIdx.x / tpr, ct = threadIdx.x - rl * tpr;
    const int ho = h >> 1, wo = w >> 1;
    const long Pp = (long)n * ho * wo;
    const long r0 = (long)blockIdx.x * rows_per_chunk;
    long r1 = r0 + rows_per_chunk;
    if (r1 > Pp) r1 = Pp;
    for (int cq0 = 0; cq0 < c4n; cq0 += tpr) {
        const int cq = cq0 + ct;
        const bool active = rl < rl_n && cq < c4n;
        const int c = (cq < c4n ? cq : 0) * 4;
        const f32x4 mu = ld4(mean + c), rs = ld4(rstd + c), ga = ld4(gamma + c), be = ld4(beta + c);
        f32x4 s = zero4(), s2 = zero4();
        if (active) {
            for (long p = r0 + rl; p < r1; p += rl_n) {
                const int x = (int)(p % wo);
                const long r = p / wo;
                const int yy = (int)(r % ho);
                const long ni = r / ho;
                const float* zb = z + ((ni * h + 2 * yy) * w + 2 * x) * C + c;
                f32x4 zz[4];
                zz[0] = ld4(zb);
                zz[1] = ld4(zb + C);
                zz[2] = ld4(zb + (long)w * C);
                zz[3] = ld4(zb + (long)w * C + C);
                const PoolPick pk = pool_route(zz, ld4(dy + p * C + c), mu, rs, ga, be);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    s += pk.g[t];
                    s2 += pk.g[t] * ((zz[t] - mu) * rs);
                }
            }
        }
        red[threadIdx.x] = s;
        red2[threadIdx.x] = s2;
        __syncthreads();
        if (rl == 0 && cq < c4n) {
            for (int k = 1; k < rl_n; ++k) {
                s += red[k * tpr + ct];
                s2 += red2[k * tpr + ct];
            }
            st4(partial + ((size_t)blockIdx.x * C + c), s);
            st4(partial + ((size_t)(gridDim.x + blockIdx.x) * C + c), s2);
        }
        __syncthreads();
    }
}

__global__ void bn_bwd_pool_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, const float* __restrict__ dy,
                                         const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                         float* __restrict__ dz, const float* __restrict__ add, int n, int h, int w,
                                         int c, float inv_count) {
    const int c4 = c >> 2, ho = h >> 1, wo = w >> 1;
    const long total = (long)n * ho * wo * c4;
    const float invP = inv_count > 0.f ? inv_count : 1.f / (float)((long)n * h * w);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % c4) * 4;
        long r = idx / c4;
        const int x = (int)(r % wo);
        r /= wo;
        const int yy = (int)(r % ho);
        const long ni = r / ho;
        const f32x4 mu = ld4(mean + cc), rs = ld4(rstd + cc), ga = ld4(gamma + cc), be = ld4(beta + cc);
        const f32x4 sb = ld4(dbeta + cc) * invP, sg = ld4(dgamma + cc) * invP;
        const long base = ((ni * h + 2 * yy) * w + 2 * x) * c + cc;
        const long off[4] = {0, c, (long)w * c, (long)w * c + c};
        f32x4 zz[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) zz[t] = ld4(z + base + off[t]);
        const PoolPick pk = pool_route(zz, ld4(dy + idx * 4), mu, rs, ga, be);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 o = ga * rs * (pk.g[t] - sb - (zz[t] - mu) * rs * sg);
            if (add) o += ld4(add + base + off[t]);
            st4(dz + base + off[t], o);
        }
    }
}

// MASK: g is recomputed as relu'(bn(z)) * dy (dy may alias g_dz); otherwise g_dz holds g on entry
template <bool MASK>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* dy,
                                    const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                    float* g_dz, const float* __restrict__ add, long P, int c, float inv_count = 0.f) {
    const int c4 = c >> 2;
    const long total = P * c4;
    const float invP = inv_count > 0.f ? inv_count : 1.f / (float)P;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % c4) * 4;
        const f32x4 mu = ld4(mean + cc), rs = ld4(rstd + cc), ga = ld4(gamma + cc);
        const f32x4 sb = ld4(dbeta + cc), sg = ld4(dgamma + cc);
        const f32x4 zz = ld4(z + idx * 4);
        const f32x4 xh = (zz - mu) * rs;
        f32x4 g;
        if (MASK) {
            g = ld4(dy + idx * 4);
            const f32x4 y = bn_relu4(zz, mu, rs, ga, ld4(beta + cc));
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = y[i] > 0.f ? g[i] : 0.f;
        } else {
            g = ld4(g_dz + idx * 4);
        }
        f32x4 o = ga * rs * (g - sb * invP - xh * (sg * invP));
        if (add) o += ld4(add + idx * 4);
        st4(g_dz + idx * 4, o);
    }
}

int launch_bn_bwd(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                  const float* dy, float* dz, float* dgamma, float* dbeta, int n, int h, int w, int c, int pool,
                  float* partial, hipStream_t stream, const float* add, const BnSync* sync) {
    S3D_CHECK_ARG(c % 4 == 0, "bn_bwd: C %% 4");
    S3D_CHECK_ARG(!pool || (h % 2 == 0 && w % 2 == 0), "bn_bwd: pooled map %d x %d must be even", h, w);
    const long P = (long)n * h * w;
    long rpc;
    // cross-rank statistics: dz needs the sums over ALL ranks (and their element count); the parameter gradients
    // dgamma / dbeta stay this rank's sums (the gradient all-reduce averages them like every other gradient)
    const float* sum_b = dbeta;
    const float* sum_g = dgamma;
    float inv_count = 0.f;
    auto sync_sums = [&]() -> int {
        if (!sync) return 0;
        hipLaunchKernelGGL(pack2_kernel, dim3((c + 255) / 256), dim3(256), 0, stream, dbeta, dgamma, sync->scratch, c);
        S3D_LAUNCH_CHECK();
        TRY_RET(bn_sync_all_reduce(sync, c, stream));
        sum_b = sync->scratch;
        sum_g = sync->scratch + c;
        inv_count = 1.f / ((float)P * (float)sync->world_size);
        return 0;
    };
    if (pool) {   // both passes walk the pooled grid and redo the window comparison
        const long Pp = (long)n * (h / 2) * (w / 2);
        const int nch = colsum_chunks(Pp, rpc);
        hipLaunchKernelGGL(bn_bwd_pool_sums_kernel, dim3(nch), dim3(256), 0, stream, z, dy, n, h, w, c, mean, rstd,
                           gamma, beta, partial, rpc);
        S3D_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_final_kernel, dim3((c + 63) / 64), dim3(1024), 0, stream, partial, nch, c, 1.f,
                           dbeta, 0);
        S3D_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_final_kernel, dim3((c + 63) / 64), dim3(1024), 0, stream,
                           partial + (size_t)nch * c, nch, c, 1.f, dgamma, 0);
        S3D_LAUNCH_CHECK();
        const long tot = Pp * (c / 4);
        const int ba = (int)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192);
        TRY_RET(sync_sums());
        hipLaunchKernelGGL(bn_bwd_pool_apply_kernel, dim3(ba), dim3(256), 0, stream, z, mean, rstd, gamma, beta, dy,
                           sum_b, sum_g, dz, add, n, h, w, c, inv_count);
        S3D_LAUNCH_CHECK();
        return 0;
    }
    // g = relu mask * dy is recomputed in both passes instead of stored
    const int nch = colsum_chunks(P, rpc);
    hipLaunchKernelGGL((colsum_kernel<3>), dim3(nch), dim3(256), 0, stream, dy, z, P, c, 0, c, mean, rstd, partial,
                       rpc, gamma, beta);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((c + 63) / 64), dim3(1024), 0, stream, partial, nch, c, 1.f, dbeta,
                       0);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((c + 63) / 64), dim3(1024), 0, stream, partial + (size_t)nch * c,
                       nch, c, 1.f, dgamma, 0);
    S3D_LAUNCH_CHECK();
    const long tot = P * (c / 4);
    const int ba = (int)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192);
    TRY_RET(sync_sums());
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true>), dim3(ba), dim3(256), 0, stream, z, mean, rstd, gamma, beta, dy,
                       sum_b, sum_g, dz, add, P, c, inv_count);
    S3D_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// first encoder conv: 3x3, C_in <= 4 straight from the NCHW image, 64 output channels, exact fp32 FMA.  The layer
// is a 0.1 GFLOP/image trickle behind an 805 MB output write: the implicit-GEMM tile (K padded to 16 channels)
// spent 0.84 ms on it, this kernel is bound by the write.  Thread = 4 pixels of a row x 4 output channels.
// =============================================================================================
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_first_kernel(const float* __restrict__ img,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int n, int H, int W, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu) {
    __shared__ f32x4 s_w[9 * CIN][16];   // [tap * CIN + ci][channel quad]
    for (int idx = threadIdx.x; idx < 9 * CIN * 64; idx += 256) {
        const int co = idx & 63, k = idx >> 6, tap = k / CIN, ci = k - tap * CIN;
        ((float*)s_w)[k * 64 + co] = w[(co * CIN + ci) * 9 + tap];
    }
    __syncthreads();
    const int cq = threadIdx.x & 15, wq = W >> 2;
    const f32x4 b4 = bias ? ld4(bias + cq * 4) : zero4();
    const long groups = (long)n * H * wq;
    for (long grp = (long)blockIdx.x * 16 + (threadIdx.x >> 4); grp < groups; grp += (long)gridDim.x * 16) {
        const int x0 = (int)(grp % wq) * 4;
        const long r = grp / wq;
        const int y = (int)(r % H);
        const long ni = r / H;
        f32x4 acc[4] = {b4, b4, b4, b4};
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = y + dy - 1;
                float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (yy >= 0 && yy < H) {
                    const float* row = img + ((ni * CIN + ci) * H + yy) * W;
                    const f32x4 mid = ld4(row + x0);
                    v[0] = x0 > 0 ? row[x0 - 1] : 0.f;
                    v[1] = mid[0]; v[2] = mid[1]; v[3] = mid[2]; v[4] = mid[3];
                    v[5] = x0 + 4 < W ? row[x0 + 4] : 0.f;
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x4 wv = s_w[(dy * 3 + dx) * CIN + ci][cq];
#pragma unroll
                    for (int px = 0; px < 4; ++px) acc[px] += wv * v[px + dx];
                }
            }
        }
        float* o = out + ((ni * H + y) * W + x0) * 64 + cq * 4;
        if (scale) {   // inference epilogue: folded BatchNorm (+ ReLU)
            const f32x4 sc = ld4(scale + cq * 4), sh = ld4(shift + cq * 4);
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                acc[px] = acc[px] * sc + sh;
                if (relu)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[px][i] = fmaxf(acc[px][i], 0.f);
            }
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) st4(o + px * 64, acc[px]);
    }
}

int launch_conv3x3_first(const float* img_nchw, int cin, const float* w_oihw, const float* bias, float* out_nhwc,
                         int n, int h, int w, hipStream_t stream, const float* scale, const float* shift, int relu) {
    S3D_CHECK_ARG((cin == 1 || cin == 3) && w % 4 == 0, "conv3x3_first: C_in %d, W %d", cin, w);
    const long groups = (long)n * h * (w / 4);
    const int blocks = (int)((groups + 15) / 16 < 16384 ? (groups + 15) / 16 : 16384);
    if (cin == 1)
        hipLaunchKernelGGL((conv3x3_first_kernel<1>), dim3(blocks), dim3(256), 0, stream, img_nchw, w_oihw, bias,
                           out_nhwc, n, h, w, scale, shift, relu);
    else
        hipLaunchKernelGGL((conv3x3_first_kernel<3>), dim3(blocks), dim3(256), 0, stream, img_nchw, w_oihw, bias,
                           out_nhwc, n, h, w, scale, shift, relu);
    S3D_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// elementwise
// =============================================================================================
static inline int ew_blocks(long n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] += alpha * x[i];
}
int launch_axpy(float* y, const float* x, float alpha, long n, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, y, x, alpha, n);
    S3D_LAUNCH_CHECK();
    return 0;
}

__global__ void slice_sum_kernel(const float* __restrict__ in, float* __restrict__ out, int batch, int ns,
                                 long per_img, int accumulate) {
    const long total = (long)batch * per_img;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / per_img, r = i - b * per_img;
        float s = 0.f;
        for (int k = 0; k < ns; ++k) s += in[(b * ns + k) * per_img + r];
        out[i] = accumulate ? out[i] + s : s;
    }
}
int launch_slice_sum(const float* in, float* out, int batch, int ns, long per_img, int accumulate,
                     hipStream_t stream) {
    hipLaunchKernelGGL(slice_sum_kernel, dim3(ew_blocks((long)batch * per_img)), dim3(256), 0, stream, in, out, batch,
                       ns, per_img, accumulate);
    S3D_LAUNCH_CHECK();
    return 0;
}

__global__ void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dz,
                                int n, int c, int h, int w, int cpad) {
    const long total = (long)n * h * w * cpad;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % cpad);
        long r = idx / cpad;
        const int x = (int)(r % w);
        r /= w;
        const int yy = (int)(r % h);
        const int ni = (int)(r / h);
        float v = 0.f;
        if (cc < c) {
            const long s = ((long)(ni * c + cc) * h + yy) * w + x;
            v = dy[s] * (1.f - y[s] * y[s]);
        }
        dz[idx] = v;
    }
}
int launch_tanh_bwd(const float* y, const float* dy, float* dz, int n, int c, int h, int w, int cpad,
                    hipStream_t stream) {
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(ew_blocks((long)n * h * w * cpad)), dim3(256), 0, stream, y, dy, dz, n, c,
                       h, w, cpad);
    S3D_LAUNCH_CHECK();
    return 0;
}

#define L1B S3D_L1_PARTIAL_FLOATS
// sum |a - b| (per-block partials) and, optionally, its gradient scale * sign(a - b) written to / added into grad.
// relu_mask: grad is the gradient w.r.t. a ReLU output `a` and what leaves is the gradient w.r.t. the ReLU's input —
// the finished element (the accumulated value included) is zeroed where a <= 0 (the VGG loss taps: saves the separate
// mask pass over the tensor).  Four elements per thread and step when n and the pointers allow 16-byte accesses.
template <bool VEC>
__global__ __launch_bounds__(256) void l1_fb_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                    float scale, float* __restrict__ grad, int acc_grad, int relu_mask,
                                                    float* __restrict__ partial) {   // scale: of the gradient
    __shared__ float red[4];
    float s = 0.f;
    constexpr int V = VEC ? 4 : 1;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * V; i < n; i += (long)gridDim.x * 256 * V) {
        float av[4], bv[4], gv[4] = {0.f, 0.f, 0.f, 0.f};
        if (VEC) {
            const f32x4 a4 = ld4(a + i), b4 = ld4(b + i);
            f32x4 g4 = zero4();
            if (grad && acc_grad) g4 = ld4(grad + i);
#pragma unroll
            for (int t = 0; t < 4; ++t) { av[t] = a4[t]; bv[t] = b4[t]; gv[t] = g4[t]; }
        } else {
            av[0] = a[i]; bv[0] = b[i];
            if (grad && acc_grad) gv[0] = grad[i];
        }
#pragma unroll
        for (int t = 0; t < V; ++t) {
            const float d = av[t] - bv[t];
            s += fabsf(d);
            const float gsign = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
            gv[t] += gsign;
            if (relu_mask && !(av[t] > 0.f)) gv[t] = 0.f;
        }
        if (grad) {
            if (VEC) st4(grad + i, f32x4{gv[0], gv[1], gv[2], gv[3]});
            else grad[i] = gv[0];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void scalar_final_kernel(const float* __restrict__ partial, int n, float scale,
                                                           float* __restrict__ acc) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) acc[0] += scale * ((red[0] + red[1]) + (red[2] + red[3]));
}
int launch_l1_fwd_bwd(const float* a, const float* b, long n, float scale, float* grad, int accumulate_grad,
                      float* partial, float* loss_acc, hipStream_t stream, float grad_mul, int relu_mask) {
    if (n <= 0) return 0;
    const bool vec = n % 4 == 0 && (((size_t)a | (size_t)b | (size_t)grad) & 15) == 0;
    const long work = vec ? n / 4 : n;
    const int blocks = (int)((work + 255) / 256 < L1B ? (work + 255) / 256 : L1B);
    if (vec)
        hipLaunchKernelGGL(l1_fb_kernel<true>, dim3(blocks), dim3(256), 0, stream, a, b, n, scale * grad_mul, grad,
                           accumulate_grad, relu_mask, partial);
    else
        hipLaunchKernelGGL(l1_fb_kernel<false>, dim3(blocks), dim3(256), 0, stream, a, b, n, scale * grad_mul, grad,
                           accumulate_grad, relu_mask, partial);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(scalar_final_kernel, dim3(1), dim3(256), 0, stream, partial, blocks, scale, loss_acc);
    S3D_LAUNCH_CHECK();
    return 0;
}

// in-place x *= s on a table of tensors (blockIdx.y = table entry)
__global__ __launch_bounds__(256) void scale_table_kernel(const ScaleTable t, int first, float s) {
    float* p = t.p[first + blockIdx.y];
    const long n = t.n[first + blockIdx.y];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] *= s;
}
int launch_scale_table(const ScaleTable& t, float s, hipStream_t stream) {
    if (t.count <= 0 || s == 1.f) return 0;
    hipLaunchKernelGGL(scale_table_kernel, dim3(64, (unsigned)t.count), dim3(256), 0, stream, t, 0, s);
    S3D_LAUNCH_CHECK();
    return 0;
}

__global__ void relu_mask_bwd_kernel(const float* __restrict__ y, float* __restrict__ dy, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        if (!(y[i] > 0.f)) dy[i] = 0.f;
}
int launch_relu_mask_bwd(const float* y, float* dy, long n, hipStream_t stream) {
    hipLaunchKernelGGL(relu_mask_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, y, dy, n);
    S3D_LAUNCH_CHECK();
    return 0;
}

// max-pool 2x2 backward: y (N,H,W,C) pre-pool, dyp (N,H/2,W/2,C) -> dy (N,H,W,C), first maximum wins; relu_mask: y is a ReLU
// output and dy leaves as the gradient w.r.t. the ReLU's input (zero where y <= 0)
__global__ void pool_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dyp, float* __restrict__ dy,
                                int n, int h, int w, int c, int relu_mask) {
    const int c4 = c >> 2, ho = h >> 1, wo = w >> 1;
    const long total = (long)n * ho * wo * c4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % c4) * 4;
        long r = idx / c4;
        const int x = (int)(r % wo);
        r /= wo;
        const int yy = (int)(r % ho);
        const int ni = (int)(r / ho);
        const f32x4 d = ld4(dyp + idx * 4);
        f32x4 v[4], best;
        int arg[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            v[t] = ld4(y + ((long)(ni * h + 2 * yy + (t >> 1)) * w + 2 * x + (t & 1)) * c + cc);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (t == 0 || v[t][i] > best[i]) {
                    best[i] = v[t][i];
                    arg[i] = t;
                }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (arg[i] == t && !(relu_mask && !(v[t][i] > 0.f))) ? d[i] : 0.f;
            st4(dy + ((long)(ni * h + 2 * yy + (t >> 1)) * w + 2 * x + (t & 1)) * c + cc, o);
        }
    }
}
int launch_pool_bwd(const float* y, const float* dyp, float* dy, int n, int h, int w, int c, hipStream_t stream,
                    int relu_mask) {
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(ew_blocks((long)n * (h / 2) * (w / 2) * (c / 4))), dim3(256), 0, stream, y,
                       dyp, dy, n, h, w, c, relu_mask);
    S3D_LAUNCH_CHECK();
    return 0;
}

// torch.optim.Adam (no weight decay / amsgrad); bc1 = 1-b1^t, bc2 = 1-b2^t
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2) {
    const float step = lr / bc1, rbc2 = 1.f / sqrtf(bc2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step * mi / (sqrtf(vi) * rbc2 + eps);
    }
}
int launch_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                float bc1, float bc2, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2);
    S3D_LAUNCH_CHECK();
    return 0;
}

// the same update for a table of tensors whose gradient / moment buffers are slices of three flat arrays
__global__ __launch_bounds__(256) void adam_table_kernel(const AdamTable t, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, float lr, float b1,
                                                         float b2, float eps, float bc1, float bc2) {
    float* __restrict__ p = t.p[blockIdx.y];
    const long off = t.off[blockIdx.y], n = t.n[blockIdx.y];
    const float step = lr / bc1, rbc2 = 1.f / sqrtf(bc2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[off + i];
        const float mi = b1 * m[off + i] + (1.f - b1) * gi;
        const float vi = b2 * v[off + i] + (1.f - b2) * gi * gi;
        m[off + i] = mi;
        v[off + i] = vi;
        p[i] -= step * mi / (sqrtf(vi) * rbc2 + eps);
    }
}
int launch_adam_table(const AdamTable& t, const float* g, float* m, float* v, float lr, float b1, float b2, float eps,
                      float bc1, float bc2, hipStream_t stream) {
    if (t.count <= 0) return 0;
    hipLaunchKernelGGL(adam_table_kernel, dim3(64, (unsigned)t.count), dim3(256), 0, stream, t, g, m, v, lr, b1, b2, eps,
                       bc1, bc2);
    S3D_LAUNCH_CHECK();
    return 0;
}

// a <- dropout(relu(a)), dh <- dh * mask * (a > 0); element i of the block is hidden unit (row0*2048 + i)
__global__ void relu_bwd_inplace_kernel(float* __restrict__ a, float* __restrict__ dh, long n, long row0,
                                        const DropCfg drop) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = a[i];
        if (v > 0.f) {
            if (drop.p > 0.f) {
                const float mk = s3d_drop(drop, (unsigned long long)row0 * S3D_FFN + i);
                a[i] = v * mk;
                dh[i] *= mk;
            }
        } else {
            a[i] = 0.f;
            dh[i] = 0.f;
        }
    }
}
int launch_relu_bwd_inplace(float* a, float* dh, long n, long row0, const DropCfg& drop, hipStream_t stream) {
    hipLaunchKernelGGL(relu_bwd_inplace_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, a, dh, n, row0, drop);
    S3D_LAUNCH_CHECK();
    return 0;
}

__global__ void dropout_apply_kernel(const float* __restrict__ in, float* __restrict__ out, long n,
                                     unsigned long long idx0, const DropCfg drop) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (in ? in[i] : 1.f) * s3d_drop(drop, idx0 + i);
}
int launch_dropout_apply(const float* in, float* out, long n, unsigned long long idx0, const DropCfg& drop,
                         hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, in, out, n, idx0, drop);
    S3D_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// LayerNorm over rows of 128 (32 lanes x float4 per row; a wave handles 2 rows per pass)
// =============================================================================================
__device__ __forceinline__ float half_sum(float v) {  // sum over the 32 lanes sharing lane>>5
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ u, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     long rows) {
    const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;
    const f32x4 ga = ld4(gamma + 4 * l), be = ld4(beta + 4 * l);
    for (long row = (long)blockIdx.x * 8 + sub; row < rows; row += (long)gridDim.x * 8) {
        const f32x4 v = ld4(u + row * 128 + 4 * l);
        const float mean = half_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 128.f);
        const f32x4 d = v - mean;
        const float var = half_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 128.f);
        const float rstd = 1.f / sqrtf(var + 1e-5f);
        st4(y + row * 128 + 4 * l, d * rstd * ga + be);
    }
}
int launch_ln_fwd(const float* u, const float* gamma, const float* beta, float* y, long rows, hipStream_t stream) {
    if (rows <= 0) return 0;
    const long nb = (rows + 7) / 8 < 4096 ? (rows + 7) / 8 : 4096;
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, stream, u, gamma, beta, y, rows);
    S3D_LAUNCH_CHECK();
    return 0;
}

#define LN_BLOCKS 1024
// DUM: also writes dum = du * dropout mask of (row, channel) (the gradient w.r.t. the pre-dropout branch output that
// the layer's GEMMs consume; dum == du without dropout, then only its column sums are new) and the column sums of dum
// (the bias gradient of the linear layer in front of the dropout) — one pass instead of ln_bwd + dropout_apply + colsum
template <bool DUM>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ u, const float* __restrict__ gamma,
                                                     const float* __restrict__ dy, float* __restrict__ du, long rows,
                                                     float* __restrict__ partial, const DropCfg drop,
                                                     float* __restrict__ dum) {
    constexpr int NP = DUM ? 3 : 2;
    __shared__ __attribute__((aligned(16))) float s_p[NP][8][128];
    const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;
    const f32x4 ga = ld4(gamma + 4 * l);
    f32x4 dg = zero4(), db = zero4(), dm = zero4();
    // (Round 5: four rows in flight per 32-lane group measured SLOWER, 1.44 against 1.27 ms average per call — the kernel runs at
    // 5.6 TB/s over its four row streams, it is bandwidth-bound, not latency-bound; profiles/r05_train_experiments.md.)
    for (long row = (long)blockIdx.x * 8 + sub; row < rows; row += (long)gridDim.x * 8) {
        const f32x4 v = ld4(u + row * 128 + 4 * l);
        const f32x4 g = ld4(dy + row * 128 + 4 * l);
        const float mean = half_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 128.f);
        const f32x4 d = v - mean;
        const float var = half_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 128.f);
        const float rstd = 1.f / sqrtf(var + 1e-5f);
        const f32x4 xh = d * rstd;
        const f32x4 gx = g * ga;
        const float m1 = half_sum((gx[0] + gx[1]) + (gx[2] + gx[3])) * (1.f / 128.f);
        const float m2 = half_sum(gx[0] * xh[0] + gx[1] * xh[1] + gx[2] * xh[2] + gx[3] * xh[3]) * (1.f / 128.f);
        const f32x4 r = (gx - m1 - xh * m2) * rstd;
        st4(du + row * 128 + 4 * l, r);
        dg += g * xh;
        db += g;
        if (DUM) {
            f32x4 rm = r;
            if (drop.p > 0.f) {
                float mk[4];
                s3d_drop4(drop, (unsigned long long)row * 128 + 4 * l, mk);
                rm = f32x4{r[0] * mk[0], r[1] * mk[1], r[2] * mk[2], r[3] * mk[3]};
            }
            if (dum && dum != du) st4(dum + row * 128 + 4 * l, rm);   // NULL: column sums only (the consumer regenerates the masks)
            dm += rm;
        }
    }
    st4(&s_p[0][sub][4 * l], dg);
    st4(&s_p[1][sub][4 * l], db);
    if (DUM) st4(&s_p[NP - 1][sub][4 * l], dm);
    __syncthreads();
    if (threadIdx.x < 128) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += s_p[q][k][threadIdx.x];
            partial[(size_t)blockIdx.x * (128 * NP) + 128 * q + threadIdx.x] = acc;
        }
    }
}
// dum / dsum: optional third output (see the kernel): dum may equal du when drop.p == 0, or be NULL — then only its column
// sums are formed and the consumer rebuilds dum = du * mask itself from the counter-based masks (the split-precision FFN
// data pass does: one 2.66 GB store and one load less per 5.2 M-row layer); dsum[128] = column sums of dum
int launch_ln_bwd(const float* u, const float* gamma, const float* dy, float* du, long rows, float* dgamma,
                  float* dbeta, int accumulate, float* partial, hipStream_t stream, const DropCfg* drop, float* dum,
                  float* dsum) {
    if (rows <= 0) return 0;
    const int nb = (int)((rows + 7) / 8 < LN_BLOCKS ? (rows + 7) / 8 : LN_BLOCKS);
    const bool third = dsum != nullptr;
    S3D_CHECK_ARG(!third || (drop && (drop->p <= 0.f || dum != du)), "ln_bwd: dropped output needs its own buffer");
    const int np = third ? 3 : 2;
    if (third)
        hipLaunchKernelGGL(ln_bwd_kernel<true>, dim3(nb), dim3(256), 0, stream, u, gamma, dy, du, rows, partial, *drop, dum);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<false>, dim3(nb), dim3(256), 0, stream, u, gamma, dy, du, rows, partial,
                           make_drop(0, 0.f, 0), nullptr);
    S3D_LAUNCH_CHECK();
    // partial rows are [block][128 np] = dgamma(128) | dbeta(128) [| dsum(128)]
    hipLaunchKernelGGL(colsum_final_kernel, dim3((128 * np + 63) / 64), dim3(1024), 0, stream, partial, nb, 128 * np, 1.f,
                       partial + (size_t)nb * 128 * np, 0);
    S3D_LAUNCH_CHECK();
    float* fin = partial + (size_t)nb * 128 * np;
    if (accumulate) {
        TRY_RET(launch_axpy(dgamma, fin, 1.f, 128, stream));
        TRY_RET(launch_axpy(dbeta, fin + 128, 1.f, 128, stream));
    } else {
        if (hipMemcpyAsync(dgamma, fin, 128 * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess ||
            hipMemcpyAsync(dbeta, fin + 128, 128 * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) {
            s3d_set_error("ln_bwd: memcpy failed");
            return (int)hipErrorUnknown;
        }
    }
    if (third && hipMemcpyAsync(dsum, fin + 256, 128 * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) {
        s3d_set_error("ln_bwd: memcpy failed");
        return (int)hipErrorUnknown;
    }
    return 0;
}

// =============================================================================================
// attention core on stored QKV: [groups][T][16 queries][384 = q|k|v, head h at cols 32h..32h+31]
// One workgroup per (group, head); thread = (query ql = tid & 15, token tt = tid >> 4).  The head's K and V
// rows (T*16 rows of 32 floats) are staged once in LDS (36-float row stride: 16 query lanes hit 16
// distinct bank quads, the token lanes of a wave broadcast), so every global byte is read once.
// =============================================================================================
#define ATT_SCALE 0.17677669529663687f
#define ATT_LD 36
#define ATT_THREADS 256

// the same copy in two halves — global -> registers (issued a phase early, so that its latency runs under the compute of
// the phase in between), registers -> LDS: T*16*8 16-byte pieces over 256 threads = up to 7 per thread
#define ATT_PF 7
__device__ __forceinline__ void att_fetch(const float* __restrict__ src, int src_ld, int coff, int T, f32x4 (&r)[ATT_PF],
                                          int tid = threadIdx.x) {
#pragma unroll
    for (int k = 0; k < ATT_PF; ++k) {
        const int i = tid + ATT_THREADS * k;
        const int ic = i < T * 128 ? i : T * 128 - 1;
        r[k] = ld4(src + (long)(ic >> 3) * src_ld + coff + 4 * (ic & 7));
    }
}
__device__ __forceinline__ void att_put(const f32x4 (&r)[ATT_PF], int T, float* dst, int tid = threadIdx.x) {
#pragma unroll
    for (int k = 0; k < ATT_PF; ++k) {
        const int i = tid + ATT_THREADS * k;
        if (i < T * 128) st4(dst + (i >> 3) * ATT_LD + 4 * (i & 7), r[k]);
    }
}
// rows [T*16][32] of one head (column offset coff in the 384/128-wide source) -> LDS [row][ATT_LD]
__device__ __forceinline__ void att_stage(const float* __restrict__ src, int src_ld, int coff, int T, float* dst,
                                          int tid = threadIdx.x) {
    for (int i = tid; i < T * 16 * 8; i += ATT_THREADS) {
        const int row = i >> 3, q4 = i & 7;
        st4(dst + row * ATT_LD + 4 * q4, ld4(src + (long)row * src_ld + coff + 4 * q4));
    }
}

__global__ __launch_bounds__(ATT_THREADS, 2) void attn_core_fwd_kernel(const float* __restrict__ qkv,
                                                                    float* __restrict__ o, long groups, int T,
                                                                    const DropCfg drop) {
    __shared__ __attribute__((aligned(16))) float sK[S3D_N_TOKENS_MAX * 16 * ATT_LD], sV[S3D_N_TOKENS_MAX * 16 * ATT_LD];
    const int tid = threadIdx.x;
    const int ql = tid & 15, tq = tid >> 4;
    // the next task's K / V pieces are requested before this task's compute and parked in LDS after it (the kernel moves
    // 10.7 GB per call and used to expose every load: stage -> barrier -> compute -> barrier)
    f32x4 pk[ATT_PF], pv[ATT_PF];
    if ((long)blockIdx.x < groups * 4) {
        const float* b0 = qkv + ((long)blockIdx.x >> 2) * T * 16 * 384;
        att_fetch(b0, 384, 128 + 32 * (int)(blockIdx.x & 3), T, pk);
        att_fetch(b0, 384, 256 + 32 * (int)(blockIdx.x & 3), T, pv);
    }
    for (long task = blockIdx.x; task < groups * 4; task += gridDim.x) {
        const long grp = task >> 2;
        const int h = (int)(task & 3);
        const float* base = qkv + grp * T * 16 * 384;
        __syncthreads();
        att_put(pk, T, sK);
        att_put(pv, T, sV);
        const bool on = tq < T;
        const float* qrow = base + ((on ? tq : 0) * 16 + ql) * 384 + 32 * h;
        f32x4 qv[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) qv[d] = ld4(qrow + 4 * d);
        {
            const long nt = task + gridDim.x;
            if (nt < groups * 4) {
                const float* nb = qkv + (nt >> 2) * T * 16 * 384;
                att_fetch(nb, 384, 128 + 32 * (int)(nt & 3), T, pk);
                att_fetch(nb, 384, 256 + 32 * (int)(nt & 3), T, pv);
            }
        }
        __syncthreads();
        if (!on) continue;
        float sc[S3D_N_TOKENS_MAX];
        float mx = -1e30f;
#pragma unroll
        for (int tk = 0; tk < S3D_N_TOKENS_MAX; ++tk)
            if (tk < T) {
                const float* krow = sK + (tk * 16 + ql) * ATT_LD;
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const f32x4 kv = ld4(krow + 4 * d);
                    s += qv[d][0] * kv[0] + qv[d][1] * kv[1] + qv[d][2] * kv[2] + qv[d][3] * kv[3];
                }
                sc[tk] = s * ATT_SCALE;
                mx = fmaxf(mx, sc[tk]);
            }
        float den = 0.f;
#pragma unroll
        for (int tk = 0; tk < S3D_N_TOKENS_MAX; ++tk)
            if (tk < T) {
                sc[tk] = expf(sc[tk] - mx);
                den += sc[tk];
            }
        const float inv = 1.f / den;
        f32x4 ov[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) ov[d] = zero4();
#pragma unroll
        for (int tk = 0; tk < S3D_N_TOKENS_MAX; ++tk)
            if (tk < T) {
                const float* vrow = sV + (tk * 16 + ql) * ATT_LD;
                float p = sc[tk] * inv;
                if (drop.p > 0.f)   // index = ((row_q * 4 + head) * 16 + key)
                    p *= s3d_drop(drop, ((unsigned long long)((grp * T + tq) * 16 + ql) * 4 + h) * 16 + tk);
#pragma unroll
                for (int d = 0; d < 8; ++d) ov[d] += ld4(vrow + 4 * d) * p;
            }
        float* orow = o + (grp * T * 16 + tq * 16 + ql) * 128 + 32 * h;
#pragma unroll
        for (int d = 0; d < 8; ++d) st4(orow + 4 * d, ov[d]);
    }
}

int launch_attn_core_fwd(const float* qkv, float* o, long groups, int T, const DropCfg& drop, hipStream_t stream) {
    if (groups <= 0) return 0;
    S3D_CHECK_ARG(T >= 1 && T <= S3D_N_TOKENS_MAX, "attn core: T %d", T);
    const long nb = groups * 4 < 16384 ? groups * 4 : 16384;
    hipLaunchKernelGGL(attn_core_fwd_kernel, dim3((unsigned)nb), dim3(ATT_THREADS), 0, stream, qkv, o, groups, T, drop);
    S3D_LAUNCH_CHECK();
    return 0;
}

// backward: phase A (thread = (ql, tq), K/V of the head in LDS): P row, dS row -> LDS, dQ row;
//           phase B (thread = (ql, tk), Q/dO of the head in the same LDS):
//           dK[tk] = sum_tq dS[tq][tk] Q[tq] * scale,  dV[tk] = sum_tq P[tq][tk] dO[tq]
// Every global access is row-coalesced (8 lanes x 16 B = one head row of 128 B; att_fetch / att_put / att_flush): a thread
// that walks its own row — Q / dO in, dQ / dK / dV out, 13 of the kernel's 19 GB — touches 64 different lines with every
// instruction, which held the kernel at 2.6 TB/s.  Rows now change hands in LDS: Q / dO are parked first and each thread
// picks its row up from there, gradient rows are parked by their owners and flushed by all threads (the K / V tiles are
// dead by then).  Ten barriers per task instead of four; Q / dO of the next task are requested under phase B, K / V of
// this one while the Q / dO rows are being picked up.
__device__ __forceinline__ void att_flush(const float* src, int T, float* __restrict__ dst, int dst_ld, int coff, int tid) {
    for (int i = tid; i < T * 16 * 8; i += ATT_THREADS) {
        const int row = i >> 3, q4 = i & 7;
        st4(dst + (long)row * dst_ld + coff + 4 * q4, ld4(src + row * ATT_LD + 4 * q4));
    }
}
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_core_bwd_kernel(const float* __restrict__ qkv,
                                                                    const float* __restrict__ d_o,
                                                                    float* __restrict__ dqkv, long groups, int T,
                                                                    const DropCfg drop) {
    __shared__ __attribute__((aligned(16))) float sA[S3D_N_TOKENS_MAX * 16 * ATT_LD], sB[S3D_N_TOKENS_MAX * 16 * ATT_LD];
    __shared__ float sP[16 * S3D_N_TOKENS_MAX * S3D_N_TOKENS_MAX], sS[16 * S3D_N_TOKENS_MAX * S3D_N_TOKENS_MAX];
    f32x4 pa[ATT_PF], pb[ATT_PF];   // coalesced pieces in flight: Q / dO of the next task, K / V of this one
    if ((long)blockIdx.x < groups * 4) {
        const long g0 = (long)blockIdx.x >> 2;
        const int h0 = (int)(blockIdx.x & 3);
        att_fetch(qkv + g0 * T * 16 * 384, 384, 32 * h0, T, pa);
        att_fetch(d_o + g0 * T * 16 * 128, 128, 32 * h0, T, pb);
    }
    for (long task = blockIdx.x; task < groups * 4; task += gridDim.x) {
        // the thread index is made opaque per task: otherwise every piece offset of the copies below (loop-invariant 64-bit
        // values, ~60 of them) is computed before the loop, kept live through it and spilled (176 dwords per lane)
        int tid = threadIdx.x;
#define ATT_OPAQUE() asm volatile("" : "+v"(tid))   /* re-derive the copy offsets here instead of keeping them live */
        ATT_OPAQUE();
        const int ql = tid & 15, tt = tid >> 4;
        const bool act = tt < T;
        const int my_row = ((act ? tt : 0) * 16 + ql) * ATT_LD;
        const long grp = task >> 2;
        const int h = (int)(task & 3);
        const float* base = qkv + grp * T * 16 * 384;
        const float* dob = d_o + grp * T * 16 * 128;
        float* dbase = dqkv + grp * T * 16 * 384;
        __syncthreads();                // the previous task's dK / dV rows have been flushed
        att_put(pa, T, sA, tid);             // Q
        att_put(pb, T, sB, tid);             // dO
        ATT_OPAQUE();
        att_fetch(base, 384, 128 + 32 * h, T, pa, tid);   // K, V: in flight while the rows are picked up
        att_fetch(base, 384, 256 + 32 * h, T, pb, tid);
        __syncthreads();
        f32x4 qv[8], dov[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            qv[d] = ld4(sA + my_row + 4 * d);
            dov[d] = ld4(sB + my_row + 4 * d);
        }
        __syncthreads();
        ATT_OPAQUE();
        att_put(pa, T, sA, tid);             // K
        att_put(pb, T, sB, tid);             // V
        __syncthreads();
        f32x4 dq[8];
        if (act) {  // phase A, tq = tt
            float sc[S3D_N_TOKENS_MAX], dp[S3D_N_TOKENS_MAX];
            float mx = -1e30f;
#pragma unroll
            for (int tk = 0; tk < S3D_N_TOKENS_MAX; ++tk)
                if (tk < T) {
                    const float* krow = sA + (tk * 16 + ql) * ATT_LD;
                    const float* vrow = sB + (tk * 16 + ql) * ATT_LD;
                    float s = 0.f, e = 0.f;
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        const f32x4 kv = ld4(krow + 4 * d), vv = ld4(vrow + 4 * d);
                        s += qv[d][0] * kv[0] + qv[d][1] * kv[1] + qv[d][2] * kv[2] + qv[d][3] * kv[3];
                        e += dov[d][0] * vv[0] + dov[d][1] * vv[1] + dov[d][2] * vv[2] + dov[d][3] * vv[3];
                    }
                    sc[tk] = s * ATT_SCALE;
                    dp[tk] = e;
                    mx = fmaxf(mx, sc[tk]);
                    // one key at a time: left alone, the scheduler hoists the LDS reads of all 13 unrolled keys and spills
                    if (tk & 1) __builtin_amdgcn_sched_barrier(0);
                }
            float den = 0.f;
#pragma unroll
            for (int tk = 0; tk < S3D_N_TOKENS_MAX; ++tk)
                if (tk < T) {
                    sc[tk] = expf(sc[tk] - mx);
                    den += sc[tk];
                }
            const float inv = 1.f / den;
            float dot = 0.f;
            float* pP = sP + (ql * T + tt) * T;
            float* pS = sS + (ql * T + tt) * T;
#pragma unroll
            for (int tk = 0; tk < S3D_N_TOKENS_MAX; ++tk)
                if (tk < T) {
                    sc[tk] *= inv;
                    const float mk = drop.p > 0.f
                                         ? s3d_drop(drop, ((unsigned long long)((grp * T + tt) * 16 + ql) * 4 + h) * 16 + tk)
                                         : 1.f;
                    pP[tk] = sc[tk] * mk;    // dropped probabilities multiply dO in dV
                    dp[tk] *= mk;            // d/dP of sum_k (P*mask)[k] V[k]
                    dot += sc[tk] * dp[tk];
                }
#pragma unroll
            for (int d = 0; d < 8; ++d) dq[d] = zero4();
#pragma unroll
            for (int tk = 0; tk < S3D_N_TOKENS_MAX; ++tk)
                if (tk < T) {
                    const float ds = sc[tk] * (dp[tk] - dot);
                    pS[tk] = ds * ATT_SCALE;
                    const float* krow = sA + (tk * 16 + ql) * ATT_LD;
                    const float w = ds * ATT_SCALE;
#pragma unroll
                    for (int d = 0; d < 8; ++d) dq[d] += ld4(krow + 4 * d) * w;
                    if (tk & 1) __builtin_amdgcn_sched_barrier(0);
                }
        }
        __syncthreads();                // K / V are dead
        if (act) {
#pragma unroll
            for (int d = 0; d < 8; ++d) st4(sB + my_row + 4 * d, dq[d]);
        }
        ATT_OPAQUE();
        att_fetch(base, 384, 32 * h, T, pa, tid);   // Q again (L2), parked for phase B while the dQ rows leave
        __syncthreads();
        ATT_OPAQUE();
        att_flush(sB, T, dbase, 384, 32 * h, tid);  // dQ
        att_put(pa, T, sA, tid);                    // Q
        __syncthreads();
        ATT_OPAQUE();
        att_stage(dob, 128, 32 * h, T, sB, tid);    // dO (L2)
        {
            const long nt = task + gridDim.x;
            if (nt < groups * 4) {             // the next task's Q / dO: requested under phase B
                att_fetch(qkv + (nt >> 2) * T * 16 * 384, 384, 32 * (int)(nt & 3), T, pa, tid);
                att_fetch(d_o + (nt >> 2) * T * 16 * 128, 128, 32 * (int)(nt & 3), T, pb, tid);
            }
        }
        __syncthreads();
        f32x4 dk[8], dv[8];
        if (act) {  // phase B, tk = tt
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                dk[d] = zero4();
                dv[d] = zero4();
            }
            for (int tq = 0; tq < T; ++tq) {
                const float p = sP[(ql * T + tq) * T + tt];
                const float ds = sS[(ql * T + tq) * T + tt];
                const float* qrow = sA + (tq * 16 + ql) * ATT_LD;
                const float* dorow = sB + (tq * 16 + ql) * ATT_LD;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    dk[d] += ld4(qrow + 4 * d) * ds;
                    dv[d] += ld4(dorow + 4 * d) * p;
                }
            }
        }
        __syncthreads();                // Q / dO are dead
        if (act) {
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                st4(sA + my_row + 4 * d, dk[d]);
                st4(sB + my_row + 4 * d, dv[d]);
            }
        }
        __syncthreads();
        ATT_OPAQUE();
        att_flush(sA, T, dbase, 384, 128 + 32 * h, tid);   // dK
        att_flush(sB, T, dbase, 384, 256 + 32 * h, tid);   // dV
#undef ATT_OPAQUE
    }
}

int launch_attn_core_bwd(const float* qkv, const float* d_o, float* dqkv, long groups, int T, const DropCfg& drop,
                         hipStream_t stream) {
    if (groups <= 0) return 0;
    S3D_CHECK_ARG(T >= 1 && T <= S3D_N_TOKENS_MAX, "attn core: T %d", T);
    const long nb = groups * 4 < 16384 ? groups * 4 : 16384;
    hipLaunchKernelGGL(attn_core_bwd_kernel, dim3((unsigned)nb), dim3(ATT_THREADS), 0, stream, qkv, d_o, dqkv, groups, T,
                       drop);
    S3D_LAUNCH_CHECK();
    return 0;
}
