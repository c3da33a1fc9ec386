// conv.hip — implicit-GEMM convolution engine for the U-Net slice generator (gfx950, fp32 MFMA).
//
// Replaces the stock ATen ops the reference calls in reg_slices/src/unet_custom.py:40-69 and
// reg_slices/src/unet_parts.py:8-84: conv2d 3x3 (pad 1) / 1x1, conv_transpose2d 2x2 s2, eval-mode
// batch_norm, relu, max_pool2d 2x2, tanh, torch.cat along channels and the batch tiling of expand_bs.
//
// GEMM view (swapped form, see common.h):  out^T[co][pixel] = sum_k Wp[co][k] * patch[pixel][k],
//   M = pixels of all images (flattened, so any H/W incl. 2x2 maps works), N = Cout,
//   K = sum over sources of ks*ks*C   (k order: source, tap, channel).
// A fragments (weights) come from the pre-packed lane-linear image (1 KiB contiguous per fragment,
// L2-resident); B fragments (activations) are 16-byte per-lane loads from the NHWC tensors, zero for
// padding taps.  fp32 MFMA runs at 256 FLOP/clk/CU, i.e. one 16-byte operand load feeds 128 cycles of
// MFMA per (MT|NT)-fold reuse, so the operand streams fit the L1/L2 path at this precision.
#include <stdlib.h>
#include "conv.h"
#include <vector>

template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvLaunch& a, f32x4 (&acc)[MT][NT], const int (&pn)[MT],
                                              const int (&py)[MT], const int (&px)[MT], const bool (&pv)[MT],
                                              int jt0, int g) {
    // epilogue: lane owns channels co..co+3 of pixel (pn,py,px)
    const bool convt32 = a.out_mode == S3D_OUT_CONVT && a.cout_store % 32 == 0;   // a 32-channel pair stays inside one quadrant
    if (NT % 2 == 0 && (a.out_mode == S3D_OUT_NHWC || convt32)) {
        // NHWC rows leave as full 128-byte lines (s3d_full_line_pair, common.h): the accumulator pair (nt, nt + 1) of pixel m
        // is exchanged with lane m ^ 8 FIRST; the lane then owns channels cpair .. cpair + 3 of pixel m & 7 and of pixel
        // (m & 7) + 8 of the tile and runs the same per-element epilogue as below on them (every step of it — affine,
        // activation, dropout, gate, residual — is a function of the output index alone, so the stored bits do not
        // change).  One instruction writes 32 channels of 8 pixels as whole lines; the plain form writes 64-byte runs of
        // 16 pixels (3.10 -> 2.59 ms on the 128 -> 384 row-linear layer, tools/lin_abl.sh).
        const int m = threadIdx.x & 15;
        const bool lo = m < 8;
#pragma unroll
        for (int np = 0; np < NT / 2; ++np) {
            const int co = (jt0 + 2 * np) * 16 + 16 * (m >> 3) + 4 * g;
            const f32x4 sc = a.scale ? ld4(a.scale + co) : f32x4{1.f, 1.f, 1.f, 1.f};
            const f32x4 sh = a.shift ? ld4(a.shift + co) : zero4();
            // ConvTranspose2d(2, stride 2): packed channel co = quadrant q * ct + c lands at output pixel (2y + q/2, 2x + q%2)
            const int ct = a.cout_store, cq = convt32 ? co / ct : 0, ccol = convt32 ? co - cq * ct : co;
            const bool col_ok = convt32 ? cq < 4 : co < a.cout_store;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 val[2];
                s3d_full_line_pair(acc[mt][2 * np], acc[mt][2 * np + 1], m, val[0], val[1]);
                // the partner lane's pixel: its output offset and validity travel the same way
                const long ob = convt32 ? ((long)(pn[mt] * 2 * a.H + 2 * py[mt] + (cq >> 1)) * (2 * a.W) + 2 * px[mt] + (cq & 1)) * ct
                                        : ((long)(pn[mt] * a.H + py[mt]) * a.W + px[mt]) * a.out_cstride;
                const unsigned ob_lo = s3d_row_ror8_u32((unsigned)ob), ob_hi = s3d_row_ror8_u32((unsigned)((unsigned long long)ob >> 32));
                const bool pvp = s3d_row_ror8_u32(pv[mt] ? 1u : 0u) != 0u;
                const long obp = (long)(((unsigned long long)ob_hi << 32) | ob_lo);
#pragma unroll
                for (int k = 0; k < 2; ++k) {   // k = 0: pixel m & 7, k = 1: pixel (m & 7) + 8
                    const bool own = (k == 0) == lo;
                    if (!(own ? pv[mt] : pvp) || !col_ok) continue;
                    const long oi = (own ? ob : obp) + ccol;
                    f32x4 v = val[k] * sc + sh;
                    if (a.act == S3D_ACT_RELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                    } else if (a.act == S3D_ACT_TANH) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = tanhf(v[i]);
                    }
                    if (a.drop.p > 0.f) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] *= s3d_drop(a.drop, a.drop_base + (unsigned long long)(oi + i));
                    }
                    if (a.gate) {
                        const f32x4 gt = ld4(a.gate + oi);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = gt[i] > 0.f ? v[i] * a.gate_scale : 0.f;
                    }
                    if (a.residual) v += ld4(a.residual + oi);
                    if (a.out_accumulate) v += ld4(a.out + oi);
                    // streaming store: activations are far larger than L2, keeping them out of it measured faster
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.out + oi));
                }
                __builtin_amdgcn_sched_barrier(0);   // one pixel tile at a time
            }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (jt0 + nt) * 16 + 4 * g;
        const f32x4 sc = a.scale ? ld4(a.scale + co) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 sh = a.shift ? ld4(a.shift + co) : zero4();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (!pv[mt]) continue;
            f32x4 v = acc[mt][nt] * sc + sh;
            if (a.act == S3D_ACT_RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
            } else if (a.act == S3D_ACT_TANH) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = tanhf(v[i]);
            }
            if (a.out_mode == S3D_OUT_NHWC) {
                if (co < a.cout_store) {  // cout_store is a multiple of 4 in this mode
                    const long oi = ((long)(pn[mt] * a.H + py[mt]) * a.W + px[mt]) * a.out_cstride + co;
                    if (a.drop.p > 0.f) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] *= s3d_drop(a.drop, a.drop_base + (unsigned long long)(oi + i));
                    }
                    if (a.gate) {
                        const f32x4 gt = ld4(a.gate + oi);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = gt[i] > 0.f ? v[i] * a.gate_scale : 0.f;
                    }
                    if (a.residual) v += ld4(a.residual + oi);
                    if (a.out_accumulate) v += ld4(a.out + oi);
                    // streaming store: activations are far larger than L2, keeping them out of it measured faster
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.out + oi));
                }
            } else if (a.out_mode == S3D_OUT_CONVT) {
                const int ct = a.cout_store;  // multiple of 16: a lane's 4 channels share a quadrant
                const int q = co / ct, c = co - q * ct;
                if (q < 4) {
                    const int oy = 2 * py[mt] + (q >> 1), ox = 2 * px[mt] + (q & 1);
                    float* o = a.out + ((long)(pn[mt] * 2 * a.H + oy) * (2 * a.W) + ox) * ct + c;
                    st4(o, v);
                }
            } else {  // NCHW, arbitrary cout_store
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (co + i < a.cout_store)
                        a.out[((long)(pn[mt] * a.cout_store + co + i) * a.H + py[mt]) * a.W + px[mt]] = v[i];
            }
        }
    }
}

// Workgroups are dealt to the 8 XCDs round-robin by linear block id.  The cout tiles of one pixel tile read the same
// input: mapped to consecutive block ids they land on different XCDs and every L2 fetches that input again (up to
// CoutPad / CO_WG times).  Remapped, XCD x owns the pixel tiles x, x+8, ... and runs all cout tiles of a pixel tile
// back to back, so the input comes from HBM once.
__device__ __forceinline__ void conv_block_tile(const ConvLaunch& a, int n_co_blk, int& blk_co, long& blk_px) {
    const long L = blockIdx.x;
    if (!a.xcd_remap || n_co_blk == 1) {
        blk_co = (int)(L % n_co_blk);
        blk_px = L / n_co_blk;
        return;
    }
    const long n_px = gridDim.x / n_co_blk, full = n_px & ~7L, lim = full * n_co_blk;
    if (L < lim) {
        const long k = L >> 3;
        blk_co = (int)(k % n_co_blk);
        blk_px = (k / n_co_blk) * 8 + (L & 7);
    } else {
        const long r = L - lim;
        blk_co = (int)(r % n_co_blk);
        blk_px = full + r / n_co_blk;
    }
}

template <int MT, int NT, int WM, int WN, int KS>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(const ConvLaunch a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int m = lane & 15, g = lane >> 4;
    constexpr int PIX_WG = WM * MT * 16;
    constexpr int CO_WG = WN * NT * 16;
    const long P = (long)a.N * a.H * a.W;
    const int n_co_blk = a.CoutPad / CO_WG;
    int blk_co;
    long blk_px;
    conv_block_tile(a, n_co_blk, blk_co, blk_px);
    const int HW = a.H * a.W;
    const int stride = a.stride > 1 ? a.stride : 1;
    const int Hin = a.Hin ? a.Hin : a.H, Win = a.Win ? a.Win : a.W;
    constexpr int PAD = KS == 3 ? 1 : 0;

    int pn[MT], py[MT], px[MT];
    bool pv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long p = blk_px * PIX_WG + (long)(wm * MT + mt) * 16 + m;
        pv[mt] = p < P;
        const long pc = pv[mt] ? p : 0;
        pn[mt] = (int)(pc / HW);
        const int r = (int)(pc - (long)pn[mt] * HW);
        py[mt] = r / a.W;
        px[mt] = r - py[mt] * a.W;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zero4();

    const int jt0 = blk_co * (CO_WG / 16) + wn * NT;
    int ubase = 0;
#pragma unroll 1
    for (int s = 0; s < a.nsrc; ++s) {
        const ConvSrc S = a.src[s];
        const int cu = S.C >> 4;
#pragma unroll 1
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int dy = tap / KS - PAD, dx = tap % KS - PAD;
            const float* bp[MT];
            bool ok[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int y = py[mt] * stride + dy, x = px[mt] * stride + dx;
                ok[mt] = pv[mt] && y >= 0 && y < Hin && x >= 0 && x < Win;
                const int ni = (S.bmod ? pn[mt] % S.bmod : pn[mt]) / S.bdiv;
                const long off = S.sbcast ? (long)ni * S.C : ((long)(ni * Hin + y) * Win + x) * S.C;
                bp[mt] = S.p + (ok[mt] ? off : 0) + 4 * g;
            }
            const float* wp = frag_ptr(a.wpk, jt0, ubase + tap * cu, a.KU, lane);
#pragma unroll 2
            for (int c = 0; c < cu; ++c) {
                f32x4 b[MT], w[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) b[mt] = ok[mt] ? ld4(bp[mt] + 16 * c) : zero4();
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) w[nt] = ld4(wp + ((size_t)nt * a.KU + c) * 256);
                // Every 16-deep K chunk is summed on its own (a 16-term fmaf chain from zero) and then added: the accumulation over K is
                // K / 16 additions of short partial sums instead of ONE fp32 chain of up to 4 608 terms.  Round 6 (profiles/
                // r06_f32_chains.md): with the single chain the weight gradients of the encoder's convolutions — sums with heavy
                // cancellation behind train-mode BatchNorm — sat 7e-3 .. 1e-2 from the fp64 oracle in this "exact" mode (the
                // split-precision mode, whose 32-deep f16 MFMA rounds once per 32 products, and ATen's blocked CPU kernels: 2.5e-3);
                // with the chunked sum: 2.3e-3.  MT x NT packed adds per 4 MFMAs of 32 cycles: free.
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] += mfma4(w[nt], b[mt], zero4());
            }
        }
        ubase += KS * KS * cu;
    }

    conv_epilogue<MT, NT>(a, acc, pn, py, px, pv, jt0, g);
}

// ---------------------------------------------------------------------------------------------
// split-precision variant: same tiling, 32-deep K chunks, A = pre-packed f16 hi/lo fragment pairs,
// B = fp32 activations split into hi/lo on the fly, 3 f16 MFMAs per product (see decode_f16.hip).
// ---------------------------------------------------------------------------------------------
typedef _Float16 chalf8 __attribute__((ext_vector_type(8)));

template <int MT, int NT, int WM, int WN, int KS>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_f16x3_kernel(const ConvLaunch a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int m = lane & 15, g = lane >> 4;
    constexpr int PIX_WG = WM * MT * 16;
    constexpr int CO_WG = WN * NT * 16;
    const long P = (long)a.N * a.H * a.W;
    const int n_co_blk = a.CoutPad / CO_WG;
    int blk_co;
    long blk_px;
    conv_block_tile(a, n_co_blk, blk_co, blk_px);
    const int HW = a.H * a.W;
    const int stride = a.stride > 1 ? a.stride : 1;
    const int Hin = a.Hin ? a.Hin : a.H, Win = a.Win ? a.Win : a.W;
    constexpr int PAD = KS == 3 ? 1 : 0;
    const int KU32 = a.KU >> 1;
    const _Float16* wimg = reinterpret_cast<const _Float16*>(a.wpk16);

    int pn[MT], py[MT], px[MT];
    bool pv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long p = blk_px * PIX_WG + (long)(wm * MT + mt) * 16 + m;
        pv[mt] = p < P;
        const long pc = pv[mt] ? p : 0;
        pn[mt] = (int)(pc / HW);
        const int r = (int)(pc - (long)pn[mt] * HW);
        py[mt] = r / a.W;
        px[mt] = r - py[mt] * a.W;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zero4();

    const int jt0 = blk_co * (CO_WG / 16) + wn * NT;
    int ubase = 0;
#pragma unroll 1
    for (int s = 0; s < a.nsrc; ++s) {
        const ConvSrc S = a.src[s];
        const int cu = S.C >> 5;
#pragma unroll 1
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int dy = tap / KS - PAD, dx = tap % KS - PAD;
            const float* bp[MT];
            bool ok[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int y = py[mt] * stride + dy, x = px[mt] * stride + dx;
                ok[mt] = pv[mt] && y >= 0 && y < Hin && x >= 0 && x < Win;
                const int ni = (S.bmod ? pn[mt] % S.bmod : pn[mt]) / S.bdiv;
                const long off = S.sbcast ? (long)ni * S.C : ((long)(ni * Hin + y) * Win + x) * S.C;
                bp[mt] = S.p + (ok[mt] ? off : 0) + 8 * g;
            }
            const _Float16* wp = wimg + ((size_t)jt0 * KU32 + ubase + tap * cu) * 1024 + lane * 8;
            // Every load of a K chunk (2 per activation tile, 2 per weight tile) is issued before its first use, and the NEXT
            // chunk's loads before this chunk's MFMAs (two register sets, the loop unrolled by two): one chunk per L2 round
            // trip made the small GEMMs of the LDM U-Net at batch 1 (32 x 32-pixel tiles, 20 chunks, 2.5 waves per SIMD) pure
            // latency — 15 us for 1.3 GMAC.
            auto issue = [&](int c, f32x4 (&v0)[MT], f32x4 (&v1)[MT], chalf8 (&wh)[NT], chalf8 (&wl)[NT]) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    v0[mt] = ok[mt] ? ld4(bp[mt] + 32 * c) : zero4();
                    v1[mt] = ok[mt] ? ld4(bp[mt] + 32 * c + 4) : zero4();
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const _Float16* f = wp + ((size_t)nt * KU32 + c) * 1024;
                    wh[nt] = *reinterpret_cast<const chalf8*>(f);
                    wl[nt] = *reinterpret_cast<const chalf8*>(f + 512);
                }
            };
            auto consume = [&](const f32x4 (&v0)[MT], const f32x4 (&v1)[MT], const chalf8 (&wh)[NT], const chalf8 (&wl)[NT]) {
                chalf8 bh[MT], bl[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const _Float16 h0 = (_Float16)v0[mt][t], h1 = (_Float16)v1[mt][t];
                        bh[mt][t] = h0; bh[mt][4 + t] = h1;
                        bl[mt][t] = (_Float16)(v0[mt][t] - (float)h0);
                        bl[mt][4 + t] = (_Float16)(v1[mt][t] - (float)h1);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (!a.single_pass) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], bl[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt], bh[mt], acc[mt][nt], 0, 0, 0);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], bh[mt], acc[mt][nt], 0, 0, 0);
                }
            };
            f32x4 p0[MT], p1[MT];
            chalf8 pwh[NT], pwl[NT];
            if constexpr (MT * NT <= 8) {   // (the 4 x 4 tile would need 300 registers for two sets: it has the MFMA work to cover a round trip)
                f32x4 q0[MT], q1[MT];
                chalf8 qwh[NT], qwl[NT];
                issue(0, p0, p1, pwh, pwl);
                int c = 0;
#pragma unroll 1
                for (; c + 1 < cu; c += 2) {
                    issue(c + 1, q0, q1, qwh, qwl);
                    __builtin_amdgcn_sched_barrier(0);
                    consume(p0, p1, pwh, pwl);
                    if (c + 2 < cu) issue(c + 2, p0, p1, pwh, pwl);
                    __builtin_amdgcn_sched_barrier(0);
                    consume(q0, q1, qwh, qwl);
                }
                if (c < cu) {
                    __builtin_amdgcn_sched_barrier(0);
                    consume(p0, p1, pwh, pwl);
                }
            } else {
#pragma unroll 1
                for (int c = 0; c < cu; ++c) {
                    issue(c, p0, p1, pwh, pwl);
                    __builtin_amdgcn_sched_barrier(0);
                    consume(p0, p1, pwh, pwl);
                }
            }
        }
        ubase += KS * KS * cu;
    }
    conv_epilogue<MT, NT>(a, acc, pn, py, px, pv, jt0, g);
}

// ---------------------------------------------------------------------------------------------
// Row-linear layers (1x1 convolution of ONE plain source with K <= 128 input channels) on long row sets: the decoder's
// attention-block projections, the pyramid level projections.  The implicit-GEMM kernel above walks (pixel tile, cout
// tile) blocks and re-requests its few K chunks block by block (3.2 TB/s on 1.3 M rows); here a wave owns 48 rows,
// requests ALL of their K channels at once (24 x 16 B per lane in flight), keeps them in registers as f16 hi/lo
// fragments and walks the output channels 32 at a time — every input byte is requested once, weights come from L2.
// No LDS, no barriers.  Same accumulation order as conv_igemm_f16x3_kernel, same epilogue.
// ---------------------------------------------------------------------------------------------
#define LR_MT 3
__global__ __launch_bounds__(256, 2) void lin_rows_f16x3_kernel(const ConvLaunch a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const long P = (long)a.N * a.H * a.W;
    const int HW = a.H * a.W;
    const ConvSrc S = a.src[0];
    const int KU32 = S.C >> 5;   // 1 .. 4
    const _Float16* wimg = reinterpret_cast<const _Float16*>(a.wpk16);
    int pn[LR_MT], py[LR_MT], px[LR_MT];
    bool pv[LR_MT];
    f32x4 v0[LR_MT][4], v1[LR_MT][4];
#pragma unroll
    for (int mt = 0; mt < LR_MT; ++mt) {
        const long p = ((long)blockIdx.x * 4 + wave) * (16 * LR_MT) + 16 * mt + m;
        pv[mt] = p < P;
        const long pc = pv[mt] ? p : 0;
        pn[mt] = (int)(pc / HW);
        const int r = (int)(pc - (long)pn[mt] * HW);
        py[mt] = r / a.W;
        px[mt] = r - py[mt] * a.W;
        const float* row = S.p + pc * S.C + 8 * g;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = pv[mt] && u < KU32;
            v0[mt][u] = ok ? ld4(row + 32 * u) : zero4();
            v1[mt][u] = ok ? ld4(row + 32 * u + 4) : zero4();
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // every row request is out before the first conversion waits
    // split in place: v0 becomes the hi fragment, v1 the lo fragment (same registers: 128 for the rows, not 256)
#pragma unroll
    for (int mt = 0; mt < LR_MT; ++mt)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            chalf8 h, l;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const _Float16 h0 = (_Float16)v0[mt][u][t], h1 = (_Float16)v1[mt][u][t];
                h[t] = h0; h[4 + t] = h1;
                l[t] = (_Float16)(v0[mt][u][t] - (float)h0);
                l[4 + t] = (_Float16)(v1[mt][u][t] - (float)h1);
            }
            v0[mt][u] = __builtin_bit_cast(f32x4, h);
            v1[mt][u] = __builtin_bit_cast(f32x4, l);
        }
#define LR_XH(mt, u) __builtin_bit_cast(chalf8, v0[mt][u])
#define LR_XL(mt, u) __builtin_bit_cast(chalf8, v1[mt][u])
    const int n_jt = a.CoutPad >> 4;
#pragma unroll 1
    for (int jt = 0; jt < n_jt; jt += 2) {
        f32x4 acc[LR_MT][2];
#pragma unroll
        for (int mt = 0; mt < LR_MT; ++mt) acc[mt][0] = acc[mt][1] = zero4();
        const _Float16* wp = wimg + (size_t)jt * KU32 * 1024 + lane * 8;
        chalf8 wh[2][2], wl[2][2];   // [buffer][nt]: the next k-step's fragments are requested before this one's MFMAs
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            wh[0][nt] = *reinterpret_cast<const chalf8*>(wp + (size_t)nt * KU32 * 1024);
            wl[0][nt] = *reinterpret_cast<const chalf8*>(wp + (size_t)nt * KU32 * 1024 + 512);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u < KU32) {
                if (u + 1 < KU32) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const _Float16* f = wp + ((size_t)nt * KU32 + u + 1) * 1024;
                        wh[(u + 1) & 1][nt] = *reinterpret_cast<const chalf8*>(f);
                        wl[(u + 1) & 1][nt] = *reinterpret_cast<const chalf8*>(f + 512);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (!a.single_pass) {
#pragma unroll
                        for (int mt = 0; mt < LR_MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[u & 1][nt], LR_XL(mt, u), acc[mt][nt], 0, 0, 0);
#pragma unroll
                        for (int mt = 0; mt < LR_MT; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[u & 1][nt], LR_XH(mt, u), acc[mt][nt], 0, 0, 0);
                    }
#pragma unroll
                    for (int mt = 0; mt < LR_MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[u & 1][nt], LR_XH(mt, u), acc[mt][nt], 0, 0, 0);
                }
            }
        }
        conv_epilogue<LR_MT, 2>(a, acc, pn, py, px, pv, jt, g);
    }
}

static bool lin_rows_eligible(const ConvLaunch& a) {
    if (a.ks != 1 || a.nsrc != 1 || a.stride > 1 || !a.wpk16 || (a.KU & 1) || a.CoutPad % 32) return false;
    if ((a.Hin && a.Hin != a.H) || (a.Win && a.Win != a.W)) return false;
    const ConvSrc& S = a.src[0];
    if (S.sbcast || S.bmod || S.bdiv != 1 || S.C % 32 || S.C > 128 || a.KU != S.C / 16) return false;
    return (long)a.N * a.H * a.W >= 65536;   // short row sets stay on the tile menu (more, smaller workgroups)
}
static int launch_lin_rows(const ConvLaunch& a, hipStream_t stream) {
    const long P = (long)a.N * a.H * a.W;
    const long nblk = (P + 64 * LR_MT - 1) / (64 * LR_MT);
    S3D_CHECK_ARG(nblk < (1L << 31), "conv grid out of range (%ld)", nblk);
    hipLaunchKernelGGL(lin_rows_f16x3_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, a);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Row-linear layers on the long row sets of the training step (K and Cout multiples of 128: the attention block's in_proj
// 128 -> 384, its data gradient 384 -> 128, out_proj, the last layer's K|V gradient 256 -> 128), streaming form.
// lin_rows above re-requests all weight fragments from L2 for every 48 rows of every wave (196 KB per 48 rows for in_proj:
// 21 GB of L2 traffic per call) and its MFMA phases wait on them: with the stores removed the 128 -> 384 call still took
// 2.3 ms for 2.7 GB of input (tools/lin_abl.sh).  Here a persistent workgroup (4 waves x 48 rows; 32 for K > 128) streams the weight image
// through a three-slot LDS ring of 16 KiB slots (slot = 128 input channels x 32 output channels, hi | lo fragments) with
// LDS-DMA two phases ahead, so a weight byte leaves L2 once per 192 rows and the fragments of a phase come out of LDS.
// A wave keeps the 128 input channels of its rows in registers as f16 hi / lo (requested all at once); for K > 128 the
// accumulators of all 128 outputs stay live across the K chunks.  One s_barrier per phase (72 MFMAs per wave).
//
// vmcnt bookkeeping: a phase issues [DMA of phase + 2][MFMAs][s_waitcnt vmcnt(4)][stores of this phase].  The wait leaves
// only the four DMA instructions just issued outstanding, i.e. the DMA of phase + 1 (a phase old) has landed and the
// stores of the phase before have retired — whatever their number (tail rows store nothing) — before the wave arrives
// at the next barrier; the stores of this phase get a whole phase to retire.  LDS is read by hand-issued ds_read_b128
// with counted lgkmcnt waits only (no compiler-visible LDS access: nothing makes hipcc guard them with vmcnt(0)).
// Per accumulator the products are added in the order of lin_rows / conv_igemm_f16x3 (k ascending; hi*lo, lo*hi, hi*hi):
// the same bits.
// ---------------------------------------------------------------------------------------------
#define LS_SLOT_HALFS 8192
#define LS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define LS_WAIT4(n, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(n))
#define LS_GLOAD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
// the epilogue of conv_epilogue restricted to what these calls use (bias, ReLU, output dropout, residual; NHWC rows:
// the output index is the flat row index times the channel stride) — the general one costs 50 registers more
template <int MT, bool DROP, bool AFFINE>
__device__ __forceinline__ void lin_stream_epilogue(const ConvLaunch& a, const f32x4 (&acc)[MT][2], const f32x4 (&res)[MT][2],
                                                    const f32x4 (&sh)[2], const long (&p)[MT], long tile0, long P, int jt0,
                                                    int m, int g) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 v[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            v[nt] = AFFINE ? acc[mt][nt] + sh[nt] : acc[mt][nt];
            if (a.act == S3D_ACT_RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[nt][i] = fmaxf(v[nt][i], 0.f);
            }
            if (DROP && a.drop.p > 0.f) {
                const long oi = p[mt] * a.out_cstride + (jt0 + nt) * 16 + 4 * g;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[nt][i] *= s3d_drop(a.drop, a.drop_base + (unsigned long long)(oi + i));
            }
            if (a.residual) v[nt] += res[mt][nt];
        }
        // full 128-byte lines: rows 0-7 of the tile in one instruction, rows 8-15 in the next (s3d_full_line_pair)
        f32x4 oa, ob;
        s3d_full_line_pair(v[0], v[1], m, oa, ob);
        const long ra = tile0 + 16 * mt + (m & 7);
        const int co = jt0 * 16 + 16 * (m >> 3) + 4 * g;
        if (co < a.cout_store) {
            if (ra < P) __builtin_nontemporal_store(oa, reinterpret_cast<f32x4*>(a.out + ra * a.out_cstride + co));
            if (ra + 8 < P) __builtin_nontemporal_store(ob, reinterpret_cast<f32x4*>(a.out + (ra + 8) * a.out_cstride + co));
        }
    }
}

template <int KC>
__global__ __launch_bounds__(256, 2) void lin_stream_f16x3_kernel(const ConvLaunch a, long n_tasks) {
    __shared__ __attribute__((aligned(16))) _Float16 s_w[3 * LS_SLOT_HALFS];
    constexpr int NSL = KC > 1 ? 4 : 1;   // output slots whose accumulators stay live across the K chunks
    constexpr int LS_MT = KC > 1 ? 2 : 3;  // row tiles per wave: 128 live accumulator registers at most
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    const long P = (long)a.N * a.H * a.W;
    const ConvSrc S = a.src[0];
    const int KU32 = S.C >> 5;
    const int n_slots = KC > 1 ? 4 : a.CoutPad >> 5;
    const int ppt = KC * n_slots;   // phases per task
    const _Float16* wimg = reinterpret_cast<const _Float16*>(a.wpk16);
    const unsigned lw0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_w) + lane * 16;
    // a wave's share of a slot: fragments (nt = wave >> 1, u = 2 (wave & 1) + {0, 1}), hi | lo each = 4 KiB contiguous in the
    // image and in the slot (fragment f = 4 nt + u at f * 2 KiB)
    auto dma_slot = [&](int q, int slot) {   // q = kc * n_slots + ns: phase within a task
        const int kc = KC > 1 ? q >> 2 : 0, ns = KC > 1 ? q & 3 : q;
        const __attribute__((address_space(1))) void* gp = (const __attribute__((address_space(1))) void*)(
            wimg + ((size_t)(2 * ns + (wave >> 1)) * KU32 + 4 * kc + 2 * (wave & 1)) * 1024 + lane * 8);
        __attribute__((address_space(3))) void* lp =
            (__attribute__((address_space(3))) void*)(s_w + slot * LS_SLOT_HALFS + wave * 2048);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
    };
    int slot = 0, q = 0;   // ring slot and task phase of the NEXT phase to run
    dma_slot(0, 0);
    dma_slot(1 % ppt, 1);
    bool first = true;
#pragma unroll 1
    for (long task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        long prow[LS_MT];
        bool pv[LS_MT];
        const float* row[LS_MT];
#pragma unroll
        for (int mt = 0; mt < LS_MT; ++mt) {
            const long p = (task * 4 + wave) * (16 * LS_MT) + 16 * mt + m;
            pv[mt] = p < P;
            prow[mt] = pv[mt] ? p : 0;
            row[mt] = S.p + prow[mt] * S.C + 8 * g;
        }
        f32x4 acc[NSL][LS_MT][2];
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int mt = 0; mt < LS_MT; ++mt) acc[sl][mt][0] = acc[sl][mt][1] = zero4();
#pragma unroll 1
        for (int kc = 0; kc < KC; ++kc) {
            // the rows' 128 channels of this K chunk: all 24 requests out, then split in place (v0 -> hi, v1 -> lo)
            f32x4 v0[LS_MT][4], v1[LS_MT][4];
#pragma unroll
            for (int mt = 0; mt < LS_MT; ++mt)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v0[mt][u] = ld4(row[mt] + 128 * kc + 32 * u);   // rows past the end read row 0 and store nothing
                    v1[mt][u] = ld4(row[mt] + 128 * kc + 32 * u + 4);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < LS_MT; ++mt)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float x[8] = {v0[mt][u][0], v0[mt][u][1], v0[mt][u][2], v0[mt][u][3],
                                        v1[mt][u][0], v1[mt][u][1], v1[mt][u][2], v1[mt][u][3]};
                    s3d_half8 h, l;
                    s3d_split8(x, h, l);
                    v0[mt][u] = __builtin_bit_cast(f32x4, h);
                    v1[mt][u] = __builtin_bit_cast(f32x4, l);
                }
#define LS_XH(mt, u) __builtin_bit_cast(chalf8, v0[mt][u])
#define LS_XL(mt, u) __builtin_bit_cast(chalf8, v1[mt][u])
            if (first) {   // the two prologue slots
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                first = false;
            }
#pragma unroll
            for (int sl = 0; sl < (KC > 1 ? 4 : 1); ++sl) {
#pragma unroll 1
                for (int ns = (KC > 1 ? sl : 0); ns < (KC > 1 ? sl + 1 : n_slots); ++ns) {
                    // ---- one phase: slot `slot` holds (kc, ns) ----
                    asm volatile("s_barrier" ::: "memory");   // every wave's share of this slot has landed; slot + 2 is free
                    // residual rows of this phase's outputs: requested BEFORE the DMA (the vmcnt(4) below then covers them)
                    // residual rows and bias of this phase's outputs: requested BEFORE the DMA so that the vmcnt(4)
                    // below covers them, and by hand — hipcc waits with vmcnt(0) for a load of its own that has LDS-DMA
                    // requests behind it, i.e. for the weights of phase + 2
                    f32x4 res[LS_MT][2], sh[2];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        sh[nt] = zero4();
#pragma unroll
                        for (int mt = 0; mt < LS_MT; ++mt) res[mt][nt] = zero4();
                    }
                    if (KC == 1) {   // K > 128 serves plain products (no bias: see lin_stream_eligible)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const int co = (2 * ns + nt) * 16 + 4 * g;
                            if (a.shift) LS_GLOAD(sh[nt], a.shift + co);
                        }
                    }
                    if (kc == KC - 1 && a.residual) {
#pragma unroll
                        for (int mt = 0; mt < LS_MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                LS_GLOAD(res[mt][nt], a.residual + prow[mt] * a.out_cstride + (2 * ns + nt) * 16 + 4 * g);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        int q2 = q + 2;
                        if (q2 >= ppt) q2 -= ppt;
                        if (q2 >= ppt) q2 -= ppt;   // ppt = 1: a single 32-channel output slot
                        int s2 = slot + 2;
                        if (s2 >= 3) s2 -= 3;
                        dma_slot(q2, s2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned lwa = lw0 + (unsigned)slot * (LS_SLOT_HALFS * 2);
                    chalf8 wh[2][2], wl[2][2];   // [buffer][nt]
#define LS_READS(B, U)                                  \
    LS_READ(wh[B][0], lwa, (U) * 2048);                 \
    LS_READ(wl[B][0], lwa, (U) * 2048 + 1024);          \
    LS_READ(wh[B][1], lwa, (4 + (U)) * 2048);           \
    LS_READ(wl[B][1], lwa, (4 + (U)) * 2048 + 1024);
#define LS_MFMA(B, U)                                                                                                  \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                                 \
        if (!a.single_pass) {                                                                                          \
            _Pragma("unroll") for (int mt = 0; mt < LS_MT; ++mt) acc[sl][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16( \
                wh[B][nt], LS_XL(mt, U), (KC == 1 && (U) == 0) ? zero4() : acc[sl][mt][nt], 0, 0, 0);                  \
            _Pragma("unroll") for (int mt = 0; mt < LS_MT; ++mt) acc[sl][mt][nt] =                                     \
                __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[B][nt], LS_XH(mt, U), acc[sl][mt][nt], 0, 0, 0);             \
            _Pragma("unroll") for (int mt = 0; mt < LS_MT; ++mt) acc[sl][mt][nt] =                                     \
                __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[B][nt], LS_XH(mt, U), acc[sl][mt][nt], 0, 0, 0);             \
        } else {                                                                                                       \
            _Pragma("unroll") for (int mt = 0; mt < LS_MT; ++mt) acc[sl][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16( \
                wh[B][nt], LS_XH(mt, U), (KC == 1 && (U) == 0) ? zero4() : acc[sl][mt][nt], 0, 0, 0);                  \
        }                                                                                                              \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);
                    LS_READS(0, 0)
                    LS_READS(1, 1)
                    LS_WAIT4(4, wh[0][0], wl[0][0], wh[0][1], wl[0][1]);
                    LS_MFMA(0, 0)
                    LS_READS(0, 2)
                    LS_WAIT4(4, wh[1][0], wl[1][0], wh[1][1], wl[1][1]);
                    LS_MFMA(1, 1)
                    LS_READS(1, 3)
                    LS_WAIT4(4, wh[0][0], wl[0][0], wh[0][1], wl[0][1]);
                    LS_MFMA(0, 2)
                    LS_WAIT4(0, wh[1][0], wl[1][0], wh[1][1], wl[1][1]);
                    LS_MFMA(1, 3)
#undef LS_READS
#undef LS_MFMA
                    // DMA of the next phase landed, stores of the previous phase retired (see the header)
                    if (KC == 1)
                        asm volatile("s_waitcnt vmcnt(4)"
                                     : "+v"(res[0][0]), "+v"(res[0][1]), "+v"(res[1][0]), "+v"(res[1][1]), "+v"(res[LS_MT - 1][0]),
                                       "+v"(res[LS_MT - 1][1]), "+v"(sh[0]), "+v"(sh[1])
                                     :
                                     : "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(4)"
                                     : "+v"(res[0][0]), "+v"(res[0][1]), "+v"(res[1][0]), "+v"(res[1][1])
                                     :
                                     : "memory");
                    if (kc == KC - 1) lin_stream_epilogue<LS_MT, KC == 1, KC == 1>(a, acc[sl], res, sh, prow, (task * 4 + wave) * (16 * LS_MT), P,
                                                                         2 * ns, m, g);
                    if (++q == ppt) q = 0;
                    if (++slot == 3) slot = 0;
                }
            }
#undef LS_XH
#undef LS_XL
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last prefetches must not outlive the workgroup's LDS
}

static bool lin_stream_eligible(const ConvLaunch& a) {
    if (a.ks != 1 || a.nsrc != 1 || a.stride > 1 || !a.wpk16 || a.CoutPad % 32) return false;
    if ((a.Hin && a.Hin != a.H) || (a.Win && a.Win != a.W)) return false;
    const ConvSrc& S = a.src[0];
    if (S.sbcast || S.bmod || S.bdiv != 1 || S.C % 128 || S.C > 384 || a.KU != S.C / 16) return false;
    // K > 128: all 128 outputs live; plain products only (no scale / bias / output dropout)
    if (S.C > 128 && (a.CoutPad != 128 || a.drop.p > 0.f || a.scale || a.shift)) return false;
    // cout_store == CoutPad: the residual rows of a 32-channel output slot are requested without a per-channel guard
    // (hand-issued loads), so every slot must lie inside the output / residual row
    if (a.out_mode != S3D_OUT_NHWC || a.gate || a.out_accumulate || a.cout_store != a.CoutPad || a.scale) return false;
    if (a.act != S3D_ACT_NONE && a.act != S3D_ACT_RELU) return false;
    return (long)a.N * a.H * a.W >= (1L << 17);   // long row sets (the decoder token rows); short ones keep lin_rows (more, smaller workgroups)
}
static int launch_lin_stream(const ConvLaunch& a, hipStream_t stream) {
    const long P = (long)a.N * a.H * a.W;
    const int kc = a.src[0].C / 128;
    const int rows_wg = kc > 1 ? 128 : 192;   // 4 waves x 2 or 3 row tiles
    const long n_tasks = (P + rows_wg - 1) / rows_wg;
    const unsigned grid = (unsigned)(n_tasks < 512 ? n_tasks : 512);   // two workgroups per CU
    if (kc == 1)
        hipLaunchKernelGGL(lin_stream_f16x3_kernel<1>, dim3(grid), dim3(256), 0, stream, a, n_tasks);
    else if (kc == 2)
        hipLaunchKernelGGL(lin_stream_f16x3_kernel<2>, dim3(grid), dim3(256), 0, stream, a, n_tasks);
    else
        hipLaunchKernelGGL(lin_stream_f16x3_kernel<3>, dim3(grid), dim3(256), 0, stream, a, n_tasks);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// LDS-staged 3x3 convolution (stride 1, "same"), split precision.
// A workgroup (4 waves) owns an 8-row x 16-column pixel tile of one image and CO_WG output channels.  Per
// 32-channel K chunk the (8+2) x (16+2) input halo is fetched ONCE (coalesced 16 B/lane), split into f16 hi/lo
// ONCE and parked in LDS as [pixel][32 halfs] with the four 16-byte quarters of a pixel XOR-swizzled by
// (pixel >> 1) & 3: ds_read_b128 is served in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... and with that
// swizzle the 16 lanes of every group fall on 16 distinct bank quads for ANY pixel offset of a tap (the earlier
// 80-byte padded stride did not: SQ_LDS_BANK_CONFLICT was 2.3x the LDS-active cycles); the staging writes cover whole
// 64-byte pixels either way.  The nine taps read their B fragments from LDS at shifted pixel offsets.  The direct kernel above loads and splits every input value 9 x (Cout / CO_WG) times instead.
// Zero padding = zeros written for halo pixels outside the image.  Waves are arranged WP x WC: wave (wp, wc) owns
// pixel rows {MT*wp .. +MT-1} of the tile and NT = 2 output-channel tiles; weight fragments come straight from
// the packed image (L2), requested one tap ahead.  The next chunk's halo is in flight under the MFMAs.
// ---------------------------------------------------------------------------------------------
#define C3_PXS 32            // halfs per pixel in LDS (swizzled quarters, no padding)
#define C3_SWZ(pix, q) (((q) ^ (((pix) >> 1) & 3)) * 8)   // half offset of 16-byte quarter q of a pixel
#define C3_HALO (10 * 18)    // pixels of the halo tile
// NBUF = 2: the next chunk's halo is parked while this one is consumed (45 KiB, three workgroups per CU).  NBUF = 1:
// one buffer (22.5 KiB, seven workgroups per CU) for layers whose K is one or two chunks — there a workgroup's life is a
// fetch, a split and 108-216 MFMAs per wave, nothing overlaps inside it, and what hides the fetch latency is the number
// of OTHER workgroups on the CU (the 32- and 64-channel layers at full resolution: 0.51 / 0.34 ms -> see DESIGN.md).
// -DC3_STAMPS (tools/r06_conv_stamps.sh; not the shipped library): wave 0 of every workgroup adds the 100 MHz ticks of its
// phases to device counters keyed by the launch's grid (gridDim.x 96 / 48 / 12 / other)
#ifdef C3_STAMPS
__device__ unsigned long long c3_stamps[4][8];
#define C3_STAMP(i)                                                                                              \
    do {                                                                                                         \
        const long long t__ = wall_clock64();                                                                    \
        if (threadIdx.x == 0) atomicAdd(&c3_stamps[c3_cls][i], (unsigned long long)(t__ - c3_last));              \
        c3_last = t__;                                                                                           \
    } while (0)
extern "C" void s3d_debug_c3_stamps(unsigned long long* out) {
    unsigned long long z[32] = {0};
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(c3_stamps), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c3_stamps), z, sizeof(z));
}
#else
#define C3_STAMP(i)
#endif
template <int MT, int WP, int WC, int NBUF, bool GN = false>
__global__ __launch_bounds__(256) void conv3x3_lds_f16x3_kernel(const ConvLaunch a, int tiles_x, int tiles_y,
                                                                int chunks_per_split) {
    static_assert(MT * WP == 8 && WP * WC == 4, "tile is 8 rows, 4 waves");
    constexpr int NT = 2, CO_WG = WC * NT * 16;
    __shared__ __attribute__((aligned(16))) _Float16 s_in[NBUF][2][C3_HALO * C3_PXS];   // [buf][hi|lo]
    // GN: per-channel affine table of this image's GroupNorm (+ FiLM), A | B
    __shared__ __attribute__((aligned(16))) float s_gn[GN ? 2 * S3D_GN_CMAX : 4];
#ifdef C3_STAMPS
    long long c3_last = wall_clock64();
    const int c3_cls = gridDim.x == 96 ? 0 : gridDim.x == 48 ? 1 : gridDim.x == 12 ? 2 : 3;
    if (threadIdx.x == 0) atomicAdd(&c3_stamps[c3_cls][7], 1ull);
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int wp = wave % WP, wc = wave / WP;
    const int n_co_blk = a.CoutPad / CO_WG;
    int blk_co;
    long blk_tile;
    conv_block_tile(a, n_co_blk, blk_co, blk_tile);
    int tile = (int)blk_tile;
    const int tx = tile % tiles_x;
    tile /= tiles_x;
    const int ty = tile % tiles_y, n = tile / tiles_y;
    const int x0 = tx * 16, y0 = ty * 8;
    const int KU32 = a.KU >> 1;
    const _Float16* wimg = reinterpret_cast<const _Float16*>(a.wpk16);
    const int jt0 = blk_co * (CO_WG / 16) + wc * NT;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zero4();

    // staging role: thread t handles halo slots t, t+256, ... ; slot = pixel * 8 + channel quad
    constexpr int NSLOT = C3_HALO * 8;              // 1440 float4 per chunk
    constexpr int NPRE = (NSLOT + 255) / 256;       // 6
    f32x4 pre[NPRE];
    unsigned pre_ok = 0u;     // GN: which of the prefetched slots lie inside the image (zero padding pads the NORMALISED input)
    int pre_cb = 0;           // GN: first channel of the prefetched chunk in the concatenated input
    auto fetch = [&](const ConvSrc& S, int c, int cbase) {
        const int ni = (S.bmod ? n % S.bmod : n) / S.bdiv;
        pre_ok = 0u;
        pre_cb = cbase + 32 * c;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int slot = threadIdx.x + 256 * i;
            const int pix = slot >> 3, q4 = slot & 7;
            const int hy = pix / 18, hx = pix - hy * 18;
            const int y = y0 + hy - 1, x = x0 + hx - 1;
            const bool ok = slot < NSLOT && y >= 0 && y < a.H && x >= 0 && x < a.W;
            const long off = ((long)(ni * a.H + (ok ? y : 0)) * a.W + (ok ? x : 0)) * S.C + 32 * c + 4 * q4;
            const f32x4 v = ld4(S.p + off);           // always a valid address; masked below
            pre[i] = ok ? v : zero4();
            if (GN && ok) pre_ok |= 1u << i;
        }
    };
    auto park = [&](int buf) {
        typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int slot = threadIdx.x + 256 * i;
            if (slot < NSLOT) {
                const int pix = slot >> 3, q4 = slot & 7;
                if (GN) {   // y = x * A[c] + B[c] (GroupNorm, gamma / beta, FiLM folded), SiLU; outside the image: 0
                    const f32x4 A = ld4(s_gn + pre_cb + 4 * q4), B = ld4(s_gn + S3D_GN_CMAX + pre_cb + 4 * q4);
                    f32x4 t = pre[i] * A + B;
                    if (a.gn.silu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            t[e] = t[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t[e]));
                    }
                    pre[i] = (pre_ok >> i) & 1u ? t : zero4();
                }
                half4_t hi, lo;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const _Float16 h = (_Float16)pre[i][t];
                    hi[t] = h;
                    lo[t] = (_Float16)(pre[i][t] - (float)h);
                }
                const int po = pix * C3_PXS + C3_SWZ(pix, q4 >> 1) + 4 * (q4 & 1);
                *reinterpret_cast<half4_t*>(&s_in[buf][0][po]) = hi;
                *reinterpret_cast<half4_t*>(&s_in[buf][1][po]) = lo;
            }
        }
    };

    // flat chunk sequence over the sources
    const int cu0 = a.src[0].C >> 5, cu1 = a.nsrc > 1 ? a.src[1].C >> 5 : 0;
    // split-K: blockIdx.y owns chunks [ch_lo, ch_hi) and writes raw partial sums (deep layers have few pixels)
    const int ch_lo = blockIdx.y * chunks_per_split;
    const int nchunk = min(cu0 + cu1, ch_lo + chunks_per_split);
    fetch(a.src[ch_lo < cu0 ? 0 : 1], ch_lo < cu0 ? ch_lo : ch_lo - cu0, ch_lo < cu0 ? 0 : a.src[0].C);
    if (GN) {   // this image's affine table -> LDS (the first chunk's halo loads are in flight meanwhile)
        const int Ct = a.src[0].C + (a.nsrc > 1 ? a.src[1].C : 0);
        const float* tb = a.gn.table + (size_t)n * 2 * Ct;
        for (int i = threadIdx.x; i < Ct / 4; i += 256) {
            st4(s_gn + 4 * i, ld4(tb + 4 * i));
            st4(s_gn + S3D_GN_CMAX + 4 * i, ld4(tb + Ct + 4 * i));
        }
        __syncthreads();
    }
    C3_STAMP(0);   // first fetch issued, table in LDS
    park(ch_lo & (NBUF - 1));
    C3_STAMP(1);   // first halo arrived and parked
    __syncthreads();
    C3_STAMP(2);   // barrier
#pragma unroll 1
    for (int ch = ch_lo; ch < nchunk; ++ch) {
        const int s = ch < cu0 ? 0 : 1, c = s ? ch - cu0 : ch, cu = s ? cu1 : cu0;
        const int ubase = s ? 9 * cu0 : 0;
        const _Float16* wp0 = wimg + ((size_t)jt0 * KU32 + ubase + c) * 1024 + lane * 8;   // + tap*cu*1024, + nt*KU32*1024
        // weight fragments of tap 0 are requested BEFORE the halo prefetch (vmcnt retires in order)
        chalf8 wh[NT], wl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const _Float16* f = wp0 + (size_t)nt * KU32 * 1024;
            wh[nt] = *reinterpret_cast<const chalf8*>(f);
            wl[nt] = *reinterpret_cast<const chalf8*>(f + 512);
        }
        if (ch + 1 < nchunk) {
            const int s1 = ch + 1 < cu0 ? 0 : 1;
            fetch(a.src[s1], s1 ? ch + 1 - cu0 : ch + 1, s1 ? a.src[0].C : 0);
        }
        const _Float16* sh = s_in[ch & (NBUF - 1)][0];
        const _Float16* sl = s_in[ch & (NBUF - 1)][1];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;   // halo-relative: output (r, m) reads halo (r + dy, m + dx)
            chalf8 nwh[NT], nwl[NT];
            if (tap < 8) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const _Float16* f = wp0 + ((size_t)nt * KU32 + (size_t)(tap + 1) * cu) * 1024;
                    nwh[nt] = *reinterpret_cast<const chalf8*>(f);
                    nwl[nt] = *reinterpret_cast<const chalf8*>(f + 512);
                }
            }
            chalf8 bh[MT], bl[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int pix = (MT * wp + mt + dy) * 18 + m + dx;
                const int po = pix * C3_PXS + C3_SWZ(pix, g);
                bh[mt] = *reinterpret_cast<const chalf8*>(sh + po);
                bl[mt] = *reinterpret_cast<const chalf8*>(sl + po);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (!a.single_pass) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], bl[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt], bh[mt], acc[mt][nt], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], bh[mt], acc[mt][nt], 0, 0, 0);
            }
            if (tap < 8) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    wh[nt] = nwh[nt];
                    wl[nt] = nwl[nt];
                }
            }
        }
        C3_STAMP(3);   // nine taps
        if (NBUF == 1) __syncthreads();   // everyone is done reading the single buffer before it is refilled
        if (ch + 1 < nchunk) park((ch + 1) & (NBUF - 1));
        C3_STAMP(4);   // next halo parked
        __syncthreads();
        C3_STAMP(5);   // barrier
    }
    int pn[MT], py[MT], px[MT];
    bool pv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        pn[mt] = n;
        py[mt] = y0 + MT * wp + mt;
        px[mt] = x0 + m;
        pv[mt] = py[mt] < a.H && px[mt] < a.W;
    }
    if (gridDim.y > 1) {   // partial[split][pixel][CoutPad]
        float* part = a.splitk_ws + (size_t)blockIdx.y * ((size_t)a.N * a.H * a.W * a.CoutPad);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if (pv[mt]) {
                float* o = part + ((size_t)(n * a.H + py[mt]) * a.W + px[mt]) * a.CoutPad + 4 * g;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) st4(o + (jt0 + nt) * 16, acc[mt][nt]);
            }
        C3_STAMP(6);   // partial stores issued
        return;
    }
    conv_epilogue<MT, NT>(a, acc, pn, py, px, pv, jt0, g);
}

// sums the split-K partials of 4 consecutive output channels of one pixel and applies the conv epilogue
__global__ void conv_splitk_finish_kernel(const ConvLaunch a, int nsplit) {
    const long P = (long)a.N * a.H * a.W;
    const int cq = a.CoutPad >> 2;
    const long total = P * cq;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long p = idx / cq;
        const int co = (int)(idx - p * cq) * 4;
        f32x4 s = zero4();
        const float* src = a.splitk_ws + (size_t)p * a.CoutPad + co;
        const size_t stride = (size_t)P * a.CoutPad;
        int k = 0;
        for (; k + 4 <= nsplit; k += 4) {   // four partials in flight (the kernel is a few microseconds of load latency)
            const f32x4 v0 = ld4(src + (size_t)k * stride), v1 = ld4(src + (size_t)(k + 1) * stride),
                        v2 = ld4(src + (size_t)(k + 2) * stride), v3 = ld4(src + (size_t)(k + 3) * stride);
            __builtin_amdgcn_sched_barrier(0);
            s += (v0 + v1) + (v2 + v3);
        }
        for (; k < nsplit; ++k) s += ld4(src + (size_t)k * stride);
        f32x4 acc[1][1] = {{s}};
        const int n = (int)(p / ((long)a.H * a.W));
        const int r = (int)(p - (long)n * a.H * a.W);
        const int pn[1] = {n}, py[1] = {r / a.W}, px[1] = {r % a.W};
        const bool pv[1] = {true};
        conv_epilogue<1, 1>(a, acc, pn, py, px, pv, co >> 4, (co >> 2) & 3);
    }
}

int launch_conv_splitk_finish(const ConvLaunch& a, int nsplit, hipStream_t stream) {
    const long total = (long)(((size_t)a.N * a.H * a.W * a.CoutPad) >> 2);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3(blocks), dim3(256), 0, stream, a, nsplit);
    S3D_LAUNCH_CHECK();
    return 0;
}


// ---------------------------------------------------------------------------------------------
// The same convolution for maps of a few pixels (round 6): N * H * W <= 64 output pixels in all (dispatched up to 32: the
// 4 x 4 level of the latent-diffusion U-Net at batch 1 and 2 — 21 of its 70 3x3 convolutions per step; see the launch).  The tiled
// kernel above gives such a map an 8 x 16 pixel tile per image: 87 % (4 x 4) or 50 % (8 x 8) of its MFMAs multiply padding,
// two of its four waves own rows that do not exist, and its weights arrive one tap at a time.  Here a pixel tile is 16
// consecutive output pixels of the flattened (n, y, x) index (PT = 1, 2 or 4 of them), every wave owns ONE 16-channel
// output tile for all pixel tiles (the accumulators are 4 PT registers), and that leaves room to request all nine taps'
// weight fragments of a chunk at once (72 registers) and the next chunk's under the current one's products.  The zero-padded
// input of ALL images ((H + 2) x (W + 2) pixels each) is parked per 32-channel chunk as in the tiled kernel (same swizzle).
// Split-K only (the launch falls back to the tiled kernel otherwise): partial[split][pixel][CoutPad], same chunk ranges, same
// order of products per accumulator — bit-identical to the tiled kernel at the same split count.
// ---------------------------------------------------------------------------------------------
#define C3S_HP 160           // padded pixels of all images a workgroup parks: 8 x 8 -> 100, four 4 x 4 images -> 144
#define C3S_HP1 64           // ... of the one-pixel-tile form (a 4 x 4 map: 36)
// q / d for q < 409, 1 <= d <= 160 as a 16-bit fixed-point multiply with magic = 65536 / d + 1: magic d = 65536 + e, 1 <= e <= d, and
// the quotient is exact while q e < 65536 (the dispatch keeps every divisor here at or below 160 and every dividend below 256)
__device__ __forceinline__ int c3s_div(int q, int magic) { return (q * magic) >> 16; }
// TAPS = 1: the 1x1 convolutions of the same levels (qkv / proj_out of the attention blocks, the ResBlocks' skip convolutions):
// no zero border, one weight fragment pair per chunk — what it buys them is the split-K (the implicit-GEMM kernel they ran on
// has none: 768 -> 2 304 at a 4 x 4 map was 24 workgroups pulling 7 MB of weights, 21 us).
template <int PT, bool GN, int TAPS = 9>
__global__ __launch_bounds__(256) void conv3x3_small_f16x3_kernel(const ConvLaunch a, int chunks_per_split) {
    constexpr int BORD = TAPS == 9 ? 1 : 0;
    __shared__ __attribute__((aligned(16))) _Float16 s_in[2][2][C3S_HP * C3_PXS];   // [buf][hi|lo]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int Wp = a.W + 2 * BORD, HpWp = (a.H + 2 * BORD) * Wp, HP = a.N * HpWp, HW = a.H * a.W, P = a.N * HW;
    const int KU32 = a.KU >> 1;
    const _Float16* wimg = reinterpret_cast<const _Float16*>(a.wpk16);
    const int jt = blockIdx.x * 4 + wave;
    const int mg_hpwp = 65536 / HpWp + 1, mg_wp = 65536 / Wp + 1, mg_hw = 65536 / HW + 1, mg_w = 65536 / a.W + 1;   // uniform
    f32x4 acc[PT];
    int pix0[PT], pout[PT];   // padded pixel read by tap (0, 0); output pixel index (or -1)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        acc[pt] = zero4();
        const int p = 16 * pt + m;
        const int pc = p < P ? p : 0;
        const int n = c3s_div(pc, mg_hw), r = pc - n * HW, y = c3s_div(r, mg_w), x = r - y * a.W;
        pix0[pt] = n * HpWp + y * Wp + x;   // tap (dy, dx) reads padded pixel pix0 + dy * Wp + dx
        pout[pt] = p < P ? p : -1;
    }
    constexpr int NPRE = ((PT == 1 ? C3S_HP1 : C3S_HP) * 8 + 255) / 256;   // float4 per thread and chunk: 2 (one pixel tile) or 5
    f32x4 pre[NPRE];
    unsigned pre_ok = 0u;
    int pre_cb = 0;
    const int Ct = a.src[0].C + (a.nsrc > 1 ? a.src[1].C : 0);
    auto fetch = [&](const ConvSrc& S, int c, int cbase) {
        pre_ok = 0u;
        pre_cb = cbase + 32 * c;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int slot = threadIdx.x + 256 * i;
            const int pix = slot >> 3, q4 = slot & 7;
            const int n = c3s_div(pix, mg_hpwp), r = pix - n * HpWp, hy = c3s_div(r, mg_wp), hx = r - hy * Wp;
            const int y = hy - BORD, x = hx - BORD;
            const bool ok = pix < HP && y >= 0 && y < a.H && x >= 0 && x < a.W;
            const int ni = ok ? (S.bmod ? n % S.bmod : n) / S.bdiv : 0;
            const long off = ((long)(ni * a.H + (ok ? y : 0)) * a.W + (ok ? x : 0)) * S.C + 32 * c + 4 * q4;
            const f32x4 v = ld4(S.p + off);
            pre[i] = ok ? v : zero4();
            if (ok) pre_ok |= 1u << i;
        }
    };
    // GN: the affine table entries of the prefetched slots, straight from L2 (y = x A + B).  Requested BEFORE a chunk's
    // weights: the vector memory queue returns in order, and park() must not wait behind 18 KB of weight fragments
    f32x4 gnA[GN ? NPRE : 1], gnB[GN ? NPRE : 1];
    auto gn_load = [&]() {
        if (!GN) return;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int slot = threadIdx.x + 256 * i;
            const int pix = slot >> 3, q4 = slot & 7;
            const int n = pix < HP ? c3s_div(pix, mg_hpwp) : 0;
            const float* tb = a.gn.table + (size_t)n * 2 * Ct + pre_cb + 4 * q4;
            gnA[i] = ld4(tb);
            gnB[i] = ld4(tb + Ct);
        }
    };
    auto park = [&](int buf) {
        typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int slot = threadIdx.x + 256 * i;
            const int pix = slot >> 3, q4 = slot & 7;
            if (pix < HP) {
                if (GN) {
                    const f32x4 A = gnA[i], B = gnB[i];
                    f32x4 t = pre[i] * A + B;
                    if (a.gn.silu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            t[e] = t[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t[e]));
                    }
                    pre[i] = (pre_ok >> i) & 1u ? t : zero4();
                }
                half4_t hi, lo;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const _Float16 h = (_Float16)pre[i][t];
                    hi[t] = h;
                    lo[t] = (_Float16)(pre[i][t] - (float)h);
                }
                const int po = pix * C3_PXS + C3_SWZ(pix, q4 >> 1) + 4 * (q4 & 1);
                *reinterpret_cast<half4_t*>(&s_in[buf][0][po]) = hi;
                *reinterpret_cast<half4_t*>(&s_in[buf][1][po]) = lo;
            }
        }
    };
    const int cu0 = a.src[0].C >> 5, cu1 = a.nsrc > 1 ? a.src[1].C >> 5 : 0;
    const int ch_lo = blockIdx.y * chunks_per_split;
    const int nchunk = min(cu0 + cu1, ch_lo + chunks_per_split);
    constexpr bool DBW = PT == 1;   // the next chunk's weights under this chunk's products (144 registers); PT >= 2: after them
    chalf8 wh[TAPS], wl[TAPS], nwh[DBW ? TAPS : 1], nwl[DBW ? TAPS : 1];
    auto request = [&](int ch, chalf8 (&h)[TAPS], chalf8 (&l)[TAPS]) {
        const int s = ch < cu0 ? 0 : 1, c = s ? ch - cu0 : ch, cu = s ? cu1 : cu0;
        const _Float16* f0 = wimg + ((size_t)jt * KU32 + (s ? TAPS * cu0 : 0) + c) * 1024 + lane * 8;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            h[tap] = *reinterpret_cast<const chalf8*>(f0 + (size_t)tap * cu * 1024);
            l[tap] = *reinterpret_cast<const chalf8*>(f0 + (size_t)tap * cu * 1024 + 512);
        }
    };
    fetch(a.src[ch_lo < cu0 ? 0 : 1], ch_lo < cu0 ? ch_lo : ch_lo - cu0, ch_lo < cu0 ? 0 : a.src[0].C);
    gn_load();
    request(ch_lo, wh, wl);   // behind the input loads in the (in-order) queue: park() waits for those only
    park(ch_lo & 1);
    __syncthreads();
#pragma unroll 1
    for (int ch = ch_lo; ch < nchunk; ++ch) {
        if (ch + 1 < nchunk) {
            const int s1 = ch + 1 < cu0 ? 0 : 1;
            fetch(a.src[s1], s1 ? ch + 1 - cu0 : ch + 1, s1 ? a.src[0].C : 0);
            if constexpr (DBW) request(ch + 1, nwh, nwl);
        }
        const _Float16* sh = s_in[ch & 1][0];
        const _Float16* sl = s_in[ch & 1][1];
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int toff = TAPS == 9 ? (tap / 3) * Wp + tap % 3 : 0;
            chalf8 bh[PT], bl[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int pix = pix0[pt] + toff;
                const int po = pix * C3_PXS + C3_SWZ(pix, g);
                bh[pt] = *reinterpret_cast<const chalf8*>(sh + po);
                bl[pt] = *reinterpret_cast<const chalf8*>(sl + po);
            }
            if (!a.single_pass) {
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[tap], bl[pt], acc[pt], 0, 0, 0);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[tap], bh[pt], acc[pt], 0, 0, 0);
            }
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[tap], bh[pt], acc[pt], 0, 0, 0);
        }
        if (ch + 1 < nchunk) {
            if constexpr (DBW) {
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    wh[tap] = nwh[tap];
                    wl[tap] = nwl[tap];
                }
                gn_load();
            } else {
                gn_load();
                request(ch + 1, wh, wl);
            }
            park((ch + 1) & 1);
        }
        __syncthreads();
    }
    float* part = a.splitk_ws + (size_t)blockIdx.y * ((size_t)P * a.CoutPad);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
        if (pout[pt] >= 0) st4(part + (size_t)pout[pt] * a.CoutPad + jt * 16 + 4 * g, acc[pt]);
}

static bool conv3x3_lds_eligible(const ConvLaunch& a) {
    if (a.ks != 3 || a.stride > 1 || (a.Hin && a.Hin != a.H) || (a.Win && a.Win != a.W)) return false;
    if (!a.wpk16 || a.KU % 2 || a.CoutPad % 32) return false;
    // maps smaller than the 8 x 16 tile waste lanes; they are worth it only when split-K supplies the parallelism
    if ((a.W < 16 || a.H < 8) && !(a.splitk_ws && a.out_mode == S3D_OUT_NHWC)) return false;
    for (int s = 0; s < a.nsrc; ++s)
        if (a.src[s].C % 32 || a.src[s].sbcast) return false;
    return true;
}
// Split count of a split-K launch of `nblk_k` workgroups per split over `nchunk` 32-channel chunks: the chip holds 512 of these
// workgroups at a time (two per CU); a launch runs in ceil(workgroups / 512) rounds of (chunks per split + a fixed fetch /
// epilogue share) each; the finish pass reads `se` partials.  Pick the count with the least rounds x length.
static int conv_choose_splits(long nblk_k, int nchunk, size_t out_floats, size_t splitk_floats) {
    int splits = 1;
    long best = -1;
    for (int sp = 1; sp <= nchunk && (size_t)sp * out_floats <= splitk_floats; ++sp) {
        const int c = (nchunk + sp - 1) / sp, se = (nchunk + c - 1) / c;
        if (se != sp) continue;   // (the same chunking as a smaller count)
        const long rounds = (nblk_k * se + 511) / 512;
        const long cost = rounds * (2 * c + 3) + (se > 1 ? 1 + se / 8 : 0);   // half-chunk units
        if (best < 0 || cost < best) {
            best = cost;
            splits = se;
        }
    }
    return splits;
}
static const bool g_conv_small_on = !(getenv("S3D_CONV_SMALL") && atoi(getenv("S3D_CONV_SMALL")) == 0);
static const long g_conv_small_pmax = getenv("S3D_CONV_SMALL") && atoi(getenv("S3D_CONV_SMALL")) > 1 ? atoi(getenv("S3D_CONV_SMALL")) : 32;

// 1x1 convolutions of maps of a few pixels on conv3x3_small_f16x3_kernel<PT, false, 1> (split-K); false: not served
static const bool g_conv_small1_on = !(getenv("S3D_CONV_SMALL_1X1") && atoi(getenv("S3D_CONV_SMALL_1X1")) == 0);
// (up to 64 pixels — four pixel tiles: step 4.00-4.08 -> 3.96 ms with 32, 3.88 with 64; S3D_CONV_SMALL_1X1=0 / 32 for the A/B)
static const long g_conv_small1_pmax = getenv("S3D_CONV_SMALL_1X1") && atoi(getenv("S3D_CONV_SMALL_1X1")) > 1 ? atoi(getenv("S3D_CONV_SMALL_1X1")) : 64;
static bool conv1x1_small_eligible(const ConvLaunch& a) {
    if (!g_conv_small1_on || a.ks != 1 || a.stride > 1 || a.Hin || a.Win || !a.wpk16 || a.KU % 2 || a.CoutPad % 64) return false;
    if (!a.splitk_ws || a.out_mode != S3D_OUT_NHWC || a.gn.table) return false;
    const long P = (long)a.N * a.H * a.W;
    if (P > (g_conv_small1_pmax < 64 ? g_conv_small1_pmax : 64) || P > C3S_HP) return false;
    for (int s = 0; s < a.nsrc; ++s)
        if (a.src[s].C % 32 || a.src[s].sbcast) return false;
    return true;
}
static int launch_conv1x1_small(const ConvLaunch& a, hipStream_t stream, bool& served) {
    int nchunk = 0;
    for (int s = 0; s < a.nsrc; ++s) nchunk += a.src[s].C >> 5;
    const long P = (long)a.N * a.H * a.W;
    int splits = conv_choose_splits(a.CoutPad / 64, nchunk, (size_t)P * a.CoutPad, a.splitk_floats);
    const int cps = (nchunk + splits - 1) / splits;
    splits = (nchunk + cps - 1) / cps;
    served = splits > 1;
    if (!served) return 0;
    dim3 grid((unsigned)(a.CoutPad / 64), (unsigned)splits);
    const int pt = (int)((P + 15) / 16);
    if (pt == 1 && P <= C3S_HP1) hipLaunchKernelGGL((conv3x3_small_f16x3_kernel<1, false, 1>), grid, dim3(256), 0, stream, a, cps);
    else if (pt <= 2) hipLaunchKernelGGL((conv3x3_small_f16x3_kernel<2, false, 1>), grid, dim3(256), 0, stream, a, cps);
    else hipLaunchKernelGGL((conv3x3_small_f16x3_kernel<4, false, 1>), grid, dim3(256), 0, stream, a, cps);
    S3D_LAUNCH_CHECK();
    if (a.splits_out) {
        *a.splits_out = splits;
        return 0;
    }
    return launch_conv_splitk_finish(a, splits, stream);
}

static int launch_conv3x3_lds(const ConvLaunch& a, hipStream_t stream) {
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 7) / 8;
    const long tiles = (long)tiles_x * tiles_y * a.N;
    const int co_wg = a.CoutPad % 64 == 0 ? 64 : 32;
    const long nblk = tiles * (a.CoutPad / co_wg);
    S3D_CHECK_ARG(nblk < (1L << 31), "conv grid out of range (%ld)", nblk);
    int nchunk = 0;
    for (int s = 0; s < a.nsrc; ++s) nchunk += a.src[s].C >> 5;
    // split-K when the pixel x channel tiling cannot fill the chip (deep encoder layers at batch 1)
    int splits = 1;
    const size_t out_floats = (size_t)a.N * a.H * a.W * a.CoutPad;
    // maps of a few pixels (all images together at most 64 pixels, 160 with their zero borders): conv3x3_small_f16x3_kernel
    const long P_all = (long)a.N * a.H * a.W;
    // (P_all <= 32: at the 8 x 8 maps — four pixel tiles — this kernel measured 16.7-17.5 us per call against the tiled kernel's
    //  16.1, in the compact form, in a conflict-free form over the zero-bordered index space and with the table loads ahead of
    //  the weights; at the 4 x 4 maps 11.2 against 16.0: profiles/r06_ldm_small_maps.md.  S3D_CONV_SMALL=64 forces it there.)
    const bool small = g_conv_small_on && a.splitk_ws && a.out_mode == S3D_OUT_NHWC && co_wg == 64 &&
                       (P_all <= (g_conv_small_pmax < 64 ? g_conv_small_pmax : 64) || ((long)a.H * a.W <= 16 && P_all <= 64)) &&   // (four 4 x 4 images: 7.08 -> 7.01 ms per batch-4 step)
                       (long)a.N * (a.H + 2) * (a.W + 2) <= C3S_HP;
    const long nblk_k = small ? a.CoutPad / 64 : nblk;   // workgroups per split
    if (a.splitk_ws && a.out_mode == S3D_OUT_NHWC && nblk_k < 512) splits = conv_choose_splits(nblk_k, nchunk, out_floats, a.splitk_floats);
    // (doubling until 512 workgroups was the rule before round 4: it landed on 640 — two rounds, the second a quarter full — for
    //  every 64 x 64 x 320 layer of the LDM U-Net)
    const int cps = (nchunk + splits - 1) / splits;
    splits = (nchunk + cps - 1) / cps;
    if (small && splits > 1) {
        dim3 sgrid((unsigned)(a.CoutPad / 64), (unsigned)splits);
        const int pt = (int)((P_all + 15) / 16);
        const long hp_all = (long)a.N * (a.H + 2) * (a.W + 2);
#define C3S_LAUNCH(PT_)                                                                                                  \
    if (a.gn.table) hipLaunchKernelGGL((conv3x3_small_f16x3_kernel<PT_, true>), sgrid, dim3(256), 0, stream, a, cps);      \
    else hipLaunchKernelGGL((conv3x3_small_f16x3_kernel<PT_, false>), sgrid, dim3(256), 0, stream, a, cps)
        if (pt == 1 && hp_all <= C3S_HP1) { C3S_LAUNCH(1); }
        else if (pt <= 2) { C3S_LAUNCH(2); }
        else { C3S_LAUNCH(4); }
#undef C3S_LAUNCH
        S3D_LAUNCH_CHECK();
        if (a.splits_out) {
            *a.splits_out = splits;
            return 0;
        }
        return launch_conv_splitk_finish(a, splits, stream);
    }
    dim3 grid((unsigned)nblk, (unsigned)splits);
    constexpr int one_buf_max = 2;   // chunks per workgroup up to which the single-buffer variant runs
    const bool one = cps <= one_buf_max;   // 32-channel-output layers only: measured -13 % there, +5 % on the 64-wide tile
    if (a.gn.table) {   // GroupNorm of the input in the staging path (LDM U-Net): its own instantiations (12 KiB of table)
        int ct = 0;
        for (int s = 0; s < a.nsrc; ++s) ct += a.src[s].C;
        S3D_CHECK_ARG(ct <= S3D_GN_CMAX, "conv: fused GroupNorm over %d channels", ct);
        if (co_wg == 64)
            hipLaunchKernelGGL((conv3x3_lds_f16x3_kernel<4, 2, 2, 2, true>), grid, dim3(256), 0, stream, a, tiles_x, tiles_y, cps);
        else
            hipLaunchKernelGGL((conv3x3_lds_f16x3_kernel<2, 4, 1, 2, true>), grid, dim3(256), 0, stream, a, tiles_x, tiles_y, cps);
    } else if (co_wg == 64) {
        hipLaunchKernelGGL((conv3x3_lds_f16x3_kernel<4, 2, 2, 2>), grid, dim3(256), 0, stream, a, tiles_x, tiles_y, cps);
    } else {
        if (one) hipLaunchKernelGGL((conv3x3_lds_f16x3_kernel<2, 4, 1, 1>), grid, dim3(256), 0, stream, a, tiles_x, tiles_y, cps);
        else hipLaunchKernelGGL((conv3x3_lds_f16x3_kernel<2, 4, 1, 2>), grid, dim3(256), 0, stream, a, tiles_x, tiles_y, cps);
    }
    S3D_LAUNCH_CHECK();
    if (a.splits_out) {
        *a.splits_out = splits;
        if (splits > 1) return 0;   // the consumer sums the partials (and applies bias / residual) itself
    }
    if (splits > 1) return launch_conv_splitk_finish(a, splits, stream);
    return 0;
}

template <int MT, int NT, int WM, int WN>
static int launch_cfg(const ConvLaunch& a, hipStream_t stream) {
    constexpr int PIX_WG = WM * MT * 16, CO_WG = WN * NT * 16;
    const long P = (long)a.N * a.H * a.W;
    const long nblk = ((P + PIX_WG - 1) / PIX_WG) * (a.CoutPad / CO_WG);
    S3D_CHECK_ARG(nblk > 0 && nblk < (1L << 31), "conv grid out of range (%ld)", nblk);
    dim3 grid((unsigned)nblk), block(WM * WN * 64);
    bool f16 = a.wpk16 != nullptr && (a.KU % 2 == 0);
    for (int s = 0; s < a.nsrc; ++s) f16 = f16 && (a.src[s].C % 32 == 0);
    if (f16) {
        if (a.ks == 3)
            hipLaunchKernelGGL((conv_igemm_f16x3_kernel<MT, NT, WM, WN, 3>), grid, block, 0, stream, a);
        else if (a.ks == 2)
            hipLaunchKernelGGL((conv_igemm_f16x3_kernel<MT, NT, WM, WN, 2>), grid, block, 0, stream, a);
        else
            hipLaunchKernelGGL((conv_igemm_f16x3_kernel<MT, NT, WM, WN, 1>), grid, block, 0, stream, a);
        S3D_LAUNCH_CHECK();
        return 0;
    }
    if (a.ks == 3)
        hipLaunchKernelGGL((conv_igemm_kernel<MT, NT, WM, WN, 3>), grid, block, 0, stream, a);
    else if (a.ks == 2)
        hipLaunchKernelGGL((conv_igemm_kernel<MT, NT, WM, WN, 2>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<MT, NT, WM, WN, 1>), grid, block, 0, stream, a);
    S3D_LAUNCH_CHECK();
    return 0;
}

int launch_conv(const ConvLaunch& a_in, hipStream_t stream) {
    ConvLaunch a = a_in;
    a.xcd_remap = 1;   // block -> (pixel tile, cout tile) mapping keeps the cout tiles of a pixel tile on one XCD
    S3D_CHECK_ARG(a.ks >= 1 && a.ks <= 3, "conv: ks must be 1, 2 or 3");
    S3D_CHECK_ARG(a.CoutPad % 16 == 0 && a.CoutPad > 0, "conv: CoutPad %d", a.CoutPad);
    for (int s = 0; s < a.nsrc; ++s)
        S3D_CHECK_ARG(a.src[s].C % 16 == 0 && a.src[s].bdiv >= 1, "conv: bad source %d", s);
    // ConvTranspose output: the quadrant-scatter store carries affine + activation only.  Its full-line form (32-channel
    // pairs, conv_epilogue) pairs accumulator tiles (jt0 + 2 np, jt0 + 2 np + 1) with jt0 = a multiple of NT — even for
    // every tile shape of the menu below — and would index dropout / gate / residual by the SCATTERED offset.
    S3D_CHECK_ARG(a.out_mode != S3D_OUT_CONVT || (a.drop.p <= 0.f && !a.gate && !a.residual && !a.out_accumulate),
                  "conv: ConvTranspose output takes no dropout / gate / residual / accumulate");
    if (conv3x3_lds_eligible(a)) return launch_conv3x3_lds(a, stream);
    S3D_CHECK_ARG(!a.gn.table, "conv: a fused GroupNorm needs the LDS-staged 3x3 kernel (ks 3, stride 1, channel counts multiples of 32)");
    if (conv1x1_small_eligible(a)) {
        bool served = false;
        const int rc = launch_conv1x1_small(a, stream, served);
        if (rc || served) return rc;
    }
    if (lin_stream_eligible(a)) return launch_lin_stream(a, stream);
    if (lin_rows_eligible(a)) return launch_lin_rows(a, stream);
    const long P = (long)a.N * a.H * a.W;
    // Tile menu: (pixels x couts) per workgroup of 4 waves.  Prefer the largest tile that still
    // yields >= ~2 workgroups per CU; small late-encoder maps fall through to the small tiles.
    const long want = 512;
    auto nblk = [&](int pix, int co) { return ((P + pix - 1) / pix) * (a.CoutPad / co); };
    if (a.CoutPad % 128 == 0) {
        if (nblk(128, 128) >= want) return launch_cfg<4, 4, 2, 2>(a, stream);
        if (nblk(64, 128) >= want) return launch_cfg<2, 4, 2, 2>(a, stream);
        if (a.CoutPad % 128 == 0 && nblk(32, 128) >= want / 2) return launch_cfg<1, 4, 2, 2>(a, stream);
        return launch_cfg<1, 1, 2, 2>(a, stream);  // 32 px x 32 co
    }
    if (a.CoutPad % 64 == 0) {
        if (nblk(256, 64) >= want) return launch_cfg<4, 4, 4, 1>(a, stream);
        if (nblk(64, 64) >= want / 2) return launch_cfg<2, 2, 2, 2>(a, stream);
        return launch_cfg<1, 1, 2, 2>(a, stream);
    }
    if (a.CoutPad % 32 == 0) {
        if (nblk(256, 32) >= want) return launch_cfg<4, 2, 4, 1>(a, stream);
        return launch_cfg<1, 1, 2, 2>(a, stream);
    }
    if (nblk(256, 16) >= want) return launch_cfg<4, 1, 4, 1>(a, stream);
    return launch_cfg<1, 1, 4, 1>(a, stream);
}

// ---------------------------------------------------------------------------------------------
// packers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_frag_body(const PackArgs& a) {
    const long total = (long)(a.n_pad / 16) * a.ku_seg * 256;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        int e, lane, ul, j, k;
        if (!a.f16) {
            e = idx & 3; lane = (idx >> 2) & 63;
            const long fu = idx >> 8;
            ul = (int)(fu % a.ku_seg); j = (int)(fu / a.ku_seg);
            k = 16 * ul + 4 * (lane >> 4) + e;
        } else {   // 32-deep chunks: ku_seg is even; lane (r,g) holds k = 32*u32 + 8g + t, t = 0..7
            const int t = idx & 7;
            lane = (idx >> 3) & 63;
            const long fu = idx >> 9;
            const int u32 = (int)(fu % (a.ku_seg / 2));
            j = (int)(fu / (a.ku_seg / 2));
            k = 32 * u32 + 8 * (lane >> 4) + t;
            ul = u32; e = t;
        }
        const int n = 16 * j + (lane & 15);
        float v = 0.f;
        if (n < a.n_valid) {
            if (a.kind == S3D_PACK_LINEAR) {
                if (k < a.k_valid) v = a.src[(long)n * a.ld + k];
            } else if (a.kind == S3D_PACK_CONV) {
                const int tap = k / a.cseg, c = k - tap * a.cseg;
                if (c < a.cseg_valid) v = a.src[((long)n * a.cin_tot + a.cin_begin + c) * a.taps + tap];
            } else if (a.kind == S3D_PACK_CONVT) {
                const int q = n / a.ct, co = n - q * a.ct;
                if (k < a.k_valid) v = a.src[((long)k * a.ct + co) * 4 + q];
            } else if (a.kind == S3D_PACK_CONV_DGRAD) {
                const int tap = k / a.cseg, co = k - tap * a.cseg;
                if (co < a.cseg_valid)
                    v = a.src[((long)co * a.cin_tot + a.cin_begin + n) * a.taps + (a.taps - 1 - tap)];
            } else if (a.kind == S3D_PACK_CONVT_DGRAD) {
                const int q = k / a.ct, co = k - q * a.ct;
                if (q < 4) v = a.src[((long)n * a.ct + co) * 4 + q];
            } else {  // LINEAR_T
                if (k < a.k_valid) v = a.src[(long)k * a.ld + n];
            }
        }
        if (a.f16) {
            _Float16* d = reinterpret_cast<_Float16*>(a.dst) +
                          ((long)j * (a.KU_total / 2) + a.u_off / 2 + ul) * 1024 + lane * 8 + e;
            const _Float16 h = (_Float16)v;
            d[0] = h;
            d[512] = (_Float16)(v - (float)h);
            continue;
        }
        long fi;
        if (a.chunk_ku > 0)
            fi = ((long)(ul / a.chunk_ku) * (a.n_pad / 16) + j) * a.chunk_ku + ul % a.chunk_ku;
        else
            fi = (long)j * a.KU_total + a.u_off + ul;
        a.dst[(fi * 64 + lane) * 4 + e] = v;
    }
}

__global__ void pack_frag_kernel(const PackArgs a) { pack_frag_body(a); }

// many packs in one launch (a training step repacks every weight matrix: 145 - 273 of them); blockIdx.y = entry
#define PACK_TABLE_MAX 48
struct PackTable {
    PackArgs e[PACK_TABLE_MAX];
};
__global__ void pack_frag_table_kernel(const PackTable t) { pack_frag_body(t.e[blockIdx.y]); }

static thread_local std::vector<PackArgs>* t_pack_batch = nullptr;
PackBatchScope::PackBatchScope() : prev_(t_pack_batch), q_(new std::vector<PackArgs>()) {
    t_pack_batch = (std::vector<PackArgs>*)q_;
}
PackBatchScope::~PackBatchScope() {
    if (t_pack_batch == (std::vector<PackArgs>*)q_) t_pack_batch = (std::vector<PackArgs>*)prev_;
    delete (std::vector<PackArgs>*)q_;
}
PackBatchSuspend::PackBatchSuspend() : saved_(t_pack_batch) { t_pack_batch = nullptr; }
PackBatchSuspend::~PackBatchSuspend() { t_pack_batch = (std::vector<PackArgs>*)saved_; }
int PackBatchScope::flush(hipStream_t stream) {
    std::vector<PackArgs>& q = *(std::vector<PackArgs>*)q_;
    for (size_t i0 = 0; i0 < q.size(); i0 += PACK_TABLE_MAX) {
        PackTable t;
        const int cnt = (int)(q.size() - i0 < PACK_TABLE_MAX ? q.size() - i0 : PACK_TABLE_MAX);
        long mx = 0;
        for (int i = 0; i < cnt; ++i) {
            t.e[i] = q[i0 + i];
            const long total = (long)(t.e[i].n_pad / 16) * t.e[i].ku_seg * 256;
            if (total > mx) mx = total;
        }
        const int bx = (int)((mx + 255) / 256 < 1024 ? (mx + 255) / 256 : 1024);
        hipLaunchKernelGGL(pack_frag_table_kernel, dim3(bx, cnt), dim3(256), 0, stream, t);
        S3D_LAUNCH_CHECK();
    }
    q.clear();
    t_pack_batch = (std::vector<PackArgs>*)prev_;   // later packs launch directly again
    return 0;
}

int launch_pack(const PackArgs& a, hipStream_t stream) {
    S3D_CHECK_ARG(a.n_pad % 16 == 0 && a.ku_seg > 0, "pack: bad dims");
    if (t_pack_batch) {   // inside a PackBatchScope: queued until its flush()
        t_pack_batch->push_back(a);
        return 0;
    }
    const long total = (long)(a.n_pad / 16) * a.ku_seg * 256;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(blocks), dim3(256), 0, stream, a);
    S3D_LAUNCH_CHECK();
    return 0;
}

__global__ void fold_bn_kernel(const float* bias, const float* g, const float* b, const float* mu,
                               const float* var, float* scale, float* shift, int c_valid, int c_pad, int rep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c_pad) return;
    float sc = 1.f, sh = 0.f;
    if (i < c_valid * rep) {
        const int c = i % c_valid;
        const float bv = bias ? bias[c] : 0.f;
        if (g) {
            sc = g[c] / sqrtf(var[c] + 1e-5f);
            sh = (bv - mu[c]) * sc + b[c];
        } else {
            sh = bv;
        }
    }
    scale[i] = sc;
    shift[i] = sh;
}

int launch_fold_bn(const float* bias, const float* const bn[4], float* scale, float* shift, int c_valid,
                   int c_pad, int rep, int, hipStream_t stream) {
    const float* g = bn ? bn[0] : nullptr;
    hipLaunchKernelGGL(fold_bn_kernel, dim3((c_pad + 127) / 128), dim3(128), 0, stream, bias, g,
                       g ? bn[1] : nullptr, g ? bn[2] : nullptr, g ? bn[3] : nullptr, scale, shift, c_valid,
                       c_pad, rep);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// BN(eval) + ReLU + MaxPool2d(2)  on NHWC  (the block that opens down2..down5, unet_custom.py:13-19)
// ---------------------------------------------------------------------------------------------
__global__ void bn_relu_pool_kernel(const float* __restrict__ in, const float* __restrict__ scale,
                                    const float* __restrict__ shift, float* __restrict__ out, int n, int h,
                                    int w, int c) {
    const int c4 = c >> 2, ho = h >> 1, wo = w >> 1;
    const long total = (long)n * ho * wo * c4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % c4) * 4;
        long r = idx / c4;
        const int x = (int)(r % wo);
        r /= wo;
        const int y = (int)(r % ho);
        const int ni = (int)(r / ho);
        const f32x4 sc = scale ? ld4(scale + cc) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 sh = shift ? ld4(shift + cc) : zero4();
        f32x4 best;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float* p = in + ((long)(ni * h + 2 * y + (t >> 1)) * w + 2 * x + (t & 1)) * c + cc;
            f32x4 v = ld4(p) * sc + sh;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = fmaxf(v[i], 0.f);
                best[i] = t == 0 ? v[i] : fmaxf(best[i], v[i]);
            }
        }
        st4(out + idx * 4, best);
    }
}

int launch_bn_relu_pool(const float* in, const float* scale, const float* shift, float* out, int n, int h,
                        int w, int c, hipStream_t stream) {
    S3D_CHECK_ARG(c % 4 == 0 && h % 2 == 0 && w % 2 == 0, "pool: bad dims");
    const long total = (long)n * (h / 2) * (w / 2) * (c / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(bn_relu_pool_kernel, dim3(blocks), dim3(256), 0, stream, in, scale, shift, out, n, h, w,
                       c);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// layout helpers
// ---------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int c,
                                    int h, int w, int cpad) {
    const long total = (long)n * h * w * cpad;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % cpad);
        long r = idx / cpad;
        const int x = (int)(r % w);
        r /= w;
        const int y = (int)(r % h);
        const int ni = (int)(r / h);
        out[idx] = cc < c ? in[((long)(ni * c + cc) * h + y) * w + x] : 0.f;
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int c,
                                    int h, int w) {
    const long total = (long)n * c * h * w;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w);
        long r = idx / w;
        const int y = (int)(r % h);
        r /= h;
        const int cc = (int)(r % c);
        const int ni = (int)(r / c);
        out[idx] = in[((long)(ni * h + y) * w + x) * c + cc];
    }
}

int launch_nchw_to_nhwc(const float* in, float* out, int n, int c, int h, int w, int cpad, hipStream_t stream) {
    const long total = (long)n * h * w * cpad;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(blocks), dim3(256), 0, stream, in, out, n, c, h, w, cpad);
    S3D_LAUNCH_CHECK();
    return 0;
}

// out (N,H,W,C) = a (N,H,W,C) + b (N,C,H,W): the c_fmaps injection of the gen_slices U-Net (openaimodel.py:735-746) without the
// NHWC copy of the feature map in between (the same additions as nchw_to_nhwc + add: bit-identical, one launch less)
__global__ void add_nchw_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n, int c,
                                int h, int w) {
    const long total = (long)n * h * w * c;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % c);
        long r = idx / c;
        const int x = (int)(r % w);
        r /= w;
        const int y = (int)(r % h);
        const int ni = (int)(r / h);
        out[idx] = a[idx] + b[((long)(ni * c + cc) * h + y) * w + x];
    }
}
int launch_add_nchw(const float* a, const float* b, float* out, int n, int c, int h, int w, hipStream_t stream) {
    const long total = (long)n * h * w * c;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(add_nchw_kernel, dim3(blocks), dim3(256), 0, stream, a, b, out, n, c, h, w);
    S3D_LAUNCH_CHECK();
    return 0;
}

int launch_nhwc_to_nchw(const float* in, float* out, int n, int c, int h, int w, hipStream_t stream) {
    const long total = (long)n * c * h * w;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(blocks), dim3(256), 0, stream, in, out, n, c, h, w);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// VGG perceptual loss helpers (vgg_perceptual_loss.py:53-60): ((x+1)/2 - mean)/std, both image sets
// stacked along the batch so every VGG19 conv runs once on 2*n_img images.
// ---------------------------------------------------------------------------------------------
__global__ void vgg_prep_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                const float* __restrict__ mean, const float* __restrict__ stdv,
                                float* __restrict__ out, int n_img, int size) {
    const long hw = (long)size * size;
    const long total = 2L * n_img * hw;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const long ni = idx / hw, r = idx - ni * hw;
        const float* src = ni < n_img ? pred + ni * 3 * hw : target + (ni - n_img) * 3 * hw;
        f32x4 v = zero4();
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = ((src[c * hw + r] + 1.f) / 2.f - mean[c]) / stdv[c];
        float* o = out + idx * 16;
        st4(o, v);
        st4(o + 4, zero4());
        st4(o + 8, zero4());
        st4(o + 12, zero4());
    }
}

// ---------------------------------------------------------------------------------------------
// VGG conv1_1 (3 -> 64, 3x3) straight from the NCHW image pairs, split precision.  K = 3 channels x 9 taps = 27 fits ONE
// k-step of v_mfma_f32_16x16x32_f16: lane (pixel m, g) gathers its eight (channel, tap) values of the pixel's
// neighbourhood (k = 9 ci + 3 dy + dx = 8g + t; the image is L1 / L2 resident, each value is read by nine neighbours),
// normalises them exactly as vgg_prep_kernel does (zero outside the image: the padding applies to the NORMALISED image)
// and splits them; the 64 x 27 weights sit in registers as four hi / lo fragments.  12 MFMAs per 16 pixels; the kernel is
// bound by its 64-channel output (conv_epilogue's full-line stores).  Replaces vgg_prep + the generic tile on the
// 16-channel-padded image, which ran on the fp32 MFMA (C_in 16 is no multiple of the split path's 32): 1.85 ms of the
// training step for 96 images.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vgg_first_f16x3_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                              const float* __restrict__ mean, const float* __restrict__ stdv,
                                                              const float* __restrict__ w, const ConvLaunch a, int n_img) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int S = a.H;
    const long hw = (long)S * S;
    chalf8 wh[4], wl[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = 8 * g + t < 27 ? w[(16 * nt + m) * 27 + 8 * g + t] : 0.f;   // OIHW: k = 9 ci + tap
        s3d_split8(v, wh[nt], wl[nt]);
    }
    int koff[8], kdy[8], kdx[8];
    float kmu[8], ksd[8];
    bool kok[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int k = 8 * g + t, kc = k < 27 ? k : 0;
        const int ci = kc / 9, r = kc - 9 * ci, dy = r / 3, dx = r - 3 * dy;
        kok[t] = k < 27;
        kdy[t] = dy - 1;
        kdx[t] = dx - 1 + m;
        koff[t] = ci * (int)hw;
        kmu[t] = mean[ci];
        ksd[t] = stdv[ci];
    }
    const int tiles_x = S >> 4;
    const long n_tiles = (long)a.N * S * tiles_x;
    // the NEXT tile's eight raw values are requested before this tile is worked on (a tile is a serial chain of a gather, a
    // normalisation, 12 MFMAs and the stores: unpipelined the kernel ran at the latency of that chain, 1.0 ms for 96 images)
    float raw[8];
    unsigned okm = 0u;
    auto gather = [&](long tile) {
        const int tx = (int)(tile % tiles_x);
        const long r = tile / tiles_x;
        const int y = (int)(r % S), ni = (int)(r / S);
        const float* src = ni < n_img ? pred + (long)ni * 3 * hw : target + (long)(ni - n_img) * 3 * hw;
        okm = 0u;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int yy = y + kdy[t], xx = 16 * tx + kdx[t];
            const bool ok = kok[t] && (unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S;
            okm |= ok ? 1u << t : 0u;
            raw[t] = src[koff[t] + (ok ? yy * S + xx : 0)];
        }
    };
    const long stride = (long)gridDim.x * 4;
    long tile = (long)blockIdx.x * 4 + wave;
    if (tile < n_tiles) gather(tile);
    for (; tile < n_tiles; tile += stride) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = ((okm >> t) & 1u) ? ((raw[t] + 1.f) / 2.f - kmu[t]) / ksd[t] : 0.f;
        if (tile + stride < n_tiles) gather(tile + stride);
        chalf8 bh, bl;
        s3d_split8(v, bh, bl);
        S3D_SPLIT_SETTLE();   // partial-register split results feed the MFMAs below straight from registers
        f32x4 acc[1][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], bl, zero4(), 0, 0, 0);
            acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt], bh, acc[0][nt], 0, 0, 0);
            acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], bh, acc[0][nt], 0, 0, 0);
        }
        const int tx = (int)(tile % tiles_x);
        const long r = tile / tiles_x;
        const int pn[1] = {(int)(r / S)}, py[1] = {(int)(r % S)}, px[1] = {16 * tx + m};
        const bool pv[1] = {true};
        conv_epilogue<1, 4>(a, acc, pn, py, px, pv, 0, g);
    }
}
// out: (2*n_img, S, S, 64) NHWC = relu(conv1_1(normalised [pred ; target]) + bias); w: the layer's OIHW fp32 weight
int launch_vgg_first_f16x3(const float* pred, const float* target, const float* mean, const float* stdv, const float* w,
                           const float* bias, float* out, int n_img, int size, hipStream_t stream) {
    S3D_CHECK_ARG(size % 16 == 0 && (long)3 * size * size < (1L << 31), "vgg_first: size %d", size);
    ConvLaunch a = {};
    a.N = 2 * n_img; a.H = size; a.W = size; a.CoutPad = 64;
    a.shift = bias; a.act = S3D_ACT_RELU;
    a.out = out; a.out_mode = S3D_OUT_NHWC; a.cout_store = 64; a.out_cstride = 64;
    const long n_tiles = (long)a.N * size * (size / 16);
    const int blocks = (int)((n_tiles + 3) / 4 < 8192 ? (n_tiles + 3) / 4 : 8192);
    hipLaunchKernelGGL(vgg_first_f16x3_kernel, dim3(blocks), dim3(256), 0, stream, pred, target, mean, stdv, w, a, n_img);
    S3D_LAUNCH_CHECK();
    return 0;
}

int launch_vgg_prep(const float* pred, const float* target, const float* mean, const float* stdv, float* out,
                    int n_img, int size, hipStream_t stream) {
    const long total = 2L * n_img * size * size;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(vgg_prep_kernel, dim3(blocks), dim3(256), 0, stream, pred, target, mean, stdv, out, n_img,
                       size);
    S3D_LAUNCH_CHECK();
    return 0;
}

#define L1_BLOCKS 1024
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         long n4, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 x = ld4(a + 4 * i), y = ld4(b + 4 * i);
        s += (fabsf(x[0] - y[0]) + fabsf(x[1] - y[1])) + (fabsf(x[2] - y[2]) + fabsf(x[3] - y[3]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ partial, int n, float scale,
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

int launch_l1_diff(const float* a, const float* b, long n, float scale, float* partial, float* loss_acc,
                   hipStream_t stream) {
    S3D_CHECK_ARG(n % 4 == 0, "l1_diff: n must be a multiple of 4");
    const long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < L1_BLOCKS ? (n4 + 255) / 256 : L1_BLOCKS);
    hipLaunchKernelGGL(l1_partial_kernel, dim3(blocks), dim3(256), 0, stream, a, b, n4, partial);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, stream, partial, blocks, scale, loss_acc);
    S3D_LAUNCH_CHECK();
    return 0;
}
