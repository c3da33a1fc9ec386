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
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma4(w[nt], b[mt], acc[mt][nt]);
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
#pragma unroll 1
            for (int c = 0; c < cu; ++c) {
                // issue every load of this K chunk (2 per activation tile, 2 per weight tile) before the first
                // use: left alone the scheduler interleaves load / wait / 4 MFMAs and exposes the L2 latency
                f32x4 v0[MT], v1[MT];
                chalf8 wh[NT], wl[NT];
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
                __builtin_amdgcn_sched_barrier(0);
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
template <int MT, int WP, int WC, int NBUF>
__global__ __launch_bounds__(256) void conv3x3_lds_f16x3_kernel(const ConvLaunch a, int tiles_x, int tiles_y,
                                                                int chunks_per_split) {
    static_assert(MT * WP == 8 && WP * WC == 4, "tile is 8 rows, 4 waves");
    constexpr int NT = 2, CO_WG = WC * NT * 16;
    __shared__ __attribute__((aligned(16))) _Float16 s_in[NBUF][2][C3_HALO * C3_PXS];   // [buf][hi|lo]
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
    auto fetch = [&](const ConvSrc& S, int c) {
        const int ni = (S.bmod ? n % S.bmod : n) / S.bdiv;
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
        }
    };
    auto park = [&](int buf) {
        typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int slot = threadIdx.x + 256 * i;
            if (slot < NSLOT) {
                const int pix = slot >> 3, q4 = slot & 7;
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
    fetch(a.src[ch_lo < cu0 ? 0 : 1], ch_lo < cu0 ? ch_lo : ch_lo - cu0);
    park(ch_lo & (NBUF - 1));
    __syncthreads();
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
            fetch(a.src[s1], s1 ? ch + 1 - cu0 : ch + 1);
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
        if (NBUF == 1) __syncthreads();   // everyone is done reading the single buffer before it is refilled
        if (ch + 1 < nchunk) park((ch + 1) & (NBUF - 1));
        __syncthreads();
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
        for (int k = 0; k < nsplit; ++k) s += ld4(a.splitk_ws + ((size_t)k * P + p) * a.CoutPad + co);
        f32x4 acc[1][1] = {{s}};
        const int n = (int)(p / ((long)a.H * a.W));
        const int r = (int)(p - (long)n * a.H * a.W);
        const int pn[1] = {n}, py[1] = {r / a.W}, px[1] = {r % a.W};
        const bool pv[1] = {true};
        conv_epilogue<1, 1>(a, acc, pn, py, px, pv, co >> 4, (co >> 2) & 3);
    }
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
    if (a.splitk_ws && a.out_mode == S3D_OUT_NHWC) {
        while (nblk * splits < 512 && splits * 2 <= nchunk && (size_t)(splits * 2) * out_floats <= a.splitk_floats)
            splits *= 2;
    }
    const int cps = (nchunk + splits - 1) / splits;
    splits = (nchunk + cps - 1) / cps;
    dim3 grid((unsigned)nblk, (unsigned)splits);
    constexpr int one_buf_max = 2;   // chunks per workgroup up to which the single-buffer variant runs
    const bool one = cps <= one_buf_max;   // 32-channel-output layers only: measured -13 % there, +5 % on the 64-wide tile
    if (co_wg == 64) {
        hipLaunchKernelGGL((conv3x3_lds_f16x3_kernel<4, 2, 2, 2>), grid, dim3(256), 0, stream, a, tiles_x, tiles_y, cps);
    } else {
        if (one) hipLaunchKernelGGL((conv3x3_lds_f16x3_kernel<2, 4, 1, 1>), grid, dim3(256), 0, stream, a, tiles_x, tiles_y, cps);
        else hipLaunchKernelGGL((conv3x3_lds_f16x3_kernel<2, 4, 1, 2>), grid, dim3(256), 0, stream, a, tiles_x, tiles_y, cps);
    }
    S3D_LAUNCH_CHECK();
    if (splits > 1) {
        const long total = (long)(out_floats >> 2);
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3(blocks), dim3(256), 0, stream, a, splits);
        S3D_LAUNCH_CHECK();
    }
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
    if (conv3x3_lds_eligible(a)) return launch_conv3x3_lds(a, stream);
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
