// conv.h — internal interface of the implicit-GEMM convolution engine (conv.hip).
#pragma once
#include "common.h"
#include "dropout.h"

// One activation source of a convolution.  The K loop walks sources in order, which gives the
// reference's torch.cat([skip, up], 1) (unet_parts.py:73) without materialising the concat, and
// expand_bs (unet_custom.py:35-38) / the slice-embedding tile (unet_custom.py:52) without tiling:
//   source image index = (bmod ? n % bmod : n) / bdiv ;  sbcast: one C-vector per image, any pixel.
struct ConvSrc {
    const float* p;  // (n_src, H, W, C) NHWC   (or (n_src, C) when sbcast)
    int C;           // multiple of 16
    int bdiv, bmod, sbcast;
};

// GroupNorm (+ FiLM) (+ SiLU) of a convolution's INPUT, applied while the LDS-staged 3x3 kernel parks its input tile (round 5:
// the latent-diffusion U-Net's GN -> SiLU -> conv3x3 chains, openaimodel.py:188-194,229-236).  The GroupNorm statistics
// launch leaves a per-(image, channel) affine table  y = x * A + B  (statistics, gamma / beta and the FiLM scale / shift folded:
// ldm_ops.hip); every workgroup copies its image's table into LDS once, so no normalised tensor is written or read.
// (First form, measured neutral: every workgroup merged the 64 slab moments of its image itself — 2-3 us of dependent loads
// in front of workgroups that live 8-15 us.)
#define S3D_GN_SLICES 64
#define S3D_GN_CMAX 1536     // channels of the widest GroupNorm input of the LDM U-Net (768 + 768 skip concat)
struct ConvGn {
    const float* table;   // [N][2][C]: A | B of every input channel; NULL = no GroupNorm
    int silu;
};

enum { S3D_ACT_NONE = 0, S3D_ACT_RELU = 1, S3D_ACT_TANH = 2 };
enum { S3D_OUT_NHWC = 0, S3D_OUT_CONVT = 1, S3D_OUT_NCHW = 2 };

struct ConvLaunch {
    ConvSrc src[2];
    int nsrc;
    int N, H, W;         // images and spatial size of the output grid (= input grid, "same" padding)
    int ks;              // 1, 3 (pad 1) or 2 (pad 0; with stride 2 = the data-gradient of ConvTranspose2d 2x2 s2)
    int stride;          // 0/1 = 1; 2: input pixel = 2*out + tap offset, input grid is (Hin, Win)
    int Hin, Win;        // input spatial size (0 = same as H, W)
    int CoutPad;         // GEMM N, multiple of 16 (padded rows of the packed weight are zero)
    const float* wpk;    // packed A fragments [CoutPad/16][KU][64][4]
    int KU;              // total K chunks = sum over sources of ks*ks*C/16
    const void* wpk16;   // optional split-precision image: f16 hi/lo fragment pairs [CoutPad/16][KU/2][2][64][8]
                         // (K chunks of 32); used when non-NULL and every source has C % 32 == 0
    int single_pass;     // with wpk16: only the hi*hi product (one f16 MFMA per product, S3D_PREC_F16: NOT fp32-class)
    const float* scale;  // [CoutPad]  y = acc*scale + shift   (BN folded / bias)
    const float* shift;  // [CoutPad]
    int act;
    float* out;
    int out_mode;
    int cout_store;      // channels actually stored (NHWC/NCHW), or Ct of a ConvT (CoutPad = 4*Ct)
    int out_cstride;     // channel stride of the NHWC output tensor
    const float* residual;  // NHWC mode only: v += residual[same index] (after activation)
    int out_accumulate;  // NHWC mode only: out += v
    int xcd_remap;       // set by launch_conv: XCD-aware block -> (pixel tile, cout tile) mapping
    const float* gate;   // NHWC mode only: v = gate[same index] > 0 ? v * gate_scale : 0  (ReLU/dropout backward)
    float gate_scale;
    unsigned long long drop_base;  // element index of this launch's out[0] in the dropout counter space
    float* splitk_ws;    // optional scratch for split-K partial sums (few-pixel deep layers); NULL = never split
    size_t splitk_floats;
    DropCfg drop;        // NHWC mode only: v *= dropout mask (index = output element index), before the residual
    ConvGn gn;           // LDS-staged 3x3 kernel only (launch_conv refuses it elsewhere): GroupNorm of the input, see ConvGn
    int* splits_out;     // HOST pointer, optional.  Non-NULL: a split-K launch leaves its raw partial sums in splitk_ws
                         // ([split][pixel][CoutPad]) and skips the finish pass — *splits_out = number of splits (1: `out` is
                         // final) — for a consumer that sums them itself (the GroupNorm statistics kernels, ldm_ops.hip)
};

int launch_conv(const ConvLaunch& a, hipStream_t stream);
// the finish pass of a split-K launch on its own (a.splitk_ws holds `nsplit` partials): sums them, applies the epilogue of `a`
int launch_conv_splitk_finish(const ConvLaunch& a, int nsplit, hipStream_t stream);

// weight / epilogue packers (device kernels behind them)
enum { S3D_PACK_LINEAR = 0, S3D_PACK_CONV = 1, S3D_PACK_CONVT = 2, S3D_PACK_CONV_DGRAD = 3,
       S3D_PACK_CONVT_DGRAD = 4, S3D_PACK_LINEAR_T = 5 };
// dst fragment image rows [0, n_pad) x chunks [u_off, u_off + ku_seg) of a KU_total-wide image.
//  LINEAR: elem(n,k) = src[n*ld + k]                      (k < k_valid)
//  CONV:   k = tap*cseg + c ; elem = src[(n*cin_tot + cin_begin + c)*taps + tap]  (c < cseg_valid)
//  CONVT:  n = q*ct + co ; elem(n, k=ci) = src[(ci*ct + co)*4 + q]                (ci < k_valid)
//  CONV_DGRAD (data gradient of a ks x ks "same" conv = conv of dY with the flipped, transposed weight):
//          n = ci - cin_begin ; k = tap*cseg + co ; elem = src[((co*cin_tot + cin_begin + n)*taps) + taps-1-tap]
//  CONVT_DGRAD (data gradient of ConvTranspose2d 2x2 s2 = 2x2 stride-2 conv of dY):
//          n = ci ; k = q*ct + co ; elem = src[(n*ct + co)*4 + q]
//  LINEAR_T: elem(n,k) = src[k*ld + n]
struct PackArgs {
    const float* src;
    float* dst;
    int kind;
    int n_valid, n_pad;
    int KU_total, u_off, ku_seg;
    int ld, k_valid;
    int taps, cseg, cseg_valid, cin_tot, cin_begin;
    int ct;
    int f16;       // 1: write the f16 hi/lo pair image for 32-deep K chunks instead of the fp32 image
    int chunk_ku;  // LINEAR only: if > 0 the image is stored [k-chunk group][row tile][chunk_ku] (FFN W2 staging order)
};
int launch_pack(const PackArgs& a, hipStream_t stream);
// While a scope is alive on this thread launch_pack() only queues; flush() issues the queued packs as table launches
// (48 per launch) and ends the queueing.  Nothing that READS a packed image may be launched between the first queued pack and flush().
class PackBatchScope {
public:
    PackBatchScope();
    ~PackBatchScope();
    int flush(hipStream_t stream);
    PackBatchScope(const PackBatchScope&) = delete;
    PackBatchScope& operator=(const PackBatchScope&) = delete;
private:
    void* prev_;
    void* q_;
};
// launch_pack() launches immediately while this is alive (for packs whose SOURCE is a scratch buffer that is reused)
class PackBatchSuspend {
public:
    PackBatchSuspend();
    ~PackBatchSuspend();
private:
    void* saved_;
};
// scale/shift for conv epilogues.  bn may be all-NULL (scale = 1, shift = bias or 0).
// rep > 1 tiles the vector (ConvT: 4 quadrants share the bias).
int launch_fold_bn(const float* bias, const float* const bn[4], float* scale, float* shift, int c_valid,
                   int c_pad, int rep, int bias_before_bn, hipStream_t stream);

int launch_bn_relu_pool(const float* in, const float* scale, const float* shift, float* out, int n, int h,
                        int w, int c, hipStream_t stream);
int launch_nchw_to_nhwc(const float* in, float* out, int n, int c, int h, int w, int cpad, hipStream_t stream);
int launch_nhwc_to_nchw(const float* in, float* out, int n, int c, int h, int w, hipStream_t stream);
int launch_add_nchw(const float* a, const float* b, float* out, int n, int c, int h, int w, hipStream_t stream);   // out NHWC = a NHWC + b NCHW

// VGG-loss helpers (conv.hip)
int launch_vgg_prep(const float* pred, const float* target, const float* mean, const float* stdv, float* out,
                    int n_img, int size, hipStream_t stream);   // -> (2*n_img, S, S, 16) NHWC, normalised
// conv1_1 + ReLU straight from the NCHW image pairs, split precision (replaces vgg_prep + the first conv): out (2*n_img, S, S, 64)
int launch_vgg_first_f16x3(const float* pred, const float* target, const float* mean, const float* stdv, const float* w_oihw,
                           const float* bias, float* out, int n_img, int size, hipStream_t stream);
int launch_l1_diff(const float* a, const float* b, long n, float scale, float* partial, float* loss_acc,
                   hipStream_t stream);                         // loss_acc[0] += scale * sum|a-b| (deterministic)
