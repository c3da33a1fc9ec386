/*
 * slice3d_hip.h — C ABI of libslice3d_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the Slice3D regression hot path.  The reference has NO plugin / FFI layer for
 * this path (it is nn.Module code over ATen ops); the boundary a maintainer binds is
 * therefore "one C entry point per reference function", each citing the reference code it replaces.
 * The Python mirror of the reference's module API (slice3d_amd/models.py, generator.py) calls these
 * through ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all memory
 *     (outputs, packed weights, workspace).  The library never allocates or frees device memory and
 *     keeps no global device state.
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns immediately.
 *   - return value: 0 = ok, <0 = S3D_E_* argument/size error (message via s3d_last_error()),
 *     >0 = a hipError_t raised by a launch.
 *   - all tensors are fp32.  Activations the library produces are channels-last (NHWC); the image
 *     input and the reconstructed slice images cross the boundary in the reference's NCHW layout.
 *   - batch x slice flattening is batch-major (row b*n_slices+s), as in the reference
 *     (unet_custom.py:35-38, models.py:66,70,78).
 */
#ifndef SLICE3D_HIP_H
#define SLICE3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S3D_VERSION 115          /* 0.1.1: s3d_conv_fwd prec semantics, s3d_conv_gn_supported; 111: s3d_decode_set_last_fused, s3d_decode_set_shared_footprint; 112: S3D_PREC_F16 accepted by s3d_train_*; 113: atomic-free sampling backward (bit-reproducible s3d_train_* gradients, larger workspace); 114: s3d_qkv_attention_ws_* serve head width 48; 115: s3d_add_nchw_fwd */
#define S3D_E_ARG (-1)           /* bad argument / unsupported shape */
#define S3D_E_WORKSPACE (-2)     /* workspace or packed-weight buffer too small */

#define S3D_N_LEVELS 5           /* feature pyramid levels, coarse -> fine (unet_custom.py:58-67) */
#define S3D_D_MODEL 128
#define S3D_N_LAYERS 3
#define S3D_FFN 2048
#define S3D_N_TOKENS_MAX 13      /* 1 point token + up to 12 slice tokens (models.py:82) */

/* arithmetic mode of the MFMA contractions (argument `prec`) */
#define S3D_PREC_F32 0           /* v_mfma_f32_16x16x4_f32: exact fp32, the parity mode */
#define S3D_PREC_F16X3 1         /* fp32 operands split into f16 hi+lo, 3 f16 MFMAs per product (22-bit
                                    significands, fp32 accumulate): fp32-class results; conv / attention / FFN */
#define S3D_PREC_F16 2           /* THROUGHPUT MODE, not fp32-class: operands rounded to f16, ONE f16 MFMA per product,
                                  * fp32 accumulation and fp32 everywhere else (LayerNorm, softmax, sampling).  What
                                  * BASELINE configs[1]'s "bf16" names; fails the 1e-4 parity gate by construction and is
                                  * reported separately with its measured error (bench.py `throughput_mode_f16`).
                                  * Inference entry points of Slices3DRegModel / Slices3DGTModel only. */
#define S3D_PREC_BF16 3          /* THROUGHPUT MODE on the bf16 MFMA (round 5), Slices3DRegModel inference only: the decoder's
                                  * attention and FFN GEMMs (94 % of the decoder's FLOPs; layers 0-1 and the final FFN) run
                                  * v_mfma_f32_16x16x32_bf16 on bf16-rounded operands (weights from a bf16 image, activations
                                  * through v_cvt_pk_bf16_f32), one MFMA per product, fp32 accumulation; every other
                                  * contraction of the path (U-Net, latent projections, token builder, absorbed last-layer
                                  * GEMMs) runs as in S3D_PREC_F16.  Same MFMA rate as S3D_PREC_F16, 8 significand bits
                                  * instead of 11: reported with its measured error (bench.py `throughput_mode_bf16`). */

int s3d_version(void);
const char* s3d_last_error(void);           /* thread-local, valid until the next failing call */

/* ---------------------------------------------------------------------------------------------
 * U-Net slice generator  — replaces UNet.forward (reg_slices/src/unet_custom.py:40-69) and the
 * DoubleConv/Up/OutConv blocks it calls (reg_slices/src/unet_parts.py:8-84), eval-mode BatchNorm.
 * ------------------------------------------------------------------------------------------- */

/* Raw parameters in the reference's own state_dict layout (PyTorch conv weight [Cout][Cin][kh][kw],
 * ConvTranspose2d weight [Cin][Cout][2][2]).  bn_* entries are {weight,bias,running_mean,running_var}. */
typedef struct {
    const float* w;      /* conv weight */
    const float* b;      /* conv bias or NULL */
    const float* bn[4];  /* BN following this conv (gamma, beta, mean, var) or all NULL */
} S3dConvParams;

typedef struct {
    S3dConvParams enc[13];        /* VGG16-BN convs down1.0 ... down5.40 with the BN that follows each
                                     (enc[12].bn = down5_.41 is unused by the forward and may be NULL) */
    S3dConvParams trans_c;        /* 1x1 (512+128)->512, bias            unet_custom.py:22 */
    S3dConvParams trans_up[4];    /* 1x1 skip projections trans_up1..4   unet_custom.py:24-30 */
    S3dConvParams up_t[4];        /* ConvTranspose2d 2x2 s2 of up1..4    unet_parts.py:53 */
    S3dConvParams up_c1[4];       /* DoubleConv conv0 + BN1 (no bias)    unet_parts.py:16-17 */
    S3dConvParams up_c2[4];       /* DoubleConv conv3 + BN4 (no bias)    unet_parts.py:19-20 */
    S3dConvParams outc;           /* 1x1 32->3 + tanh                    unet_parts.py:78-84 */
    const float* emds;            /* (n_slices,128) slice embeddings     unet_custom.py:32 */
    int n_slices;
} S3dUNetParams;

/* Bytes of the packed (MFMA-fragment-ordered, BN-folded) weight image for n_slices. */
size_t s3d_unet_packed_bytes(int n_slices);
/* Repack raw parameters into `packed` (device, >= s3d_unet_packed_bytes).  Re-run after every
 * parameter update.  `params_host` is a HOST struct of DEVICE pointers. */
int s3d_unet_pack(const S3dUNetParams* params_host, void* packed, size_t packed_bytes, void* stream);

/* Feature pyramid handle: level l is (B*n_slices, H_l, W_l, C_l) NHWC fp32 with
 * C = {512,256,128,64,32}, H_l = S/16 * 2^l  (coarse -> fine).  Caller allocates. */
typedef struct {
    float* level[S3D_N_LEVELS];
    int n_img;                    /* B*n_slices */
    int size;                     /* S (input image side; levels are S/16 ... S) */
} S3dPyramid;

size_t s3d_unet_workspace_bytes(int batch, int size, int n_slices);
/* img (B,3,S,S) NCHW in [-1,1]  ->  pyramid (5 NHWC levels) and, if slices_rec != NULL, the
 * reconstructed slice images (B*n_slices,3,S,S) NCHW (tanh output, models.py:65-66).
 * S must be a multiple of 16. */
int s3d_unet_encode_fwd(const void* packed, const float* img, const S3dPyramid* out,
                        float* slices_rec, int batch, int size, int n_slices, int prec,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-query decoder — replaces Slices3DRegModel.forward lines models.py:53-84:
 * query rotation / flip, project_coord (models.py:28-36), 5x sample_from_planes (models.py:38-46,
 * 69-78), fc_p / fc_s (models.py:79-80), the 3-layer post-LN TransformerEncoder (models.py:18-19,83)
 * and fc_out (models.py:22-24,84).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    const float* in_proj_w;  const float* in_proj_b;    /* (384,128), (384) */
    const float* out_proj_w; const float* out_proj_b;   /* (128,128), (128) */
    const float* lin1_w;     const float* lin1_b;       /* (2048,128), (2048) */
    const float* lin2_w;     const float* lin2_b;       /* (128,2048), (128) */
    const float* norm1_w;    const float* norm1_b;      /* (128) */
    const float* norm2_w;    const float* norm2_b;      /* (128) */
} S3dLayerParams;

typedef struct {
    const float* fc_p_w;  const float* fc_p_b;          /* (128,3), (128)   models.py:20 */
    const float* fc_s_w;  const float* fc_s_b;          /* (128,992), (128) models.py:21 */
    S3dLayerParams layer[S3D_N_LAYERS];                 /* att_decoder.layers.{0,1,2} */
    const float* fc_out_w; const float* fc_out_b;       /* (1,128), (1)     models.py:22-24 */
} S3dHeadParams;

size_t s3d_head_packed_bytes(void);
int s3d_head_pack(const S3dHeadParams* params_host, void* packed, size_t packed_bytes, void* stream);

/* "Latent code" c of the ConvONet-style encode()/decode() split: the three coarse pyramid levels with
 * fc_s folded in (bilinear sampling and fc_s are both linear, so sampling the projected maps equals
 * projecting the sampled features; SURVEY.md section 7 "Sampler layout"), plus the two fine levels raw.
 * proj[l] is (n_img, H_l, W_l, 128) NHWC for l = 0,1,2. */
typedef struct {
    float* proj[3];
    const float* fine[2];         /* pyramid levels 3 and 4, (n_img,S/2,S/2,64), (n_img,S,S,32) */
    int n_img;
    int size;
} S3dLatent;

int s3d_latent_build(const void* head_packed, const S3dPyramid* pyr, const S3dLatent* out, int prec, void* stream);

/* Decode workspace.  The decoder walks the queries in passes; `s3d_decode_workspace_bytes` sizes the preferred pass
 * (<= 524 288 queries: 6.4 GB for any larger job — X 3.5 GB, the absorbed last layer's scratch 2.7 GB), and
 * `s3d_decode_workspace_bytes_min` the smallest one the library accepts (65 536 queries per pass: 0.8 GB).  A workspace between
 * the two makes the decode calls halve the pass until it fits (more launch tails, same results); below the minimum they
 * return S3D_E_WORKSPACE. */
size_t s3d_decode_workspace_bytes(int batch, long n_qry, int n_slices);
size_t s3d_decode_workspace_bytes_min(int batch, long n_qry, int n_slices);
/* 1 (default): everything on the caller's stream.  2 (or env S3D_DECODE_LANES=2): passes of >= 131 072 queries run their two
 * halves' layer chains on the caller's stream and on a library-owned side stream (created once per device, ordered behind
 * the caller's stream by events on both ends of every call), so one half's attention kernels share the CUs with the other
 * half's FFN kernels.  Same results bit for bit; measured time-neutral on MI355X (profiles/r05_lanes_ab.md), kept opt-in.
 * Thread safety: the setting is process-wide (an atomic).  The side stream and its two events are shared by every caller of a
 * device; a call holds that device's mutex from its fork to its join (host-side enqueue time only), so host threads may
 * decode concurrently on one device with different caller streams — their side-stream work is serialised, never interleaved.
 * The objects are created under the same mutex, all or nothing; when they cannot be created the call runs in its one-stream
 * form. */
int s3d_decode_set_lanes(int n);
/* 1 (default; env S3D_LAST_FUSED=0 turns it off): the token-0 attention block of the last encoder layer (models.py:83 consumes
 * nothing else) runs as ONE kernel in the split-precision modes (csrc/decode_last.hip: absorbed q GEMM, 13-row mixing step,
 * absorbed output GEMM + residual; only the token rows are read and the 512-byte u rows written).  0: the four-launch form of
 * rounds 2-5 (row copy, row GEMM, mixing kernel, row GEMM).  Same results bit for bit (every accumulator adds the same
 * products in the same order); the switch exists for the test that says so and for A/B timing.  S3D_PREC_F32 always runs the
 * four-launch form.  Process-wide (an atomic). */
int s3d_decode_set_last_fused(int on);
/* 1 (default; env S3D_SHARED_FOOTPRINT=0 turns it off): the token builder evaluates the three folded pyramid levels of a group of
 * 16 queries through the group's shared 4 x 4 pixel window on the fp32 MFMA whenever the group's bilinear footprints fit one
 * (csrc/decode.hip, sample_tokens_kernel) and per lane otherwise.  0: always per lane.  Same results bit for bit (the MFMA is a
 * k-ordered fmaf chain and the window visits a query's taps in the per-lane order); the switch exists for the test that says so
 * and for A/B timing.  Applies to the inference decode calls and to s3d_train_step.  Process-wide (an atomic). */
int s3d_decode_set_shared_footprint(int on);
/* qry (B,Q,3); rot (B,3,3) or NULL; trans (B,4,3) = trans_mat_wo_rot_tp; flip_yz != 0 selects the
 * mode='test' prologue (y,z negated, no rotation; models.py:53-56).  sdf_out (B,Q). */
int s3d_decode_points_fwd(const void* head_packed, const S3dLatent* latent, const float* qry,
                          const float* rot, const float* trans, int flip_yz, float* sdf_out,
                          int batch, long n_qry, int n_slices, int prec,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Debug / test form of s3d_decode_points_fwd that also returns the intermediate rows the reference's stage probes look at
 * (fc_p / fc_s output models.py:79-82, token 0 after each layer of att_decoder models.py:83): stages holds
 * s3d_decode_stages_floats() floats = the token tensor [G][n_slices+1][16][128] (G = B*ceil(Q/16); row of query q of
 * object b: group b*ceil(Q/16) + q/16, lane q%16; token 0 = fc_p, token 1+s = fc_s of slice s), then [3][G*16][128] token-0
 * rows after layers 0, 1, 2.  One pass in caller order: Q < 4096 per object. */
size_t s3d_decode_stages_floats(int batch, long n_qry, int n_slices);
int s3d_decode_points_stages_fwd(const void* head_packed, const S3dLatent* latent, const float* qry, const float* rot,
                                 const float* trans, int flip_yz, float* sdf_out, float* stages, int batch, long n_qry,
                                 int n_slices, int prec, void* workspace, size_t workspace_bytes, void* stream);

/* Dense-grid evaluation for Generator3D with upsampling_steps == 0 (reconstruct.py:135-146):
 * query coordinates box*linspace(-.5,.5,nx)^3 (x slowest, z fastest; common.py:145-164) are generated
 * in-kernel; logits_out[nx^3] = -sdf (reconstruct.py:97).  batch is 1. */
int s3d_decode_grid_fwd(const void* head_packed, const S3dLatent* latent, const float* trans,
                        int nx, float box, float* logits_out, int n_slices, int prec,
                        void* workspace, size_t workspace_bytes, void* stream);
/* A contiguous slab [q_begin, q_begin + q_count) of the same grid's linear index (x slowest, z fastest):
 * logits_out[q_count].  This is the query-parallel split of reconstruct.py:135-146 over N processes
 * (SURVEY.md 8(e)): every rank encodes the object itself and decodes 1/N of the grid; the host gathers.
 * Workspace: s3d_decode_workspace_bytes(1, q_count, n_slices). */
int s3d_decode_grid_slab_fwd(const void* head_packed, const S3dLatent* latent, const float* trans,
                             int nx, float box, long q_begin, long q_count, float* logits_out,
                             int n_slices, int prec, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * VGG19 perceptual loss — replaces VGGPerceptualLoss.forward / VGG19Feats.forward
 * (reg_slices/src/vgg_perceptual_loss.py:29-70) as called at models.py:90-92.
 * Taps are what the reference actually compares: torchvision's in-place ReLU makes taps 1-4 post-ReLU,
 * tap 5 (conv5_2) pre-ReLU.  loss_out[0] = 0.001 * sum_i w_i * mean|x_i - y_i|.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    S3dConvParams conv[14];       /* vgg19.features convs 0,2,5,7,10,12,14,16,19,21,23,25,28,30 (bn NULL) */
    const float* mean;            /* (3) ImageNet mean buffer  vgg_perceptual_loss.py:47 */
    const float* std;             /* (3) */
} S3dVggParams;
size_t s3d_vgg_packed_bytes(void);
int s3d_vgg_pack(const S3dVggParams* params_host, void* packed, size_t packed_bytes, void* stream);
size_t s3d_vgg_workspace_bytes(int n_img, int size);
/* pred, target: (n_img,3,S,S) NCHW in [-1,1]; S multiple of 16. */
int s3d_vgg_loss_fwd(const void* packed, const float* pred, const float* target, int n_img, int size,
                     float* loss_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Slices3DGTModel — reg_slices/src/model_gt.py:12-111 (regression from GIVEN slice images).
 * Same split as above: encoder once per object, folded latent maps, per-query decode.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    S3dConvParams conv[13];       /* img_encoder (vgg16bn_feats.py:26-40): VGG16-BN convs 0,3,7,10,14,17,20,24,27,30,
                                     34,37,40 each with the BatchNorm that follows it (conv[12].bn unused) */
} S3dVgg16BnParams;
typedef struct {
    float* level[5];              /* raw conv1_2, conv2_2, conv3_3, conv4_3, conv5_3 outputs, channels-last:
                                     (n_img, S>>l, S>>l, {64,128,256,512,512}[l]) */
    int n_img;                    /* B*n_slices */
    int size;
} S3dGtPyramid;
size_t s3d_gt_encoder_packed_bytes(void);
int s3d_gt_encoder_pack(const S3dVgg16BnParams* params_host, void* packed, size_t packed_bytes, void* stream);
size_t s3d_gt_encoder_workspace_bytes(int n_img, int size);
/* img_slices (n_img,3,S,S) NCHW = feed_dict['img_slices'].view(B*n_slices,3,S,S) (model_gt.py:73-76) */
int s3d_gt_encode_fwd(const void* packed, const float* img_slices, const S3dGtPyramid* out, int n_img, int size,
                      int prec, void* workspace, size_t workspace_bytes, void* stream);

typedef struct {
    const float* pts_w[3];  const float* pts_b[3];      /* pts_feat_extractor.{0,2,4}: (32,3) (64,32) (128,64) */
    const float* local_w[2]; const float* local_b[2];   /* fc_local.{0,2}: (128,1472) (128,128) */
    S3dLayerParams layer[S3D_N_LAYERS];                 /* att_decoder.layers.{0,1,2} */
    const float* fc_out_w; const float* fc_out_b;       /* fc_out.0 */
} S3dGtHeadParams;
size_t s3d_gt_head_packed_bytes(void);
int s3d_gt_head_pack(const S3dGtHeadParams* params_host, void* packed, size_t packed_bytes, void* stream);
/* proj[0..3]: conv5_3, conv4_3, conv3_3, conv2_2 levels with fc_local.0 folded in (n_img, W, W, 128);
 * fine: the raw conv1_2 level (must alias pyramid level 0). */
typedef struct {
    float* proj[4];
    const float* fine;
    int n_img;
    int size;
} S3dGtLatent;
int s3d_gt_latent_build(const void* head_packed, const S3dGtPyramid* pyr, const S3dGtLatent* out, int prec,
                        void* stream);
size_t s3d_gt_decode_workspace_bytes(int batch, long n_qry, int n_slices);
int s3d_gt_decode_points_fwd(const void* head_packed, const S3dGtLatent* latent, const float* qry,
                             const float* rot, const float* trans, int flip_yz, float* sdf_out, int batch,
                             long n_qry, int n_slices, int prec, void* workspace, size_t workspace_bytes,
                             void* stream);
int s3d_gt_decode_grid_fwd(const void* head_packed, const S3dGtLatent* latent, const float* trans, int nx,
                           float box, float* logits_out, int n_slices, int prec, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Latent-diffusion denoising U-Net primitives — gen_slices/ldm/modules/diffusionmodules/openaimodel.py
 * (BASELINE configs[4]).  The reference builds UNetModel from nn.Conv2d / GroupNorm32 / SiLU /
 * QKVAttentionLegacy / F.interpolate / avg_pool2d; the host mirror (slice3d_amd/ldm_unet.py) builds the same
 * module tree over these entry points.  All tensors channels-last fp32; C padded to a multiple of 16 where a
 * convolution reads it.
 * ------------------------------------------------------------------------------------------- */
/* nn.Conv2d(cin0+cin1, cout, ks, padding=ks/2) on cat([x0, x1], channel) (openaimodel.py:772 th.cat + conv), also
 * nn.Conv1d(k=1) over tokens (qkv / proj_out, :291,:299) and the ResBlock skip 1x1 (:240). */
size_t s3d_conv_packed_bytes(int cout, int cin0, int cin1, int ks);
int s3d_conv_pack(const float* w, const float* bias, int cout, int cin0, int cin1, int ks, void* packed,
                  size_t packed_bytes, void* stream);
/* prec (since version 110): S3D_PREC_F32 = exact fp32 MFMA; S3D_PREC_F16X3 = three f16 MFMAs per product (fp32-class);
 * S3D_PREC_F16 = the single-pass throughput mode (ONE f16 MFMA per product, operands rounded to f16 — not fp32-class; before
 * version 110 this value selected the exact fp32 path here).  Layers without an f16 weight image (channel counts that are not
 * multiples of 32: the stem) run the exact fp32 MFMA in every mode.  Any other value (S3D_PREC_BF16 included): S3D_E_ARG. */
int s3d_conv_fwd(const void* packed, const float* x0, const float* x1, const float* residual, float* out, int N,
                 int H, int W, int cout, int cin0, int cin1, int ks, int prec, void* workspace,
                 size_t workspace_bytes, void* stream);
/* GroupNorm32 -> [FiLM] -> [SiLU] -> Conv2d(3x3) as ONE operator (openaimodel.py:188-194 ResBlock.in_layers, :229-236
 * out_layers with use_scale_shift_norm): s3d_group_norm_table_fwd folds the statistics of cat([x0, x1]), gamma / beta and the
 * FiLM rows (film: (N, film_stride >= 2 (c0 + c1)) rows of scale | shift, or NULL) into a per-(image, channel) affine table
 * (N * 2 * (c0 + c1) floats: A | B of y = x * A + B; stats = s3d_group_norm_stats_floats(N, groups) floats of scratch);
 * s3d_conv_gn_fwd applies it (+ SiLU) while it stages its input tile — the normalised tensor is never written.  Served for
 * ks == 3, split precision (f16x3 / f16), cin0, cin1, cout multiples of 32, cin0 + cin1 <= 1536; other shapes are refused
 * with S3D_E_ARG (use s3d_group_norm_fwd + s3d_conv_fwd). */
size_t s3d_group_norm_stats_floats(int N, int groups);
/* 1 if s3d_conv_gn_fwd serves the shape (H, W <= 0: map size not known yet), else 0 — the single statement of the fused
 * operator's eligibility; on 0 run s3d_group_norm_fwd + s3d_conv_fwd. */
int s3d_conv_gn_supported(int cout, int cin0, int cin1, int ks, int prec, int H, int W, int has_workspace);
/* A convolution output whose split-K finish pass is deferred to its consumer (s3d_conv_gn_fwd with nsplit_out != NULL reported
 * nsplit > 1): the raw partial sums [nsplit][N*H*W][cout] sit in the convolution's workspace; `out` (N,H,W,cout) is written by
 * whoever consumes the descriptor — the next GroupNorm's statistics kernel (s3d_group_norm_table_fwd / s3d_group_norm_partial_fwd:
 * it sums the splits, adds the bias of `conv_packed` and `residual`, stores `out` and goes on with the values) or
 * s3d_conv_finish_fwd.  The descriptor is dead once the workspace is reused (the next split-K convolution). */
typedef struct {
    const float* part;
    int nsplit;
    const void* conv_packed;   /* the producing convolution's packed weights (holds its bias) */
    int cout, cin0, cin1, ks;
    const float* residual;     /* or NULL */
    float* out;
} S3dConvPartial;
int s3d_group_norm_table_fwd(const float* x0, int c0, const float* x1, int c1, const float* gamma, const float* beta,
                             const float* film, long film_stride, float* table, float* stats, int N, int HW, int groups,
                             float eps, const S3dConvPartial* x0_partial, void* stream);
int s3d_group_norm_partial_fwd(const S3dConvPartial* x0_partial, const float* gamma, const float* beta, const float* film,
                               long film_stride, float* y, float* stats, int N, int HW, int groups, float eps, int silu,
                               void* stream);
int s3d_conv_gn_fwd(const void* packed, const float* x0, const float* x1, const float* residual, float* out, int N, int H,
                    int W, int cout, int cin0, int cin1, int ks, int prec, const float* table, int silu, void* workspace,
                    size_t workspace_bytes, int* nsplit_out, void* stream);
int s3d_conv_finish_fwd(const S3dConvPartial* partial, int N, int H, int W, void* stream);
/* GroupNorm32 (util.py normalization) [+ FiLM: y*(1+scale)+shift, film = (N, 2C), openaimodel.py:268-270] [+ SiLU].
 * stats: s3d_group_norm_stats_floats(N, groups) floats of scratch (N*groups*192 since round 4; a smaller buffer is
 * overrun silently — size it with the function). */
int s3d_group_norm_fwd(const float* x, const float* gamma, const float* beta, const float* film, float* y,
                       float* stats, int N, int HW, int C, int groups, float eps, int silu, void* stream);
/* the same with film rows read in place from a wider tensor: image n's (scale | shift) at film + n * film_stride floats
 * (replaces ResBlock.emb_layers + the chunk of openaimodel.py:262-270 for a block whose emb_layers output is a column
 * range of one stacked GEMV over all blocks) */
int s3d_group_norm_film_fwd(const float* x, const float* gamma, const float* beta, const float* film, long film_stride,
                            float* y, float* stats, int N, int HW, int C, int groups, float eps, int silu, void* stream);
/* the same on th.cat([x0, x1], dim=1) (openaimodel.py:750, the skip connections of the output blocks) without
 * materialising the concatenation: x0 (N,HW,c0), x1 (N,HW,c1) -> y (N,HW,c0+c1); c0, c1 multiples of 4 */
int s3d_group_norm2_fwd(const float* x0, int c0, const float* x1, int c1, const float* gamma, const float* beta,
                        const float* film, float* y, float* stats, int N, int HW, int groups, float eps, int silu,
                        void* stream);
/* QKVAttentionLegacy.forward (openaimodel.py:362-377): qkv (N, T, heads*3*ch) -> out (N, T, heads*ch);
 * prec: S3D_PREC_F32 = fp32 MFMA, S3D_PREC_F16X3 = split-precision f16 MFMA (fp32-class accuracy) */
int s3d_qkv_attention_fwd(const float* qkv, float* out, int N, int T, int heads, int ch, int prec, void* stream);
/* The same operator for head widths 8 / 16 / 24 / 32 / 48 on the f16 MFMA with fp32-class logits (q, k split three ways,
 * p, v two ways; replaces the same reference lines, openaimodel.py:353-381).  ws: s3d_qkv_attention_ws_bytes() bytes of
 * scratch for the pre-split K / V block images (0 = width not served).  Width 48 (version 114): two 32-channel k-steps;
 * where N * heads * ceil(T / 64) workgroups would not fill the chip (1 024 tokens at batch 1) the keys are split over
 * 2 - 8 workgroups per query block and a merge launch adds their partial results (the scratch holds those as well). */
size_t s3d_qkv_attention_ws_bytes(int N, int T, int heads, int ch);
int s3d_qkv_attention_ws_fwd(const float* qkv, float* out, int N, int T, int heads, int ch, void* ws, size_t ws_bytes,
                             void* stream);
/* Upsample / Downsample with use_conv=False (openaimodel.py:108-158): up != 0: nearest 2x; else 2x2 average pool */
int s3d_resample2x_fwd(const float* x, float* y, int N, int H, int W, int C, int up, void* stream);
/* linear(SiLU?(x)): time_embed (:506-510) and ResBlock.emb_layers (:222-228); x (N,K), w (M,K) */
int s3d_small_linear_fwd(const float* x, const float* w, const float* b, float* out, int N, int K, int M,
                         int silu_in, void* stream);
/* timestep_embedding (util.py:151-170) */
int s3d_timestep_embedding_fwd(const float* t, float* out, int N, int dim, float max_period, void* stream);
/* out = a + b (c_fmaps injection, openaimodel.py:735-746) */
int s3d_add_fwd(const float* a, const float* b, float* out, long n, void* stream);
int s3d_nchw_to_nhwc_pad(const float* in, float* out, int n, int c, int h, int w, int cpad, void* stream);
/* out (N,H,W,C) = a (N,H,W,C) + b (N,C,H,W): the same injection with the feature map taken as the reference hands it over (version 115) */
int s3d_add_nchw_fwd(const float* a, const float* b, float* out, int n, int c, int h, int w, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training step — replaces train_step (reg_slices/train.py:41-53): train-mode forward (batch-statistic
 * BatchNorm with the running-stat update, unet_parts.py:17,20), the three losses of cal_loss_pred
 * (train.py:29-39) + cal_acc (train.py:21-27), backward of everything, and Adam (train.py:136).
 * Gradients are written into caller-owned buffers laid out like the parameters: `unet_grad` / `head_grad`
 * are the same structs with every pointer replaced by the gradient buffer of that tensor (bn[0], bn[1] =
 * d gamma, d beta; bn[2], bn[3] ignored).  BatchNorm running_mean / running_var are updated IN PLACE
 * through unet->...bn[2], bn[3].  losses_out[4] = {L1(sdf), L1(slices), vgg_loss (x0.001), sign accuracy}.
 * The caller all-reduces the gradient buffers across ranks (data parallel) between this call and
 * s3d_adam_step.  dropout_p is nn.TransformerEncoderLayer's dropout (reference default 0.1); masks are
 * counter-based functions of (seed, site, element index), regenerated in the backward pass.
 * ------------------------------------------------------------------------------------------- */
/* Cross-rank BatchNorm statistics (train.py has no counterpart: the reference's nn.DataParallel replicas use
 * per-replica statistics, like torch DDP by default; `--sync_bn` is SURVEY.md 8(e)'s option).  The library computes
 * the per-rank partial statistics, calls all_reduce_sum on `n_floats` floats of `scratch` (device memory, >= 2048
 * floats, owned by the caller) on the call's stream, and continues with the reduced values: per BatchNorm layer two
 * small all-reduces in the forward (the means, then var_r + (mu_r - mu)^2: the count-weighted merge of
 * torch.nn.SyncBatchNorm, free of E[x^2] - mu^2 cancellation) and one in the backward: 42 + 21 per train step.  The callback must enqueue the collective so that later work
 * on `stream` sees its result (torch.distributed.all_reduce on the current stream does) and return 0. */
typedef int (*s3d_all_reduce_sum_fn)(void* user, float* device_buf, long n_floats, void* stream);
typedef struct {
    s3d_all_reduce_sum_fn all_reduce_sum;
    void* user;
    int world_size;
    float* scratch;
} S3dSyncBn;
typedef struct {
    const float* img;          /* (B,3,S,S)            img_input            */
    const float* img_slices;   /* (B,3*n_slices,S,S)   img_slices           */
    const float* qry;          /* (B,Q,3)              qry_norot            */
    const float* rot;          /* (B,3,3)              obj_rot_mat          */
    const float* trans;        /* (B,4,3)              trans_mat_wo_rot_tp  */
    const float* sdf;          /* (B,Q)                sdf                  */
    /* Optional hipEvent_t handles (NULL = none), recorded on the call's stream the moment a BUCKET of parameter
     * gradients is final, in the order the backward finishes them, so a data-parallel host can all-reduce bucket k on
     * another stream while the rest of the backward runs (replaces the backward-hook bucketing of
     * torch DDP for train.py:131-132).  [0] transformer decoder + fc_p / fc_s / fc_out; [1] the U-Net's decoder
     * half (trans_c, up*, trans_up*, outc, emds); [2] encoder convs 7..12 + their BatchNorms (13 of the encoder's
     * 14.7 M parameters); the rest (encoder convs 0..6) is final when the call's work is. */
    void* ev_grad_ready[3];
    const S3dSyncBn* sync_bn;  /* NULL: per-rank BatchNorm statistics */
} S3dTrainBatch;
size_t s3d_train_workspace_bytes(int batch, int size, long n_qry, int n_slices);
/* Reproducibility (version 113): with n_qry >= 4096 (the locality-sorted token order) and size <= 256 the step contains no float
 * atomics — the pyramid-sampling backward writes per-tile partial sums into the workspace and adds them in a fixed order
 * (train_sbd.hip) — so the same inputs, seed and weights give bit-identical losses and gradients, call after call.  Smaller query
 * counts / larger images keep the atomic scatter kernels (gradients equal to fp32 summation-order noise).  The workspace holds
 * 98 KB of partial sums per (object, slice, 16 x 16-bin image tile): 1.2 GB at 4 objects x 12 slices.
 * prec: S3D_PREC_F32 (exact fp32 MFMAs), S3D_PREC_F16X3 (split precision, fp32-class: the mode every parity test and the
 * reported train_samples_per_s use) or — since version 112 — S3D_PREC_F16: a THROUGHPUT mode in which the decoder's GEMM kernels
 * (FFN forward / data pass / both weight-gradient contractions, the fused attention forward and backward, the row-linear layers and
 * their weight gradients) run ONE f16 MFMA per product; the U-Net, VGG, the samplers, every reduction, the fp32 master weights,
 * fp32 accumulation and the power-of-two backward scale are those of S3D_PREC_F16X3.  Not fp32-class (gradients deviate by ~1e-2
 * relative from the split-precision step): bench.py reports it beside the headline as train_throughput_mode_f16. */
int s3d_train_fwd_bwd(const S3dUNetParams* unet, const S3dHeadParams* head, const S3dVggParams* vgg,
                      const S3dUNetParams* unet_grad, const S3dHeadParams* head_grad,
                      const S3dTrainBatch* batch, int batch_size, int size, long n_qry, int n_slices,
                      float dropout_p, unsigned long long seed, int prec, float* losses_out, float* sdf_pred_out,
                      float* slices_rec_out, void* workspace, size_t workspace_bytes, void* stream);
/* The same step as two calls, for a host that owns the loss — the reference's contract `x = model(batch);
 * loss(x).backward(); opt.step()` (train.py:41-53) behind a torch.autograd.Function:
 *   s3d_train_fwd : train-mode forward (batch-statistics BatchNorm with running-stat updates, dropout from `seed`);
 *                   outputs sdf_pred (B,Q), slices_rec (B*n_slices,3,S,S), vgg_loss (device scalar, already x0.001 as
 *                   models.py:92); every activation the backward needs stays in `workspace`.
 *   s3d_train_bwd : from d loss/d sdf_pred, d loss/d slices_rec (device, either may be NULL = zero) and
 *                   d loss/d vgg_loss (host scalar) to the parameter gradients (written, not accumulated).  Same
 *                   dims / dropout_p / seed / prec / workspace as the forward; slices_rec = the forward's output.
 *                   grad_scale: power of two applied to the incoming gradients inside the split-precision (f16x3)
 *                   backward and removed from the results; 0 selects 2^k ~ 8*B*Q, right for mean-reduced losses. */
int s3d_train_fwd(const S3dUNetParams* unet, const S3dHeadParams* head, const S3dVggParams* vgg,
                  const S3dTrainBatch* batch, int batch_size, int size, long n_qry, int n_slices, float dropout_p,
                  unsigned long long seed, int prec, float* vgg_loss_out, float* sdf_pred_out, float* slices_rec_out,
                  void* workspace, size_t workspace_bytes, void* stream);
int s3d_train_bwd(const S3dUNetParams* unet, const S3dHeadParams* head, const S3dVggParams* vgg,
                  const S3dUNetParams* unet_grad, const S3dHeadParams* head_grad, const S3dTrainBatch* batch,
                  int batch_size, int size, long n_qry, int n_slices, float dropout_p, unsigned long long seed,
                  int prec, const float* d_sdf_pred, const float* d_slices_rec, float d_vgg_loss, float grad_scale,
                  float* slices_rec, void* workspace, size_t workspace_bytes, void* stream);
/* Slices3DGTModel training step — reg_slices/train_gt.py:38-52 (train_step: zero_grad, model(batch), L1 on the
 * sdf, backward) without opt.step (s3d_adam_step).  Train-mode forward of model_gt.py:59-111: batch-statistics
 * BatchNorm in the VGG16-BN encoder over the B*n_slices slice images (running statistics updated in place through
 * enc->conv[i].bn[2..3], including conv[12]'s, whose output the model never uses), dropout p in the transformer.
 * batch->img is ignored (model_gt.py:75-76 concatenates it and drops the result).  Gradients are written to the
 * tensors of enc_grad / head_grad (same shapes as the parameters; conv[12].bn, the classifier and the unused
 * att_layer / fc_global never receive one, as in the reference).  losses_out[0] = L1(sdf_pred, sdf),
 * losses_out[1] = sign accuracy (train_gt.py:21-26).  sdf_pred_out (B,Q) optional. */
size_t s3d_gt_train_workspace_bytes(int batch, int size, long n_qry, int n_slices);
int s3d_gt_train_fwd_bwd(const S3dVgg16BnParams* enc, const S3dGtHeadParams* head,
                         const S3dVgg16BnParams* enc_grad, const S3dGtHeadParams* head_grad,
                         const S3dTrainBatch* batch, int batch_size, int size, long n_qry, int n_slices,
                         float dropout_p, unsigned long long seed, int prec, float* losses_out, float* sdf_pred_out,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The same Adam update for n_tensors parameters in one call (one kernel launch per 128 tensors).  params, offsets,
 * sizes: HOST arrays; tensor i takes its gradient from grad_flat[offsets[i] .. + sizes[i]) and keeps its moments at the
 * same range of exp_avg_flat / exp_avg_sq_flat (all three flat arrays on the device). */
int s3d_adam_step_multi(float* const* params, const long* offsets, const long* sizes, int n_tensors,
                        const float* grad_flat, float* exp_avg_flat, float* exp_avg_sq_flat, float lr, float beta1,
                        float beta2, float eps, int step, void* stream);
/* The dropout mask the kernels use: out[i] = keep(seed, site, idx0+i) ? 1/(1-p) : 0.  site = 4*layer +
 * {0 attention probabilities [(row*4 + head)*16 + key], 1 attention-block output [row*128 + c],
 *  2 FFN hidden [row*2048 + unit], 3 FFN output [row*128 + c]}; rows index the token tensor
 * [group][token][16 queries]; all four sites of the LAST layer index its compact token-0 rows [group][16 queries]
 * (only token 0 of that layer is computed). */
int s3d_dropout_mask(unsigned long long seed, int site, unsigned long long idx0, long n, float p, float* out,
                     void* stream);
/* torch.optim.Adam defaults semantics (no weight decay, no amsgrad); step counts from 1. */
int s3d_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                  float eps, int step, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement hook (no reference counterpart): when enabled, the launches of each kernel family are
 * bracketed with hipEvents on the caller's stream; s3d_prof_read waits for them and returns the summed
 * duration and the number of launches since s3d_prof_enable(1).  bench.py uses it to time the
 * dominant kernel inside its timed region.
 * ------------------------------------------------------------------------------------------- */
#define S3D_PROF_UNET 0          /* whole s3d_unet_encode_fwd */
#define S3D_PROF_LATENT 1        /* s3d_latent_build */
#define S3D_PROF_SAMPLE 2        /* sample_tokens_kernel */
#define S3D_PROF_ATTN 3          /* attn_layer_kernel (all layers) */
#define S3D_PROF_FFN 4           /* ffn_layer_kernel, full-row layers */
#define S3D_PROF_FFN_FINAL 5     /* ffn_layer_kernel, token-0 rows of the last layer (+fc_out) */
#define S3D_PROF_VGG 6           /* whole s3d_vgg_loss_fwd */
#define S3D_PROF_SAMPLE_PYR 7    /* sample_pyramid_kernel of s3d_sample_pyramid_fwd (without the locality sort) */
#define S3D_PROF_N 8
int s3d_prof_enable(int on);
int s3d_prof_read(int id, double* total_ms, long* count);
/* Trace ranges (tracing; no reference counterpart): the library brackets its stages and the phases of
 * the train step with roctx ranges ("s3d:...") that `rocprofv3 --marker-trace` records; the host side opens its own ranges
 * (the gradient exchange of trainer.py) through the same roctx library with these two calls.  No-ops (return 0) when no
 * roctx library can be loaded. */
int s3d_range_push(const char* name);
int s3d_range_pop(void);

/* ---------------------------------------------------------------------------------------------
 * Stand-alone ops of the module's helper API
 * ------------------------------------------------------------------------------------------- */
/* project_coord (models.py:28-36): coords (B,Q,3), trans (B,4,3) -> out (B,Q,2) */
int s3d_project_coord_fwd(const float* coords, const float* trans, float* out, int batch, long n_qry,
                          void* stream);
/* Image-space locality order of the queries (no reference counterpart: an execution-order choice of this
 * library, exposed so hosts / tests can map token-tensor rows back to queries).  Queries of each batch item are
 * projected exactly like models.py:28-36 (after models.py:53-60's flip / rotation), binned at 256^2 and ordered by
 * the Morton code of the bin, ascending query index inside a bin: perm_out[b*Q + slot] = query index.
 * Deterministic.  The decoder and the train step use this order internally for Q >= 4096. */
size_t s3d_query_sort_workspace_bytes(int batch, long n_qry);
int s3d_query_sort(const float* qry, const float* rot, const float* trans, int flip_yz, int batch, long n_qry,
                   int* perm_out, void* workspace, size_t workspace_bytes, void* stream);
/* sample_from_planes (models.py:38-46) on a channels-last plane: plane (N,H,W,C), grid (N,M,2)
 * -> out (N,M,C); bilinear, zeros padding, align_corners=True.  C % 4 == 0. */
int s3d_sample_planes_fwd(const float* plane, const float* grid, float* out, int n, int h, int w,
                          int c, long m, void* stream);
/* The reference's feature-sampling block as ONE op (models.py:63-73: five sample_from_planes calls over
 * feat_list + torch.cat(dim=2)): pyr = channels-last pyramid of B*n_slices images, grid (B,Q,2) projected
 * coordinates shared by the n_slices images of a batch item -> out (B*n_slices, Q, 992) fp32, channel order
 * [512 | 256 | 128 | 64 | 32].  HBM-bound on the 3 968-byte row writes (SURVEY.md 8(d) "stand-alone
 * feature-sample kernel").  workspace: s3d_sample_pyramid_workspace_bytes (point records + the locality order
 * the points are visited in for Q >= 4096; results do not depend on the order). */
size_t s3d_sample_pyramid_workspace_bytes(int batch, long n_qry);
int s3d_sample_pyramid_fwd(const S3dPyramid* pyr, const float* grid, float* out, int batch, int n_slices,
                           long n_qry, void* workspace, size_t workspace_bytes, void* stream);
/* layout helpers: (N,C,H,W) <-> (N,H,W,C) */
int s3d_nchw_to_nhwc(const float* in, float* out, int n, int c, int h, int w, void* stream);
int s3d_nhwc_to_nchw(const float* in, float* out, int n, int c, int h, int w, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Mesh extraction on the device — the step right after the hot path in
 * Generator3D.generate_from_latent / extract_mesh (reg_slices/reconstruct.py:148-243).
 *
 * MISE: replaces reg_slices/src_convonet/utils/libmise/mise.pyx:33-368 (MISE.query / update / to_dense).
 * The octree lives in a caller-owned device workspace; a round's points are returned as ascending linear
 * grid indices ((r*x + y)*r + z, r = resolution + 1) in a device buffer and never visit the host.  Same
 * point SET per round and same dense grid as the reference (its order within a round is insertion order).
 * The handle is a small host descriptor; s3d_mise_dev_query / _update synchronise the stream once each
 * (round count / error flag).  Host-side twin (bit-exact order): include/slice3d_mesh.h.
 * ------------------------------------------------------------------------------------------- */
size_t s3d_mise_dev_workspace_bytes(int resolution0, int depth);
void* s3d_mise_dev_create(void* workspace, size_t workspace_bytes, int resolution0, int depth,
                          double threshold, void* stream);            /* NULL on error (s3d_last_error) */
void s3d_mise_dev_destroy(void* handle);
int s3d_mise_dev_resolution(void* handle);                            /* resolution0 << depth */
/* unknown points of the next round -> idx_out[min(*n_out, capacity)] (device), *n_out (host) */
int s3d_mise_dev_query(void* handle, int* idx_out, long capacity, long* n_out, void* stream);
/* their coordinates as reconstruct.py:160-161 forms them in float32: box*(p/resolution - 0.5) -> qry_out (n,3) */
int s3d_mise_dev_points(void* handle, const int* idx, long n, float box, float* qry_out, void* stream);
/* mise.update(points, values) + one refinement step; values = the logits of the points (device) */
int s3d_mise_dev_update(void* handle, const int* idx, const float* values, long n, void* stream);
int s3d_mise_dev_update_f64(void* handle, const int* idx, const double* values, long n, void* stream);
/* mise.to_dense(): (r,r,r) float64, unknown entries forward-filled along x, then y, then z */
int s3d_mise_dev_to_dense(void* handle, double* out, void* stream);

/* Marching cubes: replaces libmcubes.marching_cubes (libmcubes/pywrapper.cpp:90-127 driving
 * marchingcubes.h:23-193) as classify -> scan -> emit.  Vertices and faces are bit-identical to the
 * reference's and in its order.  grid: device (nx,ny,nz) float32 (is_f64 = 0) or float64, C order; pad != 0
 * evaluates the grid as if np.pad(grid, 1, constant_values=pad_value) had been applied (reconstruct.py:189)
 * without materialising it.  Pass 1 returns the counts (one stream synchronisation), pass 2 — same arguments —
 * fills vertices (V,3) float64 in index units + 0.5 of the padded grid (libmcubes' convention, undone by the
 * caller: reconstruct.py:199-201) and triangles (F,3) int64, both device buffers. */
size_t s3d_mc_dev_workspace_bytes(int nx, int ny, int nz, int pad);
int s3d_mc_dev_count(const void* grid, int is_f64, int nx, int ny, int nz, int pad, double pad_value, double iso,
                     void* workspace, size_t workspace_bytes, long* n_vertices, long* n_triangles, void* stream);
int s3d_mc_dev_emit(const void* grid, int is_f64, int nx, int ny, int nz, int pad, double pad_value, double iso,
                    void* workspace, size_t workspace_bytes, double* vertices, long long* triangles, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dataset staging on the device — the per-sample tensor work of Slice3DDataset.__getitem__
 * (reg_slices/src/datasets.py:89-179) from pre-packed uint8 shards (slice3d_amd/shards.py: PNG decode, alpha
 * compositing datasets.py:73-88 and the PIL bilinear resize are done ONCE at pack time with the reference's own
 * operations, so the 13 PNG decodes per sample leave the loader's critical path).
 * ------------------------------------------------------------------------------------------- */
/* T.ToTensor() + T.Normalize(.5,.5) (datasets.py:31-34): imgs (B, 1+n_slices, S, S, 3) uint8 HWC (image 0 = input
 * view, then the slices in the order X1..X4, Z4..Z1, Y1..Y4, datasets.py:106-120) -> img_input (B,3,S,S) and
 * img_slices (B,3*n_slices,S,S) float32, bit-identical to the reference's tensors. */
int s3d_dataset_images_fwd(const unsigned char* imgs, float* img_input, float* img_slices, int batch,
                           int n_slices, int size, void* stream);
/* the query subset (datasets.py:153-165): pts (N,4) = (qry xyz, sdf) float32 as packed, idx (n) ->
 * qry (n,3), sdf (n), occ (n) = (sdf <= 0) (optional) */
int s3d_dataset_points_fwd(const float* pts, const int* idx, long n, float* qry, float* sdf, float* occ,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLICE3D_HIP_H */
