// train.h — internal interface of the backward / train-mode kernels (train.hip).
#pragma once
#include "conv.h"
#include "decode.h"

// ---------------------------------------------------------------------------------------------
// Weight gradient of a convolution / linear layer:
//   dW[n][(tap, c)] (+)= sum over output pixels p of dY[p][n] * X[pixel(p, tap)][c]
// (contraction over PIXELS on fp32 MFMA; partial sums per pixel split, then a deterministic reduce
//  that writes straight into the PyTorch parameter layout selected by `out_kind`).
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* dy;      // (Nimg, H, W, dy_cstride) NHWC; channels [dy_coff, dy_coff + N) are used
    int dy_cstride, dy_coff, N;
    ConvSrc x;            // activation source (same addressing rules as the forward conv)
    int x_coff, Cx;       // channels [x_coff, x_coff + Cx) of the source are this K segment
    int Nimg, H, W;       // dY grid
    int ks, stride, Hin, Win;
    // output mapping (same index formulas as the packers):
    float* out;
    int out_kind;         // S3D_PACK_LINEAR: out[n*ld + k]; S3D_PACK_CONV: out[((n*cin_tot+cin_begin+c)*taps)+tap];
                          // S3D_PACK_CONVT: rows n = ci, k = (q, co): out[(n*ct + co)*4 + q]
    int ld, cin_tot, cin_begin, ct;
    int accumulate;       // 1: out += result
    int prec;             // S3D_PREC_F16X3: plain 1x1 contractions run on split-precision f16 MFMA (else fp32 MFMA)
    float* partial;       // workspace
    size_t partial_floats;
    float* bias_out;      // null, or [N]: column sums of dY (the layer's bias gradient; `accumulate` applies).  The
                          // split-precision linear kernel adds them up while it converts dY (no second pass over dY);
                          // the other kernels run launch_colsum with `bias_partial` as its workspace
    float* bias_partial;
    int single;           // with prec = S3D_PREC_F16X3: only the hi * hi products (S3D_PREC_F16 training throughput mode; linear kernel only)
};
int launch_wgrad(const WgradArgs& a, hipStream_t stream);

// ---- FFN weight gradients with on-the-fly recomputation of the 2048-wide operand (split precision) ----
// out[q][hid] = sum_rows D[row][q] * Z[row][hid],  Z[row][hid] = bit(row,hid) ? (R[row] . W[hid] + bias[hid]) * scale : 0
//   dW2      = ( D = dY,  R = x,  W = lin1 weight, bias = lin1 bias )   out -> lin2 grad [128][2048]
//   dW1^T    = ( D = x,   R = dY, W = lin2 weight^T, no bias )          out -> lin1 grad [2048][128] (transposed store),
//              colsum(Z) -> lin1 bias grad
// bits: activity mask written by the forward FFN kernel.  The [rows][2048] matrices are never materialised.
struct FfnWgradArgs {
    const float* Dimg;      // transposed f16 hi/lo image of D [P][128]   (written by the pipelined FFN kernel, decode.h)
    const float* Rimg;      // row-major  f16 hi/lo image of R [P][128]
    const float* wimg;      // f16 hi|lo fragment image of W [2048][128] (launch_pack_ffn_rec_f16x3)
    const float* bias;      // [2048] or null
    const unsigned* mask;   // [P][64] dwords
    float scale;            // value of a kept unit's dropout factor (1/(1-p)), 1 without dropout
    long P;
    float* out;             // [128][2048] (transpose_out = 0) or [2048][128] (transpose_out = 1)
    int transpose_out;
    float* bias_out;        // [2048] column sums of Z, or null
    int accumulate;
    float* partial;         // workspace
    size_t partial_floats;
    int single;             // 1: single-pass f16 (S3D_PREC_F16 throughput mode of the training step): hi * hi products only
};
int launch_ffn_wgrad_rec(const FfnWgradArgs& a, hipStream_t stream);
// W: element (hid, k) at w[hid*sh + k*sk]  ->  image [128 hid tiles][4][64 lanes][8] halfs hi, then lo
int launch_pack_ffn_rec_f16x3(const float* w, int sh, int sk, float* out, hipStream_t stream);
#define S3D_FFN_REC_IMG_FLOATS (S3D_FFN * 128)   /* 2 x 2048*128 halfs */
size_t wgrad_partial_floats(long P, int N, int Ktot);

// column sums over rows/pixels: out[c] (+)= sum_p in[p*cstride + coff + c]
int launch_colsum(const float* in, long P, int cstride, int coff, int C, float* out, int accumulate,
                  float* partial, hipStream_t stream);

// ---- BatchNorm2d, train mode (NHWC, statistics over all pixels of all images) ----
// sync (data-parallel --sync_bn): statistics over the batches of ALL ranks.  The forward all-reduces per channel
// [mean_r, var_r + mean_r^2] (equal element counts per rank), the backward [sum g, sum g*xhat]; the collective is the
// host's (S3dSyncBn callback: RCCL through torch.distributed), issued on the launch stream between two kernels.
typedef S3dSyncBn BnSync;
// stats: mean[c], rstd[c] (biased variance), and the running-stat update torch does (momentum .1, unbiased)
int launch_bn_stats(const float* z, long P, int C, float* mean, float* rstd, float* running_mean,
                    float* running_var, float* partial, hipStream_t stream, const BnSync* sync = nullptr);
// y = relu(gamma*(z-mean)*rstd + beta)            (pool = 0)
// y = maxpool2x2(relu(...))                       (pool = 1; z is (N,H,W,C), y is (N,H/2,W/2,C))
int launch_bn_apply(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    float* y, int n, int h, int w, int c, int pool, hipStream_t stream);
// backward of y = relu(bn(z)) [+ maxpool]: given dy (grid of y), writes dz (grid of z), dgamma, dbeta.
// 3x3 / pad 1 conv of an NCHW image with 1 or 3 channels to 64 NHWC channels + bias, exact fp32 (first VGG conv)
// optional epilogue (inference): y = acc * scale[c] + shift[c] (folded BatchNorm; bias then NULL), ReLU
int launch_conv3x3_first(const float* img_nchw, int cin, const float* w_oihw, const float* bias, float* out_nhwc,
                         int n, int h, int w, hipStream_t stream, const float* scale = nullptr,
                         const float* shift = nullptr, int relu = 0);
int launch_bn_bwd(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                  const float* dy, float* dz, float* dgamma, float* dbeta, int n, int h, int w, int c, int pool,
                  float* partial, hipStream_t stream, const float* add = nullptr,    // dz += add (same grid) if set
                  const BnSync* sync = nullptr);

// ---- misc elementwise ----
int launch_axpy(float* y, const float* x, float alpha, long n, hipStream_t stream);            // y += alpha*x
int launch_slice_sum(const float* in, float* out, int batch, int ns, long per_img, int accumulate,
                     hipStream_t stream);                                                      // out[b] (+)= sum_s in[b*ns+s]
int launch_tanh_bwd(const float* y_nchw, const float* dy_nchw, float* dz_nhwc, int n, int c, int h, int w, int cpad,
                    hipStream_t stream);   // dz[n,y,x,c] = dy*(1-y^2), NCHW -> NHWC(cpad), zero padded
// L1 loss: loss_acc[0] += scale*sum|a-b|; grad[i] (+)= scale*sign(a-b)
// grad_mul: extra factor on the gradient only (the backward scale of the split-precision training path)
int launch_l1_fwd_bwd(const float* a, const float* b, long n, float scale, float* grad, int accumulate_grad,
                      float* partial, float* loss_acc, hipStream_t stream, float grad_mul = 1.f, int relu_mask = 0);
#define S3D_L1_PARTIAL_FLOATS 2048   /* `partial` holds one float per block */
// in-place multiply of up to S3D_SCALE_TABLE_MAX tensors by one factor (one launch)
#define S3D_SCALE_TABLE_MAX 224
struct ScaleTable {
    float* p[S3D_SCALE_TABLE_MAX];
    long n[S3D_SCALE_TABLE_MAX];
    int count;
    void add(const float* q, long k) {
        if (q && k > 0 && count < S3D_SCALE_TABLE_MAX) {
            p[count] = const_cast<float*>(q);
            n[count] = k;
            ++count;
        }
    }
};
int launch_scale_table(const ScaleTable& t, float s, hipStream_t stream);
// ---- last layer in training, absorbed token-0 form (train2.hip) ----
#define S3D_ABS_NA 544   /* xbar row: 4 x 128 mixed rows | 4 probability sums | zeros */
int launch_attn_mix0_fwd(const float* X, const float* qt, float* xbar, long groups, int T, const DropCfg& drop, hipStream_t stream);
int launch_attn_mix0_bwd(const float* X, const float* qt, const float* dxbar, float* dX, float* dqt, long groups, int T,
                         const DropCfg& drop, hipStream_t stream);
int launch_absorb_train(const float* in_w, const float* in_b, const float* out_w, float* M, float* mvec, float* Naug,
                        hipStream_t stream);
int launch_absorb_grad(const float* in_w, const float* in_b, const float* out_w, const float* dM, const float* dm,
                       const float* dNaug, float* g_in_w, float* g_in_b, float* g_out_w, hipStream_t stream);
int launch_relu_mask_bwd(const float* y, float* dy, long n, hipStream_t stream);              // dy *= (y > 0)
int launch_pool_bwd(const float* y, const float* dyp, float* dy, int n, int h, int w, int c, hipStream_t stream,
                    int relu_mask = 0);   // relu_mask: dy *= (y > 0) on the way out
int launch_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                float bc1, float bc2, hipStream_t stream);
// one launch for up to S3D_ADAM_TABLE_MAX tensors: parameter i is updated from g / m / v [off[i], off[i] + n[i])
#define S3D_ADAM_TABLE_MAX 128
struct AdamTable {
    float* p[S3D_ADAM_TABLE_MAX];
    long off[S3D_ADAM_TABLE_MAX];
    long n[S3D_ADAM_TABLE_MAX];
    int count;
};
int launch_adam_table(const AdamTable& t, const float* g, float* m, float* v, float lr, float b1, float b2, float eps,
                      float bc1, float bc2, hipStream_t stream);

// ---- decoder backward pieces ----
// LayerNorm backward over rows of 128: du = LN'(u; gamma) dy ; dgamma/dbeta (+)= column sums
// optional third output: dum = du * dropout mask (drop; dum == du allowed when drop->p == 0) and dsum[128] = its column sums
int launch_ln_bwd(const float* u, const float* gamma, const float* dy, float* du, long rows, float* dgamma,
                  float* dbeta, int accumulate, float* partial, hipStream_t stream, const DropCfg* drop = nullptr,
                  float* dum = nullptr, float* dsum = nullptr);
int launch_ln_fwd(const float* u, const float* gamma, const float* beta, float* y, long rows, hipStream_t stream);
// attention core on stored QKV [groups][T][16][384] (q | k | v, 4 heads x 32): O [groups][T][16][128]
int launch_attn_core_fwd(const float* qkv, float* o, long groups, int T, const DropCfg& drop, hipStream_t stream);
int launch_attn_core_bwd(const float* qkv, const float* d_o, float* dqkv, long groups, int T, const DropCfg& drop,
                         hipStream_t stream);
// d(hidden) *= (a > 0) ; a <- relu(a)   (FFN backward on a row block)
int launch_relu_bwd_inplace(float* a, float* dh, long n, long row0, const DropCfg& drop, hipStream_t stream);
// out[i] = in[i] * mask(idx0 + i)   (out may alias in)
int launch_dropout_apply(const float* in, float* out, long n, unsigned long long idx0, const DropCfg& drop,
                         hipStream_t stream);

// ---- train2.hip ----
struct SampleBwdArgs {
    const float* dX;         // [groups][T][16][128]
    float* dproj[3];         // zero-initialised by the caller; (n_img, H_l, W_l, 128)
    float* dfine[2];         // d pyramid levels 3, 4 (accumulated into)
    const float* ws34_t;     // fragment image of Ws34^T: [6][8] tiles (LINEAR_T pack of fc_s[:, 896:992])
    const float* ws34_t16;   // the same matrix as f16 hi|lo fragment pairs ([6][4]): split-precision form of the product in the tiled
                             // kernel (NULL: fp32)
    const float *qry, *rot, *trans;
    int flip_yz, size, n_slices;
    long n_qry, groups_per_batch, groups;
    const int* perm;         // optional: token rows are in sorted order, perm[b*Q + slot] = query (s3d_query_sort)
    const int* bin_ends;     // with perm: [B][65536] end offset of every Morton bin (launch_query_sort's ws)
    float* gxy;              // optional scratch, B * n_qry * 2 floats (with perm): the projected image coordinates of every
                             // SORTED slot, filled by a pre-pass of launch_sample_bwd — the tiled kernel then loads a slot's
                             // (gx, gy) with one coalesced read instead of walking perm -> qry -> rotate -> project (three
                             // dependent global loads) 2 x n_slices times per query
    float* partial;          // optional scratch of sample_bwd_partial_floats(size, B, n_slices) floats (with perm, Slices3DRegModel
                             // levels): the atomic-free form (train_sbd.hip) — every tile writes its footprints to its own slot and
                             // a second kernel adds the slots into the maps in a fixed order: bit-reproducible gradients
    int gt;                  // Slices3DGTModel levels: dproj[0..2] + dfine[0] are the four folded 128-ch maps
                             // (S/16 ... S/2), dfine[1] the raw 64-ch conv1_2 map, ws34_t the [4][8] image of Wraw^T
};
int launch_sample_bwd(const SampleBwdArgs& a, hipStream_t stream);
// train_sbd.hip: floats of SampleBwdArgs::partial (0: this size runs the atomic kernels); 1 = launched, 0 = not covered, < 0 error
size_t sample_bwd_partial_floats(int S, long batch, int n_slices);
bool sample_bwd_dense_covers(const SampleBwdArgs& a);
int launch_sample_bwd_dense(const SampleBwdArgs& a, hipStream_t stream);
int launch_tok0_copy(float* full, float* compact, long groups, int T, int dir, int width, hipStream_t stream);
// perm (optional): row slot -> query index within the batch item (sorted token order)
int launch_fc_out_fwd(const float* x, const float* w, const float* b, float* sdf, long rows, long gpb, long n_qry,
                      const int* perm, hipStream_t stream);
int launch_fc_out_bwd(const float* x, const float* w, const float* dsdf, float* dx, float* t, long rows, long gpb,
                      long n_qry, const int* perm, hipStream_t stream);
int launch_scalar_reduce(const float* a, const float* b, long n, int mode, float scale, float* out, int accumulate,
                         float* partial, hipStream_t stream);
int launch_vgg_prep_bwd(const float* din16, const float* stdv, float* drec, int n_img, int size, hipStream_t stream);
// Z (n_img, S, S, 32): channel ci*9 + tap of the 1x1 product dY w (27 valid) -> drec (n_img, 3, S, S) NCHW, accumulated
int launch_vgg_first_bwd(const float* Z, const float* stdv, float* drec, int n_img, int size, hipStream_t stream);
int launch_qry_rot_rows(const float* qry, const float* rot, int flip_yz, long n_qry, long gpb, long groups,
                        const int* perm,
                        float* out, hipStream_t stream);
int launch_copy_cols(const float* src, float* dst, int rows, int csrc, int cdst, hipStream_t stream);
// fused attention-core backward with Q / K / V recomputed on chip (train_attnq.hip): x, dO (rows x 128) -> dQKV (rows x 384)
struct LayerPtrs;
int launch_attn_bwd_q(const float* x, const float* d_o, float* dqkv, long groups, int T, const LayerPtrs& w,
                      const DropCfg& d0, hipStream_t stream, bool single = false);
int launch_emb_grad(const float* dF0, const float* w, float* demds, int B, int ns, int npix, hipStream_t stream);
#define CS_CHUNKS_MAX 512
