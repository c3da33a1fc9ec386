// decode.h — internal interface of the per-query decoder kernels (decode.hip).
#pragma once
#include "common.h"
#include "dropout.h"

#define S3D_GROUP 16          // queries per group = one MFMA row tile per token
#define S3D_FFN_CHUNK 32      // hidden units staged per LDS chunk
#define S3D_FFN_NCHUNK (S3D_FFN / S3D_FFN_CHUNK)

// offsets (in floats) into the packed head-weight image
struct HeadLayout {
    size_t fcp_w, fcp_b, fcs_b;
    size_t wproj[3];   // fc_s column blocks of pyramid levels 0..2 as [8][C_l/16] fragment images
    size_t wproj16[3]; // the same blocks as f16 hi/lo fragment pairs (split-precision mode)
    size_t ws34;       // fc_s columns 896..991 (levels 3,4) as an [8][6] fragment image
    size_t ws34_16;    // the same block as f16 hi/lo fragment pairs ([8][3] tiles of 32-deep k-steps)
    struct {
        size_t inw, inb, outw, outb, ln1g, ln1b;
        size_t w1, b1, w2, b2, ln2g, ln2b;   // w1: [128 tiles][8] image; w2: chunked [64][8][2] image
        size_t wf16;                         // f16 hi/lo chunk image of lin1/lin2 (64 x 32 KiB), decode_f16.hip
        size_t aq16;                         // in_proj / out_proj fragments of the query-major attention kernel
        size_t wfb16, aqb16;                 // the same two images with bf16 bit patterns (S3D_PREC_BF16)
    } L[S3D_N_LAYERS];
    size_t fco_w, fco_b;
    // token-0-only attention of the LAST layer in absorbed form (launch_attn_last_mix): fragment images (fp32 | f16
    // hi/lo) of M = [Wk_h^T Wq_h]_h (512x128) and N = [Wo_h Wv_h]_h (128x512), their bias vectors, dense scratch
    struct {
        size_t wm, wm16, bm, wn, wn16, bn, dense;
    } last;
    size_t total;
};
HeadLayout head_layout();

struct LayerPtrs {
    const float *inw, *inb, *outw, *outb, *ln1g, *ln1b, *w1, *b1, *w2, *b2, *ln2g, *ln2b, *wf16, *aq16, *wfb16, *aqb16;
};

struct SampleArgs {
    // latent
    const float* proj[3];
    const float* fine[2];
    int size;            // S
    int n_slices;        // T = n_slices + 1 tokens
    // head
    const float *fcp_w, *fcp_b, *fcs_b, *ws34;
    const float* ws34_16;   // f16 hi|lo fragment image of the same block: selects the split-precision form of the K = 96 product (NULL: fp32)
    // queries
    const float* qry;    // (B,Q,3) or NULL in grid mode
    const float* rot;    // (B,3,3) or NULL
    const float* trans;  // (B,4,3)
    int flip_yz;
    long n_qry;          // Q per batch item
    long groups_per_batch;
    long g_begin, g_count;   // group range handled by this launch (chunking)
    // grid mode (qry == NULL): coordinates box*linspace(-.5,.5,nx) generated from the query index
    int nx;
    float box;
    long q_offset;       // grid mode: linear grid index of this call's query 0 (slab decode; 0 = whole grid)
    float* X;            // out [g_count][T][16][128]
    const int* perm;     // optional: slot -> query index within the batch item (locality sort), (B, Q) ints
    float* raw_out;      // optional (training): sampled level-3/4 features [g_count][T][16][96], token-0 rows zero
    int lane_footprints; // 1: every folded level in the per-lane form (test / A-B switch; the shared-window form gives the same bits)
};

int launch_sample_tokens(const SampleArgs& a, hipStream_t stream);
int s3d_shared_footprint();   // api.hip: the process-wide switch behind s3d_decode_set_shared_footprint (1 = shared windows)
// X: [groups][T][16][128] in place;  if x0_out != NULL this is the LAST layer: only token 0 is
// produced, compactly, into x0_out [groups*16][128].
int launch_attn_layer(float* X, float* x0_out, long groups, int T, const LayerPtrs& w, hipStream_t stream);
// rows x 128 in place; if sdf_out != NULL: final layer, writes sign*(fc_out(LN2(..))) per row instead.
int launch_ffn_layer(float* X, long rows, const LayerPtrs& w, const float* fco_w, const float* fco_b,
                     float* sdf_out, float sign, long groups_per_batch, long n_qry, long g_begin, int prec,
                     const int* perm, hipStream_t stream, bool pre_ln1 = false);
// image-space locality sort of the queries of each batch item (counting sort on the Morton code of the
// projected pixel at 256^2): perm[b*Q + slot] = query index, ascending query index inside a bin (deterministic).
// ws: query_sort_ws_ints(B, Q) ints; on return ws[b*65536 + k] = end of bin k in perm[b] (bin/tile ranges).
size_t query_sort_ws_ints(int batch, long n_qry);

// ---- Slices3DGTModel sampler (model_gt.py:77-96): four fc_local[0]-folded 128-channel maps (pyramid levels
// conv2_2 .. conv5_3) + the raw 64-channel conv1_2 level through a K=64 MFMA, + bias, ReLU -> slice-token rows.
// Token-0 rows are written by launch_gt_point_tokens (pts_feat_extractor) after the second fc_local layer.
struct SampleGtArgs {
    const float* proj[4];   // [0] = conv5_3 level (S/16) ... [3] = conv2_2 level (S/2); (n_img, W, W, 128)
    const float* fine;      // conv1_2 level (n_img, S, S, 64)
    const float* wraw;      // fragment image of fc_local[0].weight[:, 0:64]  [8 j][4 u][64 lanes][4]
    const float* wraw16;    // the same block as f16 hi|lo fragment pairs ([8][2] tiles): split-precision form of the product (NULL: fp32)
    const float* bias;      // fc_local[0].bias
    int size, n_slices;
    const float *qry, *rot, *trans;
    int flip_yz;
    long n_qry, groups_per_batch, g_begin, g_count;
    int nx;
    float box;
    float* X;               // out [g_count][T][16][128]; token-0 rows are zero-filled
    const int* perm;
    float* raw_out;         // optional (training): sampled conv1_2 features [g_count][T][16][64], token-0 rows zero
};
int launch_sample_tokens_gt(const SampleGtArgs& a, hipStream_t stream);
struct GtPointArgs {
    const float *w0, *b0, *w1, *b1, *w2, *b2;   // pts_feat_extractor Linear(3,32), (32,64), (64,128), each + ReLU
    const float *qry, *rot;
    int flip_yz, n_slices;
    long n_qry, groups_per_batch, g_begin, g_count;
    int nx;
    float box;
    float* X;
    const int* perm;
    float *h1_out, *h2_out;  // optional (training): hidden activations per query row, [rows][32] and [rows][64]
};
int launch_gt_point_tokens(const GtPointArgs& a, hipStream_t stream);
// point records (gx, gy, original index) in visiting order, then the gather / row-store kernel over them
int launch_sample_pyramid_points(const float* grid, const int* perm, float* pts, int batch, long n_qry,
                                 hipStream_t stream);
int launch_sample_pyramid(const float* const* level, const float* grid, const int* perm, float* pts, float* out,
                          int batch, int n_slices, int size, long n_qry, hipStream_t stream);
int launch_query_sort(const float* qry, const float* rot, const float* trans, int flip_yz, int batch, long n_qry,
                      int* perm, int* ws, hipStream_t stream);
// training forward: y = LN2(u), u = x + FFN(x) with x read from Xin, y -> Yout, u -> Uout (pre-LN, saved)
int launch_ffn_layer_train(const float* Xin, float* Yout, float* Uout, long rows, const LayerPtrs& w,
                           const DropCfg& drop_hidden, const DropCfg& drop_out, hipStream_t stream);

int launch_project_coord(const float* coords, const float* trans, float* out, int batch, long n_qry,
                         hipStream_t stream);
int launch_sample_planes(const float* plane, const float* grid, float* out, int n, int h, int w, int c, long m,
                         hipStream_t stream);

// decode_f16.hip
// Operand images of a row tensor x [P][128] for the FFN weight-gradient kernel (ffn_wgrad_rec_kernel, train.hip): per 32-row
// block 16 KiB each, f16 hi | lo in the consumer's LDS byte order, rows past the end zero.
//   R image   [hi|lo][32 rows][128 ch] halfs: the 16-byte chunk s (channels 8s..8s+7) of row r sits at chunk s ^ (r & 15)
//   D^T image [hi|lo][128 ch][32 slots] halfs: slot 8g+t <-> row 4g+t (t < 4) / 16+4g+t-4, chunk g of channel q at
//             chunk g ^ perm[(q >> 2) & 3], perm = {0, 2, 3, 1}
// The pipelined FFN kernel writes them for the rows it holds anyway (the layer input x in the training forward, dY in the
// backward data pass): a wave's two 16-row tiles ARE one 32-row block, the R chunks are its split fragments as they
// are, and the D^T chunks come out of the MFMA D layout of a transposing product (x tile times a 0/1 selector).
#define FWR_BLK_HALFS 8192   // one image of one 32-row block: hi 4096 halfs | lo 4096 halfs = 16 KiB
__host__ __device__ inline int fwr_dperm(int q) { return (0x1320 >> (4 * ((q >> 2) & 3))) & 3; }   // {0,2,3,1}
static inline size_t ffn_rec_image_floats(long P) { return (size_t)((P + 31) / 32) * 4096; }
// Activity bits of the FFN hidden units (post-dropout h > 0), written by the training forward, read by the backward data
// pass and both weight-gradient contractions.  One dword holds 32 units of one row: the four 32-unit chunks k = 0..3 of
// group G (hidden units 128 G .. 128 G + 127), D tiles a = 0, 1, the lane quad g' — unit 128 G + 32 k + 16 a + 4 g' + i is
// bit 4 k + a + FFN_MASK_POS(i): the forward takes the bits of a tile's four units from its two packed f16 pairs at once
// (bits 0 / 16 of a pair, the second pair shifted by 2).  Dwords are laid out so that a wave's store of one (16-row tile,
// group) is 256 contiguous bytes: [tile = row >> 4][G][g'][row & 15]  (per-row layouts left every 64-byte line open for the
// whole kernel: 16 dword-sized writes, evicted half filled).
#define FFN_MASK_POS(i) ((i) == 0 ? 0 : (i) == 1 ? 16 : (i) == 2 ? 2 : 18)
__host__ __device__ inline size_t ffn_mask_dword(long row, int G, int gq) {
    return ((size_t)(row >> 4) * 16 + G) * 64 + gq * 16 + (row & 15);
}
static inline size_t ffn_mask_dwords(long P) { return (size_t)((P + 31) / 32) * 32 * 64; }
// imgd / imgr (optional, ffn_rec_image_floats(rows) floats each): D^T / R image of DY
int launch_ffn_bwd_dx_f16x3(const float* DY, const float* Dres, const unsigned* M, float* DX, long rows,
                            const float* timg, float gate_scale, hipStream_t stream, float* imgd = nullptr,
                            float* imgr = nullptr, const DropCfg* dy_mask = nullptr, bool single = false);   // single: hi * hi products only (S3D_PREC_F16 training throughput mode)
int launch_pack_ffn_f16x3_bwd(const float* w1, const float* w2, float* out, hipStream_t stream);
// imgd / imgr (optional): D^T / R image of Xin
int launch_ffn_layer_train_f16x3(const float* Xin, float* Yout, float* Uout, unsigned* Mout, long rows,
                                 const LayerPtrs& w, const DropCfg& drop_hidden, const DropCfg& drop_out,
                                 hipStream_t stream, float* imgd = nullptr, float* imgr = nullptr, bool single = false);
int launch_attn_layer_q(float* X, long groups, int T, const LayerPtrs& w, hipStream_t stream, bool single_pass = false,
                        bool bf16 = false);
// training forward of the attention block, query-major (decode_attnq.hip): y = LN1(u), u = xin + dropout1(out_proj(MHA(xin)) + b),
// o = MHA output before out_proj; nothing else is kept (the fused backward of train_attnq.hip recomputes Q / K / V)
int launch_attn_layer_q_train(const float* xin, float* y, float* u, float* o, long groups, int T, const LayerPtrs& w,
                              const DropCfg& d0, const DropCfg& d1, hipStream_t stream, bool single = false);
// Last layer, token 0 only (models.py:83 consumes nothing else), with the projections absorbed.  Per head h, with
// q = Wq_h x0 + bq_h:   score_h[t] = q . (Wk_h x_t + bk_h) = (M_h x0 + m_h) . x_t + const   (const cancels in the softmax)
//                       M_h = Wk_h^T Wq_h (128x128),  m_h = Wk_h^T bq_h
// and, since the probabilities sum to 1:   out_proj(concat_h(Wv_h xbar_h + bv_h)) = sum_h N_h xbar_h + n
//                       xbar_h = sum_t p_h[t] x_t,  N_h = Wo[:, head h] Wv_h (128x128),  n = Wo bv + bo
// so the 13 tokens are never projected: per query two dense (128 <-> 512) GEMMs + 2 x 4 x 13 x 128 MACs here.
//   qt   [rows0][512]: per head the 128-vector M_h x0 + m_h
//   xbar [rows0][512]: per head sum_t softmax_t(qt_h . x_t / sqrt(32)) x_t
int launch_attn_last_mix(const float* X, const float* qt, float* xbar, long groups, int T, hipStream_t stream);
// decode_last.hip: the whole block (x0 -> qt -> mixing -> u = N xbar + n + x0) in one kernel, split-precision modes only.
// X [groups][T][16][128] -> U [groups*16][128] (pre-LayerNorm1 sums); wm16 / wn16: the f16 hi|lo fragment images of M / N
// (HeadLayout::last), bm / bn their bias vectors.  Bit-identical to tok0_copy + row GEMM + launch_attn_last_mix + row GEMM.
int launch_attn_last_fused(const float* X, float* U, long groups, int T, const float* wm16, const float* bm,
                           const float* wn16, const float* bn, bool single_pass, hipStream_t stream);
// the absorbed matrices from in_proj_weight (384,128), in_proj_bias (384), out_proj.weight (128,128), out_proj.bias:
//   kind 0: out [512][128] = M (row h*128 + c), bias_out [512] = m
//   kind 1: out [128][512] = N (column h*128 + c), bias_out [128] = n
int launch_absorb_last(const float* in_w, const float* in_b, const float* out_w, const float* out_b, float* out,
                       float* bias_out, int kind, hipStream_t stream);
int launch_pack_attn_q_f16x3(const float* win, const float* wout, float* out, hipStream_t stream, int bf16 = 0);
int launch_ffn_layer_f16x3(float* X, long rows, const LayerPtrs& w, const float* wimg, const float* fco_w,
                           const float* fco_b, float* sdf_out, float sign, long groups_per_batch, long n_qry,
                           long g_begin, const int* perm, hipStream_t stream, bool single_pass = false,
                           bool pre_ln1 = false, bool bf16 = false);   // pre_ln1 (final layer only): the rows are pre-LayerNorm1 sums, normalised in the prologue
int launch_pack_ffn_f16x3(const float* w1, const float* w2, float* out, hipStream_t stream, int bf16 = 0);
