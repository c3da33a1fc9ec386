// train_sbd.hip — pyramid-sampling backward of Slices3DRegModel WITHOUT float atomics (round 6): the same gradients from the same
// inputs bit for bit, run after run.  Replaces sample_bwd_tiled_kernel<0> (train2.hip) when the caller hands a partial-sum scratch
// (SampleBwdArgs::partial); the reference operation is the backward of the five F.grid_sample calls + fc_s of
// /root/reference/reg_slices/src/models.py:69-81 (bilinear, align_corners=True).
//
// One workgroup per (object, slice image, 16 x 16-bin tile of the locality sort), eight waves.  ALL five levels are dense products
// on the fp32 MFMA with the accumulators stationary in registers:
//     P_l^T[c][pix] = sum_q dX[q][c] * w_l[q][pix]          (c = 128 token channels, pix = a pixel of the tile's footprint at level l)
// wave j owns channels 16j .. 16j+15 of every footprint pixel; a k-step is four queries (A[c][q] = the staged dX rows of the chunk,
// B[q][pix] = the query's bilinear weight on that pixel).  The three folded levels (3 x 3 / 4 x 4 / 6 x 6 pixels at 256^2) are five
// 16-pixel tiles that every k-step visits.  The two RAW levels (S/2: 64 channels, S: 32 channels; footprints 10 x 10 and 18 x 18)
// are cut into 4 x 4-pixel blocks — 3 x 3 and 5 x 5 of them — and a k-step visits only the rows x columns of blocks its four
// queries touch (two wave-uniform bit masks from the staging pass: sorted neighbours share their blocks, ~6 of 34 per step).
// Per chunk of 64 queries: the dX rows arrive by LDS-DMA one chunk ahead (two buffers); five waves stage, per (level, query), the
// query's weight on every footprint pixel (folded) or footprint column | row (raw: a block pixel's weight is column x row) as LDS
// tables, so that a k-step is LDS reads + MFMAs; the next step's operands are read while this step's MFMAs run.
// The raw levels' projection through W_raw^T (fc_s[:, 896:992]) is applied AFTERWARDS, once per footprint pixel instead of once per
// query: d raw_l[pix][c'] = sum_c P_l[pix][c] W^T[c][c'] — the accumulators go through LDS (a pixel's 128 channels are spread over
// the eight waves) and come back as rows of the same product the per-query form ran (f16 hi/lo x 3 on the f16 MFMA, or fp32).
// The round-5 kernel added 384 values per (query, slice) to LDS with ds_add_f32 (~3 cycles per lane: 3/4 of its 8.8 ms).
//
// No global atomics either: a workgroup writes its footprints to ITS slot of the partial scratch ([object][slice][tile][G.ptotal]
// floats) and sbd_reduce_kernel adds, for every pixel of the five gradient maps, the slots of the tiles whose footprint holds it, in
// ascending tile order.  (A tap outside its tile's footprint cannot occur for the sizes launch_sample_bwd accepts — the tile's bins
// ARE level-4 pixels and the coarser footprints' borders never fall on integers, sbd_geom — but the staging pass still handles it:
// such a query's row goes to the maps directly with global atomics, S3D_SBD_SLOW_MOD forces that path for tests.)
// Measured (4 objects x 100 k queries x 12 slices at 256^2): 4.4 ms + 0.8 ms for the reduce pass against 8.85 ms; cycle stamps
// (-DSBD_STAMPS, tools/r06_sbd_stamps.sh): 71 % of a workgroup's time in the k-steps, whose raw-level part is bound by its scalar
// bit tests and taken branches (39 conditional MFMA sites), not by the matrix pipe (29 % busy) or LDS.
#include "train.h"

#define SBD_THREADS 512
#define SBD_CHUNK 64          // queries per staged chunk = 16 k-steps
#define SBD_PROW 132          // floats per staged accumulator row of the projection epilogue
#define SBD_NB3 3             // 4 x 4-pixel blocks per axis, raw level 3 (footprint <= 12)
#define SBD_NB4 5             //                               raw level 4 (footprint <= 20)
#define SBD_NBLK (SBD_NB3 * SBD_NB3 + SBD_NB4 * SBD_NB4)
#define SBD_NACC (5 + SBD_NBLK)
#define SBD_XFLOATS (8 * 16 * SBD_PROW)   // staging region: two chunk buffers of SBD_CHUNK * 128 floats, or the epilogue's eight blocks
#define SBD_FROW 80           // floats per query of the folded weight table: 16 | 16 | 48 footprint pixels (five 16-pixel tiles)
#define SBD_RROW 72           // floats per query of the raw weight table: level 3 columns 12 | rows 12, level 4 columns 20 | rows 20
#define SBD_LDS_FLOATS (6 * 8 * 256 + SBD_XFLOATS + SBD_CHUNK * (SBD_FROW + SBD_RROW) + 32)

struct SbdGeom {
    int W[5], fw[5], cov[5], poff[5];   // level width, tight footprint width, row width of the slot's footprint image, float offset
    int ptotal;                         // floats per (object, slice, tile) slot
    int slow_mod;                       // test hook: sorted slots with qs % slow_mod == 0 take the atomic path
};
__host__ __device__ static constexpr int sbd_mt0(int l) { return l == 0 ? 0 : l == 1 ? 1 : 2; }
__host__ __device__ static constexpr int sbd_nmt(int l) { return l < 2 ? 1 : 3; }
__host__ __device__ static constexpr int sbd_C(int l) { return l < 3 ? 128 : (l == 3 ? 64 : 32); }

__device__ __forceinline__ unsigned sbd_compact8(unsigned v) {
    v &= 0x5555u;
    v = (v | (v >> 1)) & 0x3333u;
    v = (v | (v >> 2)) & 0x0F0Fu;
    v = (v | (v >> 4)) & 0x00FFu;
    return v;
}
__device__ __forceinline__ unsigned sbd_spread8(unsigned v) {
    v = (v | (v << 4)) & 0x0F0Fu;
    v = (v | (v << 2)) & 0x3333u;
    v = (v | (v << 1)) & 0x5555u;
    return v;
}
// element (k, c') of W_raw^T in the fp32 fragment image (tile u = c' / 16 of the level's first tile u0; common.h's swapped form)
__device__ __forceinline__ float sbd_wraw(const float* img, int u0, int k, int c) {
    const int u = u0 + (c >> 4), mm = c & 15;
    return img[((u * 8 + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + mm) * 4 + (k & 3)];
}

// d raw[pix][c'] of one 4 x 4-pixel block: rows = the block's 16 pixels (staged accumulators, SBD_PROW floats apart)
template <bool F16, int L>
__device__ __forceinline__ void sbd_project_block(const float* __restrict__ rows, const float* __restrict__ s_wt, float* __restrict__ out,
                                                  int cov, int fw, int bx, int by, int lane) {
    constexpr int C = sbd_C(L), NT = C / 16, U0 = L == 3 ? 0 : 4;
    const int m = lane & 15, g = lane >> 4;
    const float* dp = rows + m * SBD_PROW + (F16 ? 8 : 4) * g;
    f32x4 dt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dt[j] = ld4(dp + (F16 ? 32 * (j >> 1) + 4 * (j & 1) : 16 * j));
    f32x4 draw[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) draw[u] = zero4();
    if (F16) {
        const _Float16* sw = reinterpret_cast<const _Float16*>(s_wt);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {   // k-slot 8g + t of step kk <-> token channel 32 kk + 8g + t
            const float x8[8] = {dt[2 * kk][0], dt[2 * kk][1], dt[2 * kk][2], dt[2 * kk][3],
                                 dt[2 * kk + 1][0], dt[2 * kk + 1][1], dt[2 * kk + 1][2], dt[2 * kk + 1][3]};
            s3d_half8 bh, bl;
            s3d_split8(x8, bh, bl);
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const s3d_half8 fh = *reinterpret_cast<const s3d_half8*>(sw + ((U0 + u) * 4 + kk) * 1024 + lane * 8);
                const s3d_half8 fl = *reinterpret_cast<const s3d_half8*>(sw + ((U0 + u) * 4 + kk) * 1024 + 512 + lane * 8);
                draw[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, bl, draw[u], 0, 0, 0);
                draw[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, bh, draw[u], 0, 0, 0);
                draw[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, bh, draw[u], 0, 0, 0);
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) draw[u] = mfma4(ld4(s_wt + (((U0 + u) * 8 + j) * 64 + lane) * 4), dt[j], draw[u]);
    }
    // (pixels of the blocks beyond the footprint's fw x fw never receive weight and are never read)
    if (4 * by + (m >> 2) >= fw || 4 * bx + (m & 3) >= fw) return;
    float* o = out + ((long)(4 * by + (m >> 2)) * cov + 4 * bx + (m & 3)) * C + 4 * g;
#pragma unroll
    for (int u = 0; u < NT; ++u) st4(o + 16 * u, draw[u]);
}

// operands of one k-step (four queries, query g in lane group g): the staged dX value, the weight on the lane's pixel of the five
// folded tiles, and for the raw levels the weights on the lane's column / row of every block column / row the step's masks name
struct SbdOps {
    float xa, wb[5], wx3[SBD_NB3], wy3[SBD_NB3], wx4[SBD_NB4], wy4[SBD_NB4];
    int m3, m4;
};
__device__ __forceinline__ void sbd_load_ops(SbdOps& o, const float* __restrict__ s_x, const float* __restrict__ s_fw,
                                             const float* __restrict__ s_rw, int st, int g, int m, int wave, int m3, int m4) {
    const int qq = 4 * st + g;
    o.xa = s_x[qq * 128 + 16 * wave + m];   // (the four queries' rows share their banks: one 4-way conflict per step)
    const float* f = s_fw + qq * SBD_FROW + m;
#pragma unroll
    for (int t = 0; t < 5; ++t) o.wb[t] = f[16 * t];
    const float* r = s_rw + qq * SBD_RROW;
    o.m3 = m3; o.m4 = m4;
    // unconditional: a load under a mask bit is followed by its own lgkmcnt(0) (45 waits per step in the first build)
#pragma unroll
    for (int b = 0; b < SBD_NB3; ++b) {
        o.wx3[b] = r[4 * b + (m & 3)];
        o.wy3[b] = r[12 + 4 * b + (m >> 2)];
    }
#pragma unroll
    for (int b = 0; b < SBD_NB4; ++b) {
        o.wx4[b] = r[24 + 4 * b + (m & 3)];
        o.wy4[b] = r[44 + 4 * b + (m >> 2)];
    }
}
// one k-step of a raw level: the blocks in (rows of the mask) x (columns of the mask) take the four queries' weights
template <int NB, int BASE>
__device__ __forceinline__ void sbd_raw_step(f32x4 (&acc)[SBD_NACC], float xa, int mask, const float (&wx)[NB], const float (&wy)[NB]) {
    if (mask == 0) return;
#pragma unroll
    for (int by = 0; by < NB; ++by) {
        if (!(mask & (256 << by))) continue;
#pragma unroll
        for (int bx = 0; bx < NB; ++bx) {
            if (!(mask & (1 << bx))) continue;
            acc[BASE + by * NB + bx] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, wx[bx] * wy[by], acc[BASE + by * NB + bx], 0, 0, 0);
        }
    }
}

// -DSBD_STAMPS (tools/r06_sbd_stamps.sh): wave 0 of every workgroup adds the cycles of its phases to eight device counters
#ifdef SBD_STAMPS
__device__ unsigned long long sbd_stamps[8];
#define SBD_STAMP(i)                                                              \
    do {                                                                          \
        const long long t__ = wall_clock64();                                    \
        if (threadIdx.x == 0) atomicAdd(&sbd_stamps[i], (unsigned long long)(t__ - t_last)); \
        t_last = t__;                                                             \
    } while (0)
extern "C" void s3d_debug_sbd_stamps(unsigned long long* out) {
    unsigned long long z[8] = {0};
    hipMemcpyFromSymbol(out, HIP_SYMBOL(sbd_stamps), sizeof(z));
    hipMemcpyToSymbol(HIP_SYMBOL(sbd_stamps), z, sizeof(z));
}
#else
#define SBD_STAMP(i)
#endif

template <bool F16>
__global__ __launch_bounds__(SBD_THREADS) void sample_bwd_dense_kernel(const SampleBwdArgs a, const SbdGeom G) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_wt = smem;                                          // W_raw^T fragment image [6][8] (fp32) or [6][4] hi|lo pairs
    float* s_x = s_wt + 6 * 8 * 256;                             // the chunk's dX rows; the epilogue's accumulator rows
    float* s_fw = s_x + SBD_XFLOATS;                             // [SBD_CHUNK][SBD_FROW] folded levels: a query's weight on every footprint pixel
    float* s_rw = s_fw + SBD_CHUNK * SBD_FROW;                   // [SBD_CHUNK][SBD_RROW] raw levels: its weight on every footprint column | row
    int* s_mask = reinterpret_cast<int*>(s_rw + SBD_CHUNK * SBD_RROW);    // [2][16] block columns | rows << 8 of every k-step
    const int tile = blockIdx.x & 255;
    const int ts = (blockIdx.x >> 8) % a.n_slices;
    const int b = (blockIdx.x >> 8) / a.n_slices;
    const int* ends = a.bin_ends + (long)b * 65536;
    const long qs_lo = tile ? ends[256 * tile - 1] : 0, qs_hi = ends[256 * tile + 255];
    if (qs_lo >= qs_hi) return;   // sbd_reduce_kernel skips the slot of an empty tile
    {
        const float* wsrc = F16 ? a.ws34_t16 : a.ws34_t;
        for (int i = threadIdx.x; i < 6 * 8 * 64; i += SBD_THREADS) st4(s_wt + 4 * i, ld4(wsrc + 4 * i));
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int T = a.n_slices + 1, t = ts + 1;
    const int tx = (int)sbd_compact8((unsigned)tile), ty = (int)sbd_compact8((unsigned)tile >> 1);
    const long img = (long)b * a.n_slices + ts;
    const float* Tm = a.trans + b * 12;
    auto project_slot = [&](long qs, float& gx, float& gy) {
        if (a.gxy) {
            const s3d_float2 g2 = *reinterpret_cast<const s3d_float2*>(a.gxy + 2 * ((long)b * a.n_qry + qs));
            gx = g2[0]; gy = g2[1];
            return;
        }
        const long q = a.perm[(long)b * a.n_qry + qs];
        const float* p = a.qry + ((long)b * a.n_qry + q) * 3;
        float x = p[0], y = p[1], z = p[2];
        if (a.flip_yz) {
            y = -y; z = -z;
        } else if (a.rot) {
            const float* R = a.rot + b * 9;
            const float rx = x * R[0] + y * R[3] + z * R[6];
            const float ry = x * R[1] + y * R[4] + z * R[7];
            const float rz = x * R[2] + y * R[5] + z * R[8];
            x = rx; y = ry; z = rz;
        }
        const float X = x * Tm[0] + y * Tm[3] + z * Tm[6] + Tm[9];
        const float Y = x * Tm[1] + y * Tm[4] + z * Tm[7] + Tm[10];
        const float Z = x * Tm[2] + y * Tm[5] + z * Tm[8] + Tm[11];
        gx = fminf(fmaxf(2.f * (X / Z - 0.5f), -1.f), 1.f);
        gy = fminf(fmaxf(2.f * (Y / Z - 0.5f), -1.f), 1.f);
    };
    auto row_of = [&](long qs) { return ((((long)b * a.groups_per_batch + (qs >> 4)) * T + t) * S3D_GROUP + (qs & 15)) * 128; };
    // the chunk's 64 rows x 128 floats go straight to LDS (global_load_lds, 16 bytes per lane): 32 pieces of two rows, four per wave;
    // slots past the tile's end repeat its last row (their weights are zero, their values must be finite)
    auto dma_chunk = [&](long c0, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = 4 * wave + i;
            long qs = c0 + 2 * piece + (lane >> 5);
            qs = qs < qs_hi ? qs : qs_hi - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.dX + row_of(qs) + 4 * (lane & 31)),
                                             (__attribute__((address_space(3))) void*)(s_x + buf * (SBD_CHUNK * 128) + piece * 256), 16, 0, 0);
        }
    };

    f32x4 acc[SBD_NACC];
#pragma unroll
    for (int i = 0; i < SBD_NACC; ++i) acc[i] = zero4();
#ifdef SBD_STAMPS
    long long t_last = wall_clock64();
#endif
    dma_chunk(qs_lo, 0);
    int buf = 0;
    // a staging thread's query coordinates are fetched one chunk ahead (their latency sat in front of every chunk's staging)
    float gx_n = 0.f, gy_n = 0.f;
    if (threadIdx.x < 5 * SBD_CHUNK && qs_lo + (threadIdx.x % SBD_CHUNK) < qs_hi) project_slot(qs_lo + (threadIdx.x % SBD_CHUNK), gx_n, gy_n);
    for (long c0 = qs_lo; c0 < qs_hi; c0 += SBD_CHUNK, buf ^= 1) {
        SBD_STAMP(0);      // prologue / k-steps
        __syncthreads();   // the previous chunk's tables have been consumed (first pass: s_wt is complete)
        SBD_STAMP(1);      // wait at the first barrier
#ifdef SBD_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SBD_STAMP(5);      // outstanding loads (this chunk's rows, the staging threads' coordinates)
#endif
        if (threadIdx.x < 5 * SBD_CHUNK) {   // thread = (level, query of the chunk): wave l stages level l
            const int l = threadIdx.x / SBD_CHUNK, qq = threadIdx.x % SBD_CHUNK;
            const long qs = c0 + qq;
            const int W = G.W[l], C = l < 3 ? 128 : (l == 3 ? 64 : 32);
            const int ox = 16 * tx * (W - 1) / 255, oy = 16 * ty * (W - 1) / 255;
            int mask = 0;
            // the query's table row, zeroed; its taps are written over it below
            float* tab = l < 3 ? s_fw + qq * SBD_FROW + (l == 0 ? 0 : l == 1 ? 16 : 32) : s_rw + qq * SBD_RROW + (l == 3 ? 0 : 24);
            {
                const int n4 = l < 2 ? 4 : (l == 2 ? 12 : (l == 3 ? 6 : 10));
                for (int i = 0; i < n4; ++i) st4(tab + 4 * i, zero4());
            }
            if (qs < qs_hi) {
                const float gx = gx_n, gy = gy_n;
                const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
                const float iy = ((gy + 1.f) / 2.f) * (float)(W - 1);
                const float x0f = floorf(ix), y0f = floorf(iy);
                const int x0 = (int)x0f, y0 = (int)y0f;
                const float xe = x0f + 1.f, ye = y0f + 1.f;
                float wx[2] = {xe - ix, ix - x0f}, wy[2] = {ye - iy, iy - y0f};
                if (x0 < 0 || x0 >= W) wx[0] = 0.f;
                if (x0 + 1 < 0 || x0 + 1 >= W) wx[1] = 0.f;
                if (y0 < 0 || y0 >= W) wy[0] = 0.f;
                if (y0 + 1 < 0 || y0 + 1 >= W) wy[1] = 0.f;
                const int cov = G.cov[l], fwl = G.fw[l];   // row width of the tables / slot image; the tile's footprint (fwl <= cov)
                const int rx = x0 - ox, ry = y0 - oy;
                // The sort's bin (floor((gx + 1) * 127.5), possibly one fma) and this level's pixel index (two roundings) are the same
                // number up to an ulp: a query within 1e-5 pixels of its tile's border can land one pixel outside the footprint with
                // the weight ~1e-5 on that side.  It is taken AT the border (the side's weight dropped): the generic path below
                // would cost the workgroup ~1 ms for it.
                if (rx == -1 && wx[0] < 1e-3f) wx[0] = 0.f;
                if (rx + 1 == fwl && wx[1] < 1e-3f) wx[1] = 0.f;
                if (ry == -1 && wy[0] < 1e-3f) wy[0] = 0.f;
                if (ry + 1 == fwl && wy[1] < 1e-3f) wy[1] = 0.f;
                bool inside = true;   // every tap that carries weight lies in the tile's footprint
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int lx = rx + (k & 1), ly = ry + (k >> 1);
                    if (wx[k & 1] * wy[k >> 1] != 0.f && !(lx >= 0 && lx < fwl && ly >= 0 && ly < fwl)) inside = false;
                }
                if (G.slow_mod > 0 && qs % G.slow_mod == 0) inside = false;
                if (inside) {
                    if (l < 3) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float w = wx[k & 1] * wy[k >> 1];
                            if (w != 0.f) tab[(ry + (k >> 1)) * cov + rx + (k & 1)] = w;
                        }
                    } else {
                        // (a weightless tap may sit one pixel outside the footprint: it has no table entry)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            if (rx + k >= 0 && rx + k < fwl) {
                                tab[rx + k] = wx[k];
                                mask |= 1 << ((rx + k) >> 2);
                            }
                            if (ry + k >= 0 && ry + k < fwl) {
                                tab[cov + ry + k] = wy[k];
                                mask |= 256 << ((ry + k) >> 2);
                            }
                        }
                        if (G.slow_mod < 0) mask = 0x1F1F;
                    }
                } else {   // never for the accepted sizes (sbd_fits): this thread adds the query's row to the maps itself
                    const float* xrow = a.dX + row_of(qs);
                    float* map = (l < 3 ? a.dproj[l] : a.dfine[l - 3]) + img * (long)W * W * C;
                    for (int c = 0; c < C; ++c) {
                        float v;
                        if (l < 3) {
                            v = xrow[c];
                        } else {
                            v = 0.f;
                            for (int k = 0; k < 128; ++k) v = fmaf(xrow[k], sbd_wraw(a.ws34_t, l == 3 ? 0 : 4, k, c), v);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float w = wx[k & 1] * wy[k >> 1];
                            if (w != 0.f) unsafeAtomicAdd(map + ((long)(y0 + (k >> 1)) * W + x0 + (k & 1)) * C + c, v * w);
                        }
                    }
                }
            }
            if (l >= 3) {
                mask |= __shfl_xor(mask, 1, 64);
                mask |= __shfl_xor(mask, 2, 64);
                if ((qq & 3) == 0) s_mask[(l - 3) * 16 + (qq >> 2)] = mask;
            }
        }
        SBD_STAMP(2);      // staging
        dma_publish_barrier();   // the tables and this chunk's rows (every wave's own DMA has landed)
        SBD_STAMP(3);      // wait at the second barrier
        if (c0 + SBD_CHUNK < qs_hi) dma_chunk(c0 + SBD_CHUNK, buf ^ 1);   // the other buffer was last read before this chunk's first barrier
        const float* s_xb = s_x + buf * (SBD_CHUNK * 128);
        if (threadIdx.x < 5 * SBD_CHUNK && c0 + SBD_CHUNK + (threadIdx.x % SBD_CHUNK) < qs_hi)
            project_slot(c0 + SBD_CHUNK + (threadIdx.x % SBD_CHUNK), gx_n, gy_n);
        const int nst = (int)((qs_hi - c0 + 3) / 4 < 16 ? (qs_hi - c0 + 3) / 4 : 16);
        const int vm3 = s_mask[lane & 15], vm4 = s_mask[16 + (lane & 15)];   // lane st holds k-step st's masks: no LDS wait per step
        auto step = [&](const SbdOps& o) {
#pragma unroll
            for (int t = 0; t < 5; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.xa, o.wb[t], acc[t], 0, 0, 0);
            sbd_raw_step<SBD_NB3, 5>(acc, o.xa, o.m3, o.wx3, o.wy3);
            sbd_raw_step<SBD_NB4, 5 + SBD_NB3 * SBD_NB3>(acc, o.xa, o.m4, o.wx4, o.wy4);
        };
        auto load = [&](SbdOps& o, int st) {
            st = st < nst ? st : (nst ? nst - 1 : 0);     // (past the end: a valid step again, never computed)
            sbd_load_ops(o, s_xb, s_fw, s_rw, st, g, m, wave, __builtin_amdgcn_readlane(vm3, st), __builtin_amdgcn_readlane(vm4, st));
        };
        SbdOps oa, ob;   // the next step's operands are read while this step's MFMAs run
        load(oa, 0);
#pragma unroll 1
        for (int st = 0; st < nst; st += 2) {
            load(ob, st + 1);
            step(oa);
            if (st + 1 < nst) {
                load(oa, st + 2);
                step(ob);
            }
        }
    }

    SBD_STAMP(0);
    // ---- the slot: [level][footprint pixel][C_l] ----
    float* slot = a.partial + ((img * 256 + tile) * (long)G.ptotal);
    // folded levels straight from the registers: D[row = channel 16 wave + 4g + i][col = pixel 16k + m]
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const int npx = G.fw[l] * G.fw[l];
#pragma unroll
        for (int k = 0; k < sbd_nmt(l); ++k) {
            const int pix = 16 * k + m;
            if (pix < npx) st4(slot + G.poff[l] + (long)pix * 128 + 16 * wave + 4 * g, acc[sbd_mt0(l) + k]);
        }
    }
    // raw levels: eight blocks per round through LDS (block = 16 pixel rows of 128 channels), one block per wave back out
#pragma unroll
    for (int r = 0; r < (SBD_NBLK + 7) / 8; ++r) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (8 * r + k < SBD_NBLK) st4(s_x + (k * 16 + m) * SBD_PROW + 16 * wave + 4 * g, acc[5 + 8 * r + k]);
        __syncthreads();
        const int blk = 8 * r + wave;
        if (blk < SBD_NB3 * SBD_NB3) {
            sbd_project_block<F16, 3>(s_x + wave * 16 * SBD_PROW, s_wt, slot + G.poff[3], G.cov[3], G.fw[3], blk % SBD_NB3, blk / SBD_NB3, lane);
        } else if (blk < SBD_NBLK) {
            const int bi = blk - SBD_NB3 * SBD_NB3;
            sbd_project_block<F16, 4>(s_x + wave * 16 * SBD_PROW, s_wt, slot + G.poff[4], G.cov[4], G.fw[4], bi % SBD_NB4, bi / SBD_NB4, lane);
        }
    }
    SBD_STAMP(4);          // epilogue
}

// map[l][img][y][x][c] += sum over the tiles whose footprint holds (x, y), ascending (ty, tx), of their slot's value
__global__ __launch_bounds__(256) void sbd_reduce_kernel(const SampleBwdArgs a, const SbdGeom G, int batch) {
    const long n_img = (long)batch * a.n_slices;
    long lvl_end[5];
    {
        long s = 0;
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            s += n_img * G.W[l] * G.W[l] * (sbd_C(l) / 4);
            lvl_end[l] = s;
        }
    }
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < lvl_end[4]; idx += (long)gridDim.x * blockDim.x) {
        int l = 0;
        while (idx >= lvl_end[l]) ++l;
        // 32-bit index arithmetic from here (launch_sample_bwd_dense checks the level sizes): the 64-bit divisions of the first build
        // were most of this kernel's time
        const unsigned r = (unsigned)(idx - (l ? lvl_end[l - 1] : 0));
        const int W = G.W[l], C = sbd_C(l), c4s = l < 3 ? 5 : (l == 3 ? 4 : 3);   // C / 4 = 32, 16, 8 quads per pixel
        const int c4 = (int)(r & ((1u << c4s) - 1));
        const unsigned p = r >> c4s;
        const unsigned row = p / (unsigned)W;
        const int x = (int)(p - row * (unsigned)W);
        const unsigned img = row / (unsigned)W;
        const int y = (int)(row - img * (unsigned)W);
        const int b = (int)(img / a.n_slices);
        const int* ends = a.bin_ends + (long)b * 65536;
        const int cov = G.cov[l], fwl = G.fw[l], den = 16 * (W - 1);
        // ox(t) = 16 t (W-1) / 255 is non-decreasing in t: candidates are the t with ox(t) in (x - fw, x]
        const int tx_lo = x - fwl >= 0 ? (x - fwl) * 255 / den : 0, tx_hi = min(15, (x + 1) * 255 / den);
        const int ty_lo = y - fwl >= 0 ? (y - fwl) * 255 / den : 0, ty_hi = min(15, (y + 1) * 255 / den);
        float* o = (l < 3 ? a.dproj[l] : a.dfine[l - 3]) + (long)p * C + 4 * c4;
        const f32x4 old = ld4(o);   // requested first: the candidate walk below is two dependent loads deep
        f32x4 s = zero4();
        bool any = false;
        for (int ty = ty_lo; ty <= ty_hi; ++ty) {
            const int ly = y - 16 * ty * (W - 1) / 255;
            if (ly < 0 || ly >= fwl) continue;
            for (int tx = tx_lo; tx <= tx_hi; ++tx) {
                const int lx = x - 16 * tx * (W - 1) / 255;
                if (lx < 0 || lx >= fwl) continue;
                const int tile = (int)(sbd_spread8((unsigned)tx) | (sbd_spread8((unsigned)ty) << 1));
                const int lo = tile ? ends[256 * tile - 1] : 0, hi = ends[256 * tile + 255];
                if (lo >= hi) continue;
                s += ld4(a.partial + ((long)img * 256 + tile) * (long)G.ptotal + G.poff[l] + (ly * cov + lx) * C + 4 * c4);
                any = true;
            }
        }
        if (any) st4(o, old + s);
    }
}

static bool sbd_geom(int S, SbdGeom& G) {
    int off = 0;
    bool fits = true;
    for (int l = 0; l < 5; ++l) {
        const int W = S >> (4 - l);
        G.W[l] = W;
        G.fw[l] = (16 * (W - 1) + 254) / 255 + 2;
        if (l < 3) {
            G.cov[l] = G.fw[l];
            fits = fits && G.fw[l] * G.fw[l] <= 16 * sbd_nmt(l);
        } else {
            const int nb = (G.fw[l] + 3) / 4;
            G.cov[l] = 4 * (l == 3 ? SBD_NB3 : SBD_NB4);
            fits = fits && nb <= (l == 3 ? SBD_NB3 : SBD_NB4);
        }
        // the tile's bins are level-4 pixels of a 256-wide image ((gx + 1) * 127.5, launch_query_sort); a coarser level's footprint
        // starts at 16 tx (W-1) / 255: when that is never an integer for tx = 1..15 — or the level IS the 256-wide one, whose
        // coordinate is the bin's own expression — a rounding error of the coordinate (~1e-5 pixels) cannot move a tap across it
        if (W < 2) fits = false;
        for (int t = 1; t <= 15 && W != 256; ++t) fits = fits && (16 * t * (W - 1)) % 255 != 0;
        G.poff[l] = off;
        off += G.cov[l] * G.cov[l] * sbd_C(l);
    }
    G.ptotal = off;
    G.slow_mod = 0;
    return fits && S <= 256;
}

size_t sample_bwd_partial_floats(int S, long batch, int n_slices) {
    SbdGeom G;
    if (!sbd_geom(S, G)) return 0;
    return (size_t)batch * n_slices * 256 * G.ptotal;
}

bool sample_bwd_dense_covers(const SampleBwdArgs& a) {
    SbdGeom G;
    if (a.gt || !a.perm || !a.bin_ends || !a.partial || !sbd_geom(a.size, G)) return false;
    if (const char* e = getenv("S3D_SBD_OFF")) if (atoi(e)) return false;   // A/B switch of tools/dbg_sbd.py
    const long n_img = a.groups / a.groups_per_batch * a.n_slices;
    if (n_img * a.size * a.size * 8 >= (1L << 32)) return false;   // sbd_reduce_kernel's 32-bit pixel index (level 4: 8 quads per pixel)
    return true;
}

// 1 = launched, 0 = this size is not covered (the caller falls back to the atomic kernels), < 0: error.  a.gxy, when given, has
// been filled by the caller (launch_sample_bwd's pre-pass).
int launch_sample_bwd_dense(const SampleBwdArgs& a, hipStream_t stream) {
    SbdGeom G;
    if (!sample_bwd_dense_covers(a) || !sbd_geom(a.size, G)) return 0;
    if (const char* e = getenv("S3D_SBD_SLOW_MOD")) G.slow_mod = atoi(e);
    const size_t lds = (size_t)SBD_LDS_FLOATS * sizeof(float);
    static std::atomic<unsigned long long> attr_done{0};
    if (s3d_set_max_lds(attr_done, {(const void*)sample_bwd_dense_kernel<false>, (const void*)sample_bwd_dense_kernel<true>}, 160 * 1024))
        return -1;
    const long batch = a.groups / a.groups_per_batch;
    if (a.ws34_t16)
        hipLaunchKernelGGL((sample_bwd_dense_kernel<true>), dim3((unsigned)(batch * a.n_slices * 256)), dim3(SBD_THREADS), lds, stream, a, G);
    else
        hipLaunchKernelGGL((sample_bwd_dense_kernel<false>), dim3((unsigned)(batch * a.n_slices * 256)), dim3(SBD_THREADS), lds, stream, a, G);
    if (hipGetLastError() != hipSuccess) return -1;
    long quads = 0;
    for (int l = 0; l < 5; ++l) quads += batch * a.n_slices * G.W[l] * G.W[l] * (sbd_C(l) / 4);   // (each level's count < 2^32: checked in sample_bwd_dense_covers)
    const long blocks = (quads + 255) / 256 < 16384 ? (quads + 255) / 256 : 16384;
    hipLaunchKernelGGL(sbd_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, G, (int)batch);
    if (hipGetLastError() != hipSuccess) return -1;
    return 1;
}
