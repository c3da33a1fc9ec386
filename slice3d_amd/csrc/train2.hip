// train2.hip — decoder-specific backward kernels: pyramid-sampling backward (scatter-add), token-0
// gather/scatter, fc_out, metrics, VGG input-normalisation backward.
#include "train.h"

// ---- copies between the token tensor [groups][T][16][128] and its compact token-0 rows [groups*16][128]
__global__ void tok0_copy_kernel(float* __restrict__ full, float* __restrict__ compact, long groups, int T,
                                 int dir, int width) {
    const int w4 = width >> 2;
    const long total = groups * 16 * w4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % w4) * 4;
        const long row = idx / w4;
        const long grp = row >> 4, ql = row & 15;
        float* f = full + ((grp * T) * 16 + ql) * width + c;
        float* k = compact + row * width + c;
        if (dir == 0) st4(k, ld4(f)); else st4(f, ld4(k));
    }
}
int launch_tok0_copy(float* full, float* compact, long groups, int T, int dir, int width, hipStream_t stream) {
    const long total = groups * 16 * (width / 4);
    if (total <= 0) return 0;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(tok0_copy_kernel, dim3(blocks), dim3(256), 0, stream, full, compact, groups, T, dir, width);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---- fc_out on compact token-0 rows: sdf[b*Q+q] = x[row].w + b ; backward d x = dsdf*w, t = dsdf*x
__global__ __launch_bounds__(256) void fc_out_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ sdf,
                                                         long rows, long gpb, long n_qry,
                                                         const int* __restrict__ perm) {
    const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;
    const f32x4 wv = ld4(w + 4 * l);
    for (long row = (long)blockIdx.x * 8 + sub; row < rows; row += (long)gridDim.x * 8) {
        const f32x4 v = ld4(x + row * 128 + 4 * l);
        float s = v[0] * wv[0] + v[1] * wv[1] + v[2] * wv[2] + v[3] * wv[3];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const long grp = row >> 4, bb = grp / gpb, q = (grp % gpb) * 16 + (row & 15);
        if (l == 0 && q < n_qry) sdf[bb * n_qry + (perm ? perm[bb * n_qry + q] : q)] = s + b[0];
    }
}
__global__ __launch_bounds__(256) void fc_out_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ dsdf, float* __restrict__ dx,
                                                         float* __restrict__ t, long rows, long gpb, long n_qry,
                                                         const int* __restrict__ perm) {
    const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;
    const f32x4 wv = ld4(w + 4 * l);
    for (long row = (long)blockIdx.x * 8 + sub; row < rows; row += (long)gridDim.x * 8) {
        const long grp = row >> 4, bb = grp / gpb, q = (grp % gpb) * 16 + (row & 15);
        const float d = q < n_qry ? dsdf[bb * n_qry + (perm ? perm[bb * n_qry + q] : q)] : 0.f;
        st4(dx + row * 128 + 4 * l, wv * d);
        st4(t + row * 128 + 4 * l, ld4(x + row * 128 + 4 * l) * d);
    }
}
int launch_fc_out_fwd(const float* x, const float* w, const float* b, float* sdf, long rows, long gpb, long n_qry,
                      const int* perm, hipStream_t stream) {
    const long nb = (rows + 7) / 8 < 4096 ? (rows + 7) / 8 : 4096;
    hipLaunchKernelGGL(fc_out_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, w, b, sdf, rows, gpb, n_qry, perm);
    S3D_LAUNCH_CHECK();
    return 0;
}
int launch_fc_out_bwd(const float* x, const float* w, const float* dsdf, float* dx, float* t, long rows, long gpb,
                      long n_qry, const int* perm, hipStream_t stream) {
    const long nb = (rows + 7) / 8 < 4096 ? (rows + 7) / 8 : 4096;
    hipLaunchKernelGGL(fc_out_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, w, dsdf, dx, t, rows, gpb,
                       n_qry, perm);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---- scalar reductions: mode 0: sum x ; mode 1: count[(a>=0)==(b>=0)]
__global__ __launch_bounds__(256) void scalar_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             long n, int mode, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        s += mode == 0 ? a[i] : (((a[i] >= 0.f) == (b[i] >= 0.f)) ? 1.f : 0.f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void scalar_final2_kernel(const float* __restrict__ partial, int n, float scale,
                                                            float* __restrict__ out, int accumulate) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = scale * ((red[0] + red[1]) + (red[2] + red[3]));
        out[0] = accumulate ? out[0] + v : v;
    }
}
int launch_scalar_reduce(const float* a, const float* b, long n, int mode, float scale, float* out, int accumulate,
                         float* partial, hipStream_t stream) {
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(scalar_partial_kernel, dim3(blocks), dim3(256), 0, stream, a, b, n, mode, partial);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(scalar_final2_kernel, dim3(1), dim3(256), 0, stream, partial, blocks, scale, out, accumulate);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---- VGG input normalisation backward: d rec[n][c][y][x] += d in16[n][y][x][c] * 0.5 / std[c]
__global__ void vgg_prep_bwd_kernel(const float* __restrict__ din16, const float* __restrict__ stdv,
                                    float* __restrict__ drec, int n_img, int size) {
    const long hw = (long)size * size;
    const long total = (long)n_img * 3 * hw;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const long r = idx % hw;
        const int c = (int)((idx / hw) % 3);
        const long ni = idx / (3 * hw);
        drec[idx] += din16[(ni * hw + r) * 16 + c] * (0.5f / stdv[c]);
    }
}
int launch_vgg_prep_bwd(const float* din16, const float* stdv, float* drec, int n_img, int size,
                        hipStream_t stream) {
    const long total = (long)n_img * 3 * size * size;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(vgg_prep_bwd_kernel, dim3(blocks), dim3(256), 0, stream, din16, stdv, drec, n_img, size);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---- data gradient of VGG conv1_1 (64 -> 3, 3x3) + the input normalisation, in two steps: Z[p][ci*9 + tap] = sum_co dY[p][co]
//      w[co][ci][tap] is a 1 x 1 convolution 64 -> 27 (the row-linear kernel: the generic 3x3 tile pads the 3 outputs to 16 and
//      runs K = 576 for them, 0.87 ms), and d rec[n][ci][y][x] += 0.5 / std[ci] * sum_tap Z[(y, x) - d(tap)][ci*9 + tap] gathers the
//      nine neighbours' entries from an 18 x 18-pixel halo of Z staged in LDS (zero outside the image).
__global__ __launch_bounds__(256) void vgg_first_bwd_kernel(const float* __restrict__ Z, const float* __restrict__ stdv,
                                                            float* __restrict__ drec, int n_img, int size) {
    __shared__ float s_z[18 * 18][33];   // (padded: the 27 entries of a pixel are read at a 33-float stride)
    const int tiles = size >> 4;
    const long hw = (long)size * size;
    for (long t = blockIdx.x; t < (long)n_img * tiles * tiles; t += gridDim.x) {
        const int tx = (int)(t % tiles), ty = (int)((t / tiles) % tiles);
        const long ni = t / ((long)tiles * tiles);
        __syncthreads();
        for (int i = threadIdx.x; i < 18 * 18 * 8; i += 256) {   // float4 slots: pixel * 8 + quad
            const int pix = i >> 3, q = i & 7, hy = pix / 18, hx = pix - hy * 18;
            const int y = 16 * ty + hy - 1, x = 16 * tx + hx - 1;
            f32x4 v = zero4();
            if ((unsigned)y < (unsigned)size && (unsigned)x < (unsigned)size) v = ld4(Z + ((ni * hw + (long)y * size + x) * 32) + 4 * q);
            s_z[pix][4 * q] = v[0]; s_z[pix][4 * q + 1] = v[1]; s_z[pix][4 * q + 2] = v[2]; s_z[pix][4 * q + 3] = v[3];
        }
        __syncthreads();
        const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {   // Y[q] read X[q + (dy-1, dx-1)] with tap (dy, dx): X[p] collects from q = p - (dy-1, dx-1)
                const float* z = s_z[(ly + 1 - (dy - 1)) * 18 + (lx + 1 - (dx - 1))];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] += z[c * 9 + dy * 3 + dx];
            }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            drec[(ni * 3 + c) * hw + (long)(16 * ty + ly) * size + 16 * tx + lx] += acc[c] * (0.5f / stdv[c]);
    }
}
int launch_vgg_first_bwd(const float* Z, const float* stdv, float* drec, int n_img, int size, hipStream_t stream) {
    S3D_CHECK_ARG(size % 16 == 0, "vgg_first_bwd: size %d", size);
    const long tiles = (long)n_img * (size / 16) * (size / 16);
    hipLaunchKernelGGL(vgg_first_bwd_kernel, dim3((unsigned)(tiles < 16384 ? tiles : 16384)), dim3(256), 0, stream, Z, stdv, drec,
                       n_img, size);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---- rotated query coordinates of the token-0 rows, padded to 4 columns: [groups*16][4] = (x,y,z,0)
__global__ void qry_rot_rows_kernel(const float* __restrict__ qry, const float* __restrict__ rot, int flip_yz,
                                    long n_qry, long gpb, long groups, const int* __restrict__ perm,
                                    float* __restrict__ out) {
    const long total = groups * 16;
    for (long row = (long)blockIdx.x * blockDim.x + threadIdx.x; row < total; row += (long)gridDim.x * blockDim.x) {
        const long grp = row >> 4, b = grp / gpb, q = (grp % gpb) * 16 + (row & 15);
        f32x4 v = zero4();
        if (q < n_qry) {
            const float* p = qry + (b * n_qry + (perm ? perm[b * n_qry + q] : q)) * 3;
            float x = p[0], y = p[1], z = p[2];
            if (flip_yz) {
                y = -y; z = -z;
            } else if (rot) {
                const float* R = rot + b * 9;
                const float rx = x * R[0] + y * R[3] + z * R[6];
                const float ry = x * R[1] + y * R[4] + z * R[7];
                const float rz = x * R[2] + y * R[5] + z * R[8];
                x = rx; y = ry; z = rz;
            }
            v[0] = x; v[1] = y; v[2] = z;
        }
        st4(out + row * 4, v);
    }
}
int launch_qry_rot_rows(const float* qry, const float* rot, int flip_yz, long n_qry, long gpb, long groups,
                        const int* perm, float* out, hipStream_t stream) {
    const long total = groups * 16;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(qry_rot_rows_kernel, dim3(blocks), dim3(256), 0, stream, qry, rot, flip_yz, n_qry, gpb, groups,
                       perm, out);
    S3D_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// sampling backward: dX0 [groups][T][16][128] -> scatter-add into d(latent maps)
//   levels 0-2 (fc_s folded, 128 ch): dG_l[tap pixel][c] += w_tap * dtok[c]
//   levels 3-4: d raw34 = Ws34^T dtok (MFMA, Ws34^T fragments in LDS) then scatter with the tap weights
// fp32 hardware atomics (global_atomic_add_f32): summation order is not deterministic.
// =============================================================================================
struct Tap4b {
    int off[4];
    float w[4];
};
__device__ __forceinline__ Tap4b make_taps_b(float gx, float gy, int W, int H) {
    const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float xe = x0f + 1.f, ye = y0f + 1.f;
    Tap4b t;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
    const bool vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    t.off[0] = cy0 * W + cx0; t.w[0] = (vx0 && vy0) ? (xe - ix) * (ye - iy) : 0.f;
    t.off[1] = cy0 * W + cx1; t.w[1] = (vx1 && vy0) ? (ix - x0f) * (ye - iy) : 0.f;
    t.off[2] = cy1 * W + cx0; t.w[2] = (vx0 && vy1) ? (xe - ix) * (iy - y0f) : 0.f;
    t.off[3] = cy1 * W + cx1; t.w[3] = (vx1 && vy1) ? (ix - x0f) * (iy - y0f) : 0.f;
    return t;
}

__device__ __forceinline__ void atomic_add4(float* p, const f32x4 v) {
    unsafeAtomicAdd(p + 0, v[0]);
    unsafeAtomicAdd(p + 1, v[1]);
    unsafeAtomicAdd(p + 2, v[2]);
    unsafeAtomicAdd(p + 3, v[3]);
}

// GT = 0: Slices3DRegModel levels (3 folded 128-ch maps, raw 64-ch and 32-ch maps through Ws34^T, NU = 6 tiles)
// GT = 1: Slices3DGTModel levels (4 folded 128-ch maps, the raw 64-ch conv1_2 map through Wraw^T, NU = 4 tiles);
//         the five gradient maps are dproj[0..2], dfine[0] (128 ch), dfine[1] (64 ch)
template <int GT>
struct SbLevels {
    static constexpr int NU = GT ? 4 : 6;
    __host__ __device__ static constexpr int C(int l) { return GT ? (l < 4 ? 128 : 64) : (l < 3 ? 128 : (l == 3 ? 64 : 32)); }
    __host__ __device__ static constexpr bool folded(int l) { return GT ? l < 4 : l < 3; }
    __host__ __device__ static constexpr int draw0(int l) { return GT ? 0 : (l == 3 ? 0 : 4); }   // first draw tile of a raw level
};

template <int GT>
__global__ __launch_bounds__(256) void sample_bwd_kernel(const SampleBwdArgs a) {
    using LV = SbLevels<GT>;
    constexpr int NU = LV::NU;
    __shared__ __attribute__((aligned(16))) float s_wt[NU * 8 * 256];  // W_raw^T fragment image [NU][8]
    for (int i = threadIdx.x; i < NU * 8 * 64; i += 256) st4(s_wt + 4 * i, ld4(a.ws34_t + 4 * i));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int T = a.n_slices + 1, S = a.size;
    for (long gi = blockIdx.x; gi < a.groups; gi += gridDim.x) {
        const int b = (int)(gi / a.groups_per_batch);
        const long q = (gi % a.groups_per_batch) * S3D_GROUP + m;
        const bool qv = q < a.n_qry;
        long qc = qv ? q : a.n_qry - 1;
        if (a.perm) qc = a.perm[(long)b * a.n_qry + qc];
        const float* p = a.qry + ((long)b * a.n_qry + qc) * 3;
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
        const float* Tm = a.trans + b * 12;
        const float X = x * Tm[0] + y * Tm[3] + z * Tm[6] + Tm[9];
        const float Y = x * Tm[1] + y * Tm[4] + z * Tm[7] + Tm[10];
        const float Z = x * Tm[2] + y * Tm[5] + z * Tm[8] + Tm[11];
        const float gx = fminf(fmaxf(2.f * (X / Z - 0.5f), -1.f), 1.f);
        const float gy = fminf(fmaxf(2.f * (Y / Z - 0.5f), -1.f), 1.f);
        for (int t = 1 + wave; t < T; t += 4) {
            const long img = (long)b * a.n_slices + (t - 1);
            const float* dp = a.dX + ((gi * T + t) * S3D_GROUP + m) * 128 + 4 * g;
            f32x4 dt[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) dt[j] = qv ? ld4(dp + 16 * j) : zero4();
            // d raw = W_raw^T dtok  (NU output tiles of 16 channels, K = 128)
            f32x4 draw[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                f32x4 c = zero4();
#pragma unroll
                for (int j = 0; j < 8; ++j) c = mfma4(ld4(s_wt + ((u * 8 + j) * 64 + lane) * 4), dt[j], c);
                draw[u] = c;
            }
            if (!qv) continue;
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                const int C = LV::C(l), nv = C / 16;
                const int W = S >> (4 - l);
                const Tap4b tp = make_taps_b(gx, gy, W, W);
                float* base = (l < 3 ? a.dproj[l] : a.dfine[l - 3]) + img * (long)W * W * C + 4 * g;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (tp.w[k] == 0.f) continue;
                    float* o = base + (long)tp.off[k] * C;
#pragma unroll
                    for (int j = 0; j < nv; ++j)
                        atomic_add4(o + 16 * j, (LV::folded(l) ? dt[j] : draw[LV::draw0(l) + j]) * tp.w[k]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Tiled variant for locality-sorted queries (a.perm / a.bin_ends from launch_query_sort).
// One workgroup per (batch item, slice image, 16x16-bin image tile): the tile's queries are one contiguous
// range of the sorted order, their bilinear taps fall into a small per-level pixel footprint
// (ceil(16 (W-1)/255) + 2 squared), which is accumulated with LDS atomics and flushed to the global
// gradient maps once — ~40 global atomics per (query, slice) instead of 1920.
// ---------------------------------------------------------------------------------------------
struct SbtGeom {
    int W[5], C[5], fw[5], off[5];   // level width, channels, footprint width, LDS float offset
    int total;                       // floats of LDS accumulators
};
// LDS accumulators of the raw levels: a footprint pixel's C channels are followed by SBT_PAD unused floats (round 5).  With the
// dense stride (64 / 32 floats = 64 / 32 banks) every pixel's channel c sat in the SAME bank: the ds_add_f32 of a tap — active
// lanes = the tails of the row's runs, i.e. different pixels, same channel — was a 16-way bank conflict, and cycle stamps
// (tools/patches/sample_bwd_stamps.patch) showed the level loop at 31-50 k cycles per 16-query group with the other waves'
// fragment reads queued behind it (72 MFMAs taking 8-25 k cycles).  A stride of C + 16 floats moves consecutive pixels by 16
// banks: four neighbouring pixels x four channel quads per instruction are conflict-free.
#define SBT_PAD 16
__device__ __forceinline__ unsigned compact8(unsigned v) {   // even bits of a 16-bit Morton code
    v &= 0x5555u;
    v = (v | (v >> 1)) & 0x3333u;
    v = (v | (v >> 2)) & 0x0F0Fu;
    v = (v | (v >> 4)) & 0x00FFu;
    return v;
}
struct TapL {
    int lofs[4];   // LDS pixel index inside the footprint, or -1 -> use gofs
    int gofs[4];   // global pixel index
    float w[4];
    bool ok[4];    // tap lies inside the map (an outside tap has weight 0 and is never written)
};
__device__ __forceinline__ TapL make_taps_l(float gx, float gy, int W, int ox, int oy, int fw) {
    const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(W - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float xe = x0f + 1.f, ye = y0f + 1.f;
    const float wx[2] = {xe - ix, ix - x0f}, wy[2] = {ye - iy, iy - y0f};
    TapL t;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + (k & 1), y = y0 + (k >> 1);
        const bool ok = x >= 0 && x < W && y >= 0 && y < W;
        t.ok[k] = ok;
        t.w[k] = ok ? wx[k & 1] * wy[k >> 1] : 0.f;
        t.gofs[k] = ok ? y * W + x : 0;
        const int lx = x - ox, ly = y - oy;
        t.lofs[k] = (ok && lx >= 0 && lx < fw && ly >= 0 && ly < fw) ? ly * fw + lx : -1;
    }
    return t;
}

#define SBT_THREADS 512    // 8 waves per (object, slice, tile); 1024 threads measured the same on the LDS-add form of round 2
#define SBT_CHUNK 64       // queries per MFMA chunk of the folded levels (16 k-steps of 4)
// M tiles (16 footprint pixels each) of the folded levels at S <= 256: footprint widths 3, 4, 6 (, 10) -> 1, 1, 3 (, 7) tiles
template <int GT>
struct SbFold {
    static constexpr int NF = GT ? 4 : 3;                       // folded 128-channel levels: the MFMA path
    static constexpr int NMT = GT ? 12 : 5;                     // accumulator tiles per wave
    __host__ __device__ static constexpr int mt0(int l) { return l == 0 ? 0 : l == 1 ? 1 : l == 2 ? 2 : 5; }
    __host__ __device__ static constexpr int nmt(int l) { return l < 2 ? 1 : l == 2 ? 3 : 7; }
};

// Round 3.  The FOLDED levels (three — GT: four — coarse 128-channel maps) no longer go through LDS adds.  All queries of a
// tile hit the same few pixels there (3 x 3 / 4 x 4 / 6 x 6 at 256^2), so their scatter is a small dense product
//     dMap[pix][c] = sum_q w[q][pix] * dX[q][c]
// and runs on the fp32 MFMA (exact products, fp32 accumulate): wave j owns channels 16j..16j+15 of every folded level, a
// k-step is four queries (A[pix m][query g] = the query's bilinear weight on footprint pixel m — built from the four
// (pixel, weight) taps a cooperative pass parks in LDS per 64-query chunk; B[query g][channel m] straight from dX), the
// accumulators (5 / 12 tiles of 16 pixels) stay in registers for the whole tile and are flushed with one global atomic per
// element.  That form cost 1 536 ds_add_f32 per (query, slice) row before — 3/4 of the kernel's 15.5 ms.  The RAW levels
// (64 / 32 channels at S/2 and S: every query has its own pixels) keep the run-reduced LDS adds below.
// F16 (split-precision training): the product with W_raw^T runs on the f16x3 MFMA (round 4: cycle stamps showed its fp32 form — 192
// v_mfma_f32_16x16x4_f32 in chains of 32 dependent instructions per 16-query task, shared by the two waves of a SIMD — at 7 000 to
// 23 000 of a task's ~50 000 cycles); a lane then reads 8 consecutive channels of each 32-channel block of its dX row.
// pre-pass of the tiled kernel: gxy[b][slot] = projected, clamped image coordinates of the query in sorted slot `slot`
// (models.py:28-36 after the rotation of :58-60) — the same expressions, in the same order, as the tiled kernel used in place
__global__ void sbt_project_kernel(const SampleBwdArgs a, int batch) {
    const long total = (long)batch * a.n_qry;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / a.n_qry);
        const long q = a.perm[i];
        const float* p = a.qry + ((long)b * a.n_qry + q) * 3;
        const float* Tm = a.trans + b * 12;
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
        a.gxy[2 * i] = fminf(fmaxf(2.f * (X / Z - 0.5f), -1.f), 1.f);
        a.gxy[2 * i + 1] = fminf(fmaxf(2.f * (Y / Z - 0.5f), -1.f), 1.f);
    }
}

template <int GT, bool F16>
__global__ __launch_bounds__(SBT_THREADS) void sample_bwd_tiled_kernel(const SampleBwdArgs a, const SbtGeom G) {
    using LV = SbLevels<GT>;
    using FD = SbFold<GT>;
    constexpr int NU = LV::NU, NF = FD::NF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_wt = smem;                  // W_raw^T fragment image [NU][8]
    float* s_acc = smem + NU * 8 * 256;  // footprint accumulators of the RAW levels
    int* s_tidx = reinterpret_cast<int*>(s_acc + G.total);          // [NF][SBT_CHUNK][4] footprint pixel of each tap, or -1
    float* s_tw = reinterpret_cast<float*>(s_tidx + NF * SBT_CHUNK * 4);   // [NF][SBT_CHUNK][4] its weight
    const int tile = blockIdx.x & 255;
    const int ts = (blockIdx.x >> 8) % a.n_slices;
    const int b = (blockIdx.x >> 8) / a.n_slices;
    const int* ends = a.bin_ends + (long)b * 65536;
    const long qs_lo = tile ? ends[256 * tile - 1] : 0, qs_hi = ends[256 * tile + 255];
    if (qs_lo >= qs_hi) return;
    {
        const float* wsrc = F16 ? a.ws34_t16 : a.ws34_t;   // fp32 [NU][8] fragments or f16 hi|lo [NU][4] pairs: the same size
        for (int i = threadIdx.x; i < NU * 8 * 64; i += SBT_THREADS) st4(s_wt + 4 * i, ld4(wsrc + 4 * i));
    }
    for (int i = threadIdx.x; i < G.total / 4; i += SBT_THREADS) st4(s_acc + 4 * i, zero4());
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int T = a.n_slices + 1, t = ts + 1;
    const int tx = (int)compact8((unsigned)tile), ty = (int)compact8((unsigned)tile >> 1);
    int ox[5], oy[5];
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        ox[l] = 16 * tx * (G.W[l] - 1) / 255;
        oy[l] = 16 * ty * (G.W[l] - 1) / 255;
    }
    const long img = (long)b * a.n_slices + ts;
    const float* Tm = a.trans + b * 12;
    // image coordinates of sorted slot qs of this object (clamped into the tile's range by the caller)
    auto project_slot = [&](long qs, float& gx, float& gy) {
        if (a.gxy) {   // pre-pass (sbt_project_kernel): one coalesced 8-byte load
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

    // ================= raw levels: W_raw^T product + run-reduced LDS adds, one 16-query group per wave =================
    for (long gi = qs_lo / 16 + wave; gi * 16 < qs_hi; gi += SBT_THREADS / 64) {
        const long qs = gi * 16 + m;
        const bool qv = qs >= qs_lo && qs < qs_hi;
        // lanes outside the tile's range carry zero gradients; they borrow a valid neighbour's coordinates so
        // that they do not break the row-uniformity test below
        const long qsc = qs < qs_lo ? qs_lo : (qs >= qs_hi ? qs_hi - 1 : qs);
        float gx, gy;
        project_slot(qsc, gx, gy);
        const float* dp = a.dX + ((((long)b * a.groups_per_batch + gi) * T + t) * S3D_GROUP + m) * 128 + (F16 ? 8 : 4) * g;
        f32x4 dt[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dt[j] = qv ? ld4(dp + (F16 ? 32 * (j >> 1) + 4 * (j & 1) : 16 * j)) : zero4();
        f32x4 draw[NU];
        if (F16) {
            const _Float16* sw = reinterpret_cast<const _Float16*>(s_wt);
#pragma unroll
            for (int u = 0; u < NU; ++u) draw[u] = zero4();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {   // k-slot 8g + t of step kk <-> token channel 32 kk + 8g + t
                const float x8[8] = {dt[2 * kk][0], dt[2 * kk][1], dt[2 * kk][2], dt[2 * kk][3],
                                     dt[2 * kk + 1][0], dt[2 * kk + 1][1], dt[2 * kk + 1][2], dt[2 * kk + 1][3]};
                s3d_half8 bh, bl;
                s3d_split8(x8, bh, bl);
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const s3d_half8 fh = *reinterpret_cast<const s3d_half8*>(sw + (u * 4 + kk) * 1024 + lane * 8);
                    const s3d_half8 fl = *reinterpret_cast<const s3d_half8*>(sw + (u * 4 + kk) * 1024 + 512 + lane * 8);
                    draw[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, bl, draw[u], 0, 0, 0);
                    draw[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, bh, draw[u], 0, 0, 0);
                    draw[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, bh, draw[u], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                f32x4 c = zero4();
#pragma unroll
                for (int j = 0; j < 8; ++j) c = mfma4(ld4(s_wt + ((u * 8 + j) * 64 + lane) * 4), dt[j], c);
                draw[u] = c;
            }
        }
#pragma unroll
        for (int l = NF; l < 5; ++l) {
            const int C = LV::C(l);
            const int nv = C / 16;
            const TapL tp = make_taps_l(gx, gy, G.W[l], ox[l], oy[l], G.fw[l]);
            float* gbase = (l < 3 ? a.dproj[l] : a.dfine[l - 3]) + img * (long)G.W[l] * G.W[l] * C + 4 * g;
            float* lbase = s_acc + G.off[l] + 4 * g;
            // Segmented reduction over the lane row: sorted neighbours that share their four tap pixels form
            // contiguous runs; sum w*d over each run with 4 predicated DPP shifts (Hillis-Steele restricted to
            // the run) and let the run's last lane issue ONE LDS add per value.  LDS float atomics cost ~3
            // cycles per lane, so the number of lanes that reach them is what matters.
            // A tap outside the map (queries clamped to the right / bottom border: x0 = W-1) carries weight 0 and is
            // simply not written: whether taps 1..3 exist depends only on tap 0's pixel, i.e. it is uniform over a run.
            const bool regular = tp.lofs[0] >= 0 && (!tp.ok[1] || tp.lofs[1] == tp.lofs[0] + 1) &&
                                 (!tp.ok[2] || tp.lofs[2] == tp.lofs[0] + G.fw[l]) &&
                                 (!tp.ok[3] || tp.lofs[3] == tp.lofs[0] + G.fw[l] + 1);
            const int key = regular ? tp.lofs[0] : -1 - m;
            const int key_prev = __builtin_amdgcn_update_dpp(-100, key, 0x111, 0xF, 0xF, false);   // row_shr:1
            const int key_next = __builtin_amdgcn_update_dpp(-100, key, 0x101, 0xF, 0xF, false);   // row_shl:1
            const bool head = m == 0 || key_prev != key, tail = m == 15 || key_next != key;
            // head position of my run: inclusive max-scan of (head ? m : 0)
            int hp = head ? m : 0;
            hp = max(hp, __builtin_amdgcn_update_dpp(0, hp, 0x111, 0xF, 0xF, false));
            hp = max(hp, __builtin_amdgcn_update_dpp(0, hp, 0x112, 0xF, 0xF, false));
            hp = max(hp, __builtin_amdgcn_update_dpp(0, hp, 0x114, 0xF, 0xF, false));
            hp = max(hp, __builtin_amdgcn_update_dpp(0, hp, 0x118, 0xF, 0xF, false));
            const float p1 = hp <= m - 1 ? 1.f : 0.f, p2 = hp <= m - 2 ? 1.f : 0.f, p4 = hp <= m - 4 ? 1.f : 0.f,
                        p8 = hp <= m - 8 ? 1.f : 0.f;
            if (regular) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float* o = lbase + (tp.ok[k] ? tp.lofs[k] : 0) * (C + SBT_PAD);
                    const bool wr = tail && tp.ok[k];
#pragma unroll
                    for (int j = 0; j < nv; ++j) {
                        f32x4 v = draw[LV::draw0(l) + j] * tp.w[k];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float x = v[i];
                            x = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xF, 0xF, false)), p1, x);
                            x = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xF, 0xF, false)), p2, x);
                            x = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xF, 0xF, false)), p4, x);
                            x = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xF, 0xF, false)), p8, x);
                            v[i] = x;
                        }
                        if (wr) atomic_add4(o + 16 * j, v);
                    }
                }
            } else if (qv) {   // a tap outside the map / footprint: per-tap handling
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (tp.w[k] == 0.f) continue;
                    if (tp.lofs[k] >= 0) {
                        float* o = lbase + tp.lofs[k] * (C + SBT_PAD);
#pragma unroll
                        for (int j = 0; j < nv; ++j) atomic_add4(o + 16 * j, draw[LV::draw0(l) + j] * tp.w[k]);
                    } else {   // rounding put the tap one pixel outside the footprint: global atomic
                        float* o = gbase + (long)tp.gofs[k] * C;
#pragma unroll
                        for (int j = 0; j < nv; ++j) atomic_add4(o + 16 * j, draw[LV::draw0(l) + j] * tp.w[k]);
                    }
                }
            }
        }
    }

    // ================= folded levels: dMap[pix][c] = sum_q w[q][pix] dX[q][c] on the fp32 MFMA =================
    f32x4 accm[FD::NMT];
#pragma unroll
    for (int i = 0; i < FD::NMT; ++i) accm[i] = zero4();
    for (long c0 = qs_lo; c0 < qs_hi; c0 += SBT_CHUNK) {
        __syncthreads();   // the previous chunk's taps have been consumed
        if (threadIdx.x < NF * SBT_CHUNK) {   // thread = (level, query of the chunk): its four taps -> LDS
            const int l = threadIdx.x / SBT_CHUNK, qq = threadIdx.x % SBT_CHUNK;
            const long qs = c0 + qq;
            int ti[4] = {-1, -1, -1, -1};
            float tw[4] = {0.f, 0.f, 0.f, 0.f};
            if (qs < qs_hi) {
                float gx, gy;
                project_slot(qs, gx, gy);
                const TapL tp = make_taps_l(gx, gy, G.W[l], ox[l], oy[l], G.fw[l]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!tp.ok[k] || tp.w[k] == 0.f) continue;
                    if (tp.lofs[k] >= 0) {
                        ti[k] = tp.lofs[k];
                        tw[k] = tp.w[k];
                    } else {   // rounding put the tap one pixel outside the footprint (rare): this thread adds the row itself
                        float* o = (l < 3 ? a.dproj[l] : a.dfine[l - 3]) + (img * (long)G.W[l] * G.W[l] + tp.gofs[k]) * 128;
                        const float* xr = a.dX + row_of(qs);
                        for (int c = 0; c < 128; ++c) unsafeAtomicAdd(o + c, xr[c] * tp.w[k]);
                    }
                }
            }
            *reinterpret_cast<int4*>(s_tidx + (l * SBT_CHUNK + qq) * 4) = int4{ti[0], ti[1], ti[2], ti[3]};
            st4(s_tw + (l * SBT_CHUNK + qq) * 4, f32x4{tw[0], tw[1], tw[2], tw[3]});
        }
        // B operands of the chunk's 16 k-steps: dX[query 4 step + g][channel 16 wave + m]
        float bfr[SBT_CHUNK / 4];
#pragma unroll
        for (int st = 0; st < SBT_CHUNK / 4; ++st) {
            long qs = c0 + 4 * st + g;
            qs = qs < qs_hi ? qs : qs_hi - 1;      // weights of slots past the end are zero
            bfr[st] = a.dX[row_of(qs) + 16 * wave + m];
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < SBT_CHUNK / 4; ++st) {
#pragma unroll
            for (int l = 0; l < NF; ++l) {
                const int4 ti = *reinterpret_cast<const int4*>(s_tidx + (l * SBT_CHUNK + 4 * st + g) * 4);
                const f32x4 tw = ld4(s_tw + (l * SBT_CHUNK + 4 * st + g) * 4);
#pragma unroll
                for (int k = 0; k < FD::nmt(l); ++k) {
                    const int pm = m + 16 * k;
                    const float wa = (ti.x == pm ? tw[0] : 0.f) + (ti.y == pm ? tw[1] : 0.f) + (ti.z == pm ? tw[2] : 0.f) +
                                     (ti.w == pm ? tw[3] : 0.f);
                    accm[FD::mt0(l) + k] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, bfr[st], accm[FD::mt0(l) + k], 0, 0, 0);
                }
            }
        }
    }
    // D[row = pixel 16k + 4g + i][col = channel 16 wave + m] -> one global atomic per touched element
#pragma unroll
    for (int l = 0; l < NF; ++l) {
        const int W = G.W[l], fw = G.fw[l];
        float* gmap = (l < 3 ? a.dproj[l] : a.dfine[l - 3]) + img * (long)W * W * 128 + 16 * wave + m;
#pragma unroll
        for (int k = 0; k < FD::nmt(l); ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pix = 16 * k + 4 * g + i;
                const float v = accm[FD::mt0(l) + k][i];
                if (pix < fw * fw && v != 0.f) {
                    const int y = oy[l] + pix / fw, x = ox[l] + pix % fw;
                    if (x < W && y < W) unsafeAtomicAdd(gmap + ((long)y * W + x) * 128, v);
                }
            }
    }
    __syncthreads();
    // flush the raw levels' footprints (skip untouched entries)
#pragma unroll
    for (int l = NF; l < 5; ++l) {
        const int C = G.C[l], fw = G.fw[l], W = G.W[l];
        float* gmap = (l < 3 ? a.dproj[l] : a.dfine[l - 3]) + img * (long)W * W * C;
        const int c4 = C >> 2, n4 = fw * fw * c4;
        for (int i = threadIdx.x; i < n4; i += SBT_THREADS) {
            const int pix = i / c4, c = (i - pix * c4) * 4;
            const f32x4 v = ld4(s_acc + G.off[l] + pix * (C + SBT_PAD) + c);
            if (v[0] == 0.f && v[1] == 0.f && v[2] == 0.f && v[3] == 0.f) continue;
            const int y = oy[l] + pix / fw, x = ox[l] + pix % fw;
            if (x < W && y < W) atomic_add4(gmap + ((long)y * W + x) * C + c, v);
        }
    }
}

template <int GT>
static int launch_sample_bwd_t(const SampleBwdArgs& a, hipStream_t stream) {
    using LV = SbLevels<GT>;
    using FD = SbFold<GT>;
    if (a.perm && a.bin_ends) {
        SbtGeom G;
        int off = 0;
        bool fits = true;
        for (int l = 0; l < 5; ++l) {
            G.W[l] = a.size >> (4 - l);
            G.C[l] = LV::C(l);
            G.fw[l] = (16 * (G.W[l] - 1) + 254) / 255 + 2;
            G.off[l] = off;
            if (l >= FD::NF) off += G.fw[l] * G.fw[l] * (G.C[l] + SBT_PAD);       // raw levels: LDS accumulators (padded pixel stride)
            else fits = fits && G.fw[l] * G.fw[l] <= 16 * FD::nmt(l);             // folded levels: register tiles
        }
        G.total = off;
        const size_t lds = (size_t)(LV::NU * 8 * 256 + off + 2 * FD::NF * SBT_CHUNK * 4) * sizeof(float);
        if (fits && lds <= 160 * 1024) {
            static std::atomic<unsigned long long> attr_done{0};
            TRY_RET(s3d_set_max_lds(attr_done, {(const void*)sample_bwd_tiled_kernel<GT, false>,
                                                (const void*)sample_bwd_tiled_kernel<GT, true>}, 160 * 1024));
            const long batch = a.groups / a.groups_per_batch;
            if (a.gxy) {
                const long total = batch * a.n_qry;
                hipLaunchKernelGGL(sbt_project_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256),
                                   0, stream, a, (int)batch);
                S3D_LAUNCH_CHECK();
            }
            if (a.ws34_t16)
                hipLaunchKernelGGL((sample_bwd_tiled_kernel<GT, true>), dim3((unsigned)(batch * a.n_slices * 256)), dim3(SBT_THREADS),
                                   lds, stream, a, G);
            else
                hipLaunchKernelGGL((sample_bwd_tiled_kernel<GT, false>), dim3((unsigned)(batch * a.n_slices * 256)), dim3(SBT_THREADS),
                                   lds, stream, a, G);
            S3D_LAUNCH_CHECK();
            return 0;
        }
    }
    const long blocks = a.groups < 4096 ? a.groups : 4096;
    hipLaunchKernelGGL(sample_bwd_kernel<GT>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    S3D_LAUNCH_CHECK();
    return 0;
}

int launch_sample_bwd(const SampleBwdArgs& a, hipStream_t stream) {
    if (a.groups <= 0) return 0;
    if (sample_bwd_dense_covers(a)) {
        if (a.gxy) {
            const long batch = a.groups / a.groups_per_batch, total = batch * a.n_qry;
            hipLaunchKernelGGL(sbt_project_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0,
                               stream, a, (int)batch);
            S3D_LAUNCH_CHECK();
        }
        const int r = launch_sample_bwd_dense(a, stream);
        if (r < 0) {
            s3d_set_error("sample_bwd_dense launch failed: %s", hipGetErrorString(hipGetLastError()));
            return 1;
        }
        if (r > 0) return 0;
    }
    return a.gt ? launch_sample_bwd_t<1>(a, stream) : launch_sample_bwd_t<0>(a, stream);
}

// ---- copy the first `cdst` columns of a [rows][csrc] matrix into a dense [rows][cdst] matrix
__global__ void copy_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int csrc,
                                 int cdst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * cdst) dst[i] = src[(i / cdst) * csrc + i % cdst];
}
int launch_copy_cols(const float* src, float* dst, int rows, int csrc, int cdst, hipStream_t stream) {
    hipLaunchKernelGGL(copy_cols_kernel, dim3((rows * cdst + 255) / 256), dim3(256), 0, stream, src, dst, rows, csrc,
                       cdst);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---- slice-embedding gradient (unet_custom.py:52-57): F0[b*ns+s] = W_a x5[b] + W_b emds[s] + bias
//      d emds[s][e] = sum_co W[co][512+e] * (sum over b, pixels of dF0[b*ns+s][pix][co])
// One workgroup per slice, 1024 threads: thread (co, pixel parity) adds its half of the B x npix rows eight loads at a time
// (one thread per co walking all rows serially was 0.5 ms of latency on 12 workgroups), fixed order throughout.
__global__ __launch_bounds__(1024) void emb_grad_kernel(const float* __restrict__ dF0, const float* __restrict__ w,
                                                        float* __restrict__ demds, int B, int ns, int npix) {
    __shared__ float v[2][512];
    const int s = blockIdx.x, co = threadIdx.x & 511, h = threadIdx.x >> 9;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* p = dF0 + ((long)(b * ns + s) * npix) * 512 + co;
        int k = h;
        for (; k + 14 < npix; k += 16) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = p[(long)(k + 2 * j) * 512];
            __builtin_amdgcn_sched_barrier(0);   // 8 loads in flight
            acc += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
        for (; k < npix; k += 2) acc += p[(long)k * 512];
    }
    v[h][co] = acc;
    __syncthreads();
    if (threadIdx.x < 128) {
        float a2 = 0.f;
        for (int c = 0; c < 512; ++c) a2 += (v[0][c] + v[1][c]) * w[(long)c * 640 + 512 + threadIdx.x];
        demds[s * 128 + threadIdx.x] = a2;
    }
}
int launch_emb_grad(const float* dF0, const float* w, float* demds, int B, int ns, int npix, hipStream_t stream) {
    hipLaunchKernelGGL(emb_grad_kernel, dim3(ns), dim3(1024), 0, stream, dF0, w, demds, B, ns, npix);
    S3D_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// LAST layer in training, absorbed form (round 4).  Only token 0 of the layer is consumed (models.py:83): for that one query
// row the K and V projections move off the 13 tokens, exactly as in inference (decode.h, launch_attn_last_mix) —
//   qt_h = M_h x0 + m_h,  s_t = (qt_h . x_t) / sqrt(32),  p = softmax_t(s),  pd = dropout(p),
//   xbar_h = sum_t pd_t x_t,  sigma_h = sum_t pd_t,   out = sum_h N_h xbar_h + sum_h sigma_h c_h + b_o
// with M_h = Wk_h^T Wq_h, m_h = Wk_h^T bq_h, N_h = Wo[:, h] Wv_h, c_h = Wo[:, h] bv_h (bk cancels in the softmax: its gradient is
// zero; sigma_h is 1 without dropout and carries bv through the probability dropout otherwise).  The stored-K|V form of rounds
// 2-3 ran 17 passes over the 5.2 M-row token tensor for this layer (K|V projection, core forward / backward, dK|V, its weight
// gradient, its data gradient: 9.9 ms of the step); here the tokens are read once forward, and read + written once backward.
// The matrices are formed in double at every step (absorb_train_kernel), the gradients of M, m, N, c come out of the ordinary
// weight-gradient kernels on the token-0 rows and are carried to in_proj / out_proj by absorb_grad_kernel (chain rule through
// the 128 x 128 products).
// xbar rows are S3D_ABS_NA = 544 wide: 4 x 128 mixed rows | 4 probability sums | zeros (a multiple of 32 for the GEMMs).
// 32 lanes per query (4 channels each): the query's token rows (and, backward, their gradients) stay in registers.
// =============================================================================================
__device__ __forceinline__ float half32_allsum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
    v += __shfl_xor(v, 16, 64);                                                                                           // the other row of the half
    return v;
}
template <int BWD>
__global__ __launch_bounds__(256) void attn_mix0_kernel(const float* __restrict__ X, const float* __restrict__ qt,
                                                        float* __restrict__ xbar, const float* __restrict__ dxbar,
                                                        float* __restrict__ dX, float* __restrict__ dqt, long groups, int T,
                                                        const DropCfg drop) {
    const int qi = threadIdx.x >> 5, c4 = (threadIdx.x & 31) * 4;
    const float scale = 0.17677669529663687f;   // 1/sqrt(32)
    for (long item = blockIdx.x; item < 2 * groups; item += gridDim.x) {
        const long grp = item >> 1;
        const int mq = (int)(item & 1) * 8 + qi;
        const long row0 = grp * S3D_GROUP + mq;
        const float* xg = X + (grp * T * S3D_GROUP + mq) * 128 + c4;
        f32x4 x[S3D_N_TOKENS_MAX], dx[S3D_N_TOKENS_MAX];
#pragma unroll
        for (int t = 0; t < S3D_N_TOKENS_MAX; ++t) {
            const int tc = t < T ? t : T - 1;
            x[t] = ld4(xg + (long)tc * S3D_GROUP * 128);
            dx[t] = zero4();
        }
#pragma unroll 1
        for (int h = 0; h < 4; ++h) {
            const f32x4 q4 = ld4(qt + row0 * 512 + h * 128 + c4);
            float p[S3D_N_TOKENS_MAX], mk[16];
            float mx = -1e30f;
#pragma unroll
            for (int t = 0; t < S3D_N_TOKENS_MAX; ++t) {
                float d = q4[0] * x[t][0] + q4[1] * x[t][1] + q4[2] * x[t][2] + q4[3] * x[t][3];
                d = half32_allsum(d) * scale;
                p[t] = t < T ? d : -1e30f;
                mx = fmaxf(mx, p[t]);
            }
            float den = 0.f;
#pragma unroll
            for (int t = 0; t < S3D_N_TOKENS_MAX; ++t) {
                p[t] = t < T ? expf(p[t] - mx) : 0.f;
                den += p[t];
            }
            const float inv = 1.f / den;
#pragma unroll
            for (int t = 0; t < S3D_N_TOKENS_MAX; ++t) p[t] *= inv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // the masks of keys 4j .. 4j + 3 (index: ((row0 * 4 + head) * 16 + key): the compact token-0 rows)
                float m4[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop.p > 0.f) s3d_drop4(drop, ((unsigned long long)row0 * 4 + h) * 16 + 4 * j, m4);
#pragma unroll
                for (int i = 0; i < 4; ++i) mk[4 * j + i] = m4[i];
            }
            if (!BWD) {
                f32x4 o = zero4();
                float sg = 0.f;
#pragma unroll
                for (int t = 0; t < S3D_N_TOKENS_MAX; ++t)
                    if (t < T) {
                        const float pd = p[t] * mk[t];
                        o += x[t] * pd;
                        sg += pd;
                    }
                st4(xbar + row0 * S3D_ABS_NA + h * 128 + c4, o);
                if (c4 == 0) xbar[row0 * S3D_ABS_NA + 512 + h] = sg;
            } else {
                const f32x4 g4 = ld4(dxbar + row0 * S3D_ABS_NA + h * 128 + c4);
                const float gs = dxbar[row0 * S3D_ABS_NA + 512 + h];
                float dp[S3D_N_TOKENS_MAX];
                float dot = 0.f;
#pragma unroll
                for (int t = 0; t < S3D_N_TOKENS_MAX; ++t) {
                    float d = g4[0] * x[t][0] + g4[1] * x[t][1] + g4[2] * x[t][2] + g4[3] * x[t][3];
                    d = (half32_allsum(d) + gs) * mk[t];   // d loss / d probability
                    dp[t] = t < T ? d : 0.f;
                    dot += p[t] * dp[t];
                }
                f32x4 dq4 = zero4();
#pragma unroll
                for (int t = 0; t < S3D_N_TOKENS_MAX; ++t)
                    if (t < T) {
                        const float ds = p[t] * (dp[t] - dot) * scale;
                        dq4 += x[t] * ds;
                        dx[t] += g4 * (p[t] * mk[t]) + q4 * ds;
                    }
                st4(dqt + row0 * 512 + h * 128 + c4, dq4);
            }
        }
        if (!BWD) {   // the zero padding of the row
            if (c4 >= 4 && c4 <= S3D_ABS_NA - 516) st4(xbar + row0 * S3D_ABS_NA + 512 + c4, zero4());
        } else {
            float* dg = dX + (grp * T * S3D_GROUP + mq) * 128 + c4;
#pragma unroll
            for (int t = 0; t < S3D_N_TOKENS_MAX; ++t)
                if (t < T) st4(dg + (long)t * S3D_GROUP * 128, dx[t]);
        }
    }
}
int launch_attn_mix0_fwd(const float* X, const float* qt, float* xbar, long groups, int T, const DropCfg& drop,
                         hipStream_t stream) {
    if (groups <= 0) return 0;
    S3D_CHECK_ARG(T >= 1 && T <= S3D_N_TOKENS_MAX, "attn mix0: T %d", T);
    const long nb = 2 * groups < 16384 ? 2 * groups : 16384;
    hipLaunchKernelGGL((attn_mix0_kernel<0>), dim3((unsigned)nb), dim3(256), 0, stream, X, qt, xbar, nullptr, nullptr, nullptr,
                       groups, T, drop);
    S3D_LAUNCH_CHECK();
    return 0;
}
// dX: every element of the [groups][T][16][128] token tensor is written (token 0's rows still lack the query path: dqt M)
int launch_attn_mix0_bwd(const float* X, const float* qt, const float* dxbar, float* dX, float* dqt, long groups, int T,
                         const DropCfg& drop, hipStream_t stream) {
    if (groups <= 0) return 0;
    S3D_CHECK_ARG(T >= 1 && T <= S3D_N_TOKENS_MAX, "attn mix0: T %d", T);
    const long nb = 2 * groups < 16384 ? 2 * groups : 16384;
    hipLaunchKernelGGL((attn_mix0_kernel<1>), dim3((unsigned)nb), dim3(256), 0, stream, X, qt, nullptr, dxbar, dX, dqt, groups,
                       T, drop);
    S3D_LAUNCH_CHECK();
    return 0;
}

// absorbed matrices of one step (double accumulation, rounded once):  M [512][128], m [512], Naug [128][S3D_ABS_NA]
__global__ void absorb_train_kernel(const float* __restrict__ in_w, const float* __restrict__ in_b,
                                    const float* __restrict__ out_w, float* __restrict__ M, float* __restrict__ mvec,
                                    float* __restrict__ Naug) {
    const float* Wq = in_w;
    const float* Wk = in_w + 128 * 128;
    const float* Wv = in_w + 256 * 128;
    const int total = 512 * 128 + 512 + 128 * S3D_ABS_NA;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        if (idx < 512 * 128) {            // M[h*128 + c][k] = sum_d Wk[32h+d][c] Wq[32h+d][k]
            const int r = idx >> 7, k = idx & 127, h = r >> 7, c = r & 127;
            double s = 0.0;
            for (int d = 0; d < 32; ++d) s += (double)Wk[(32 * h + d) * 128 + c] * (double)Wq[(32 * h + d) * 128 + k];
            M[idx] = (float)s;
        } else if (idx < 512 * 128 + 512) {   // m[h*128 + c] = sum_d Wk[32h+d][c] bq[32h+d]
            const int r = idx - 512 * 128, h = r >> 7, c = r & 127;
            double s = 0.0;
            for (int d = 0; d < 32; ++d) s += (double)Wk[(32 * h + d) * 128 + c] * (double)in_b[32 * h + d];
            mvec[r] = (float)s;
        } else {
            const int j = idx - 512 * 128 - 512, n = j / S3D_ABS_NA, kk = j - n * S3D_ABS_NA;
            double s = 0.0;
            if (kk < 512) {               // N[n][h*128 + c] = sum_d Wo[n][32h+d] Wv[32h+d][c]
                const int h = kk >> 7, c = kk & 127;
                for (int d = 0; d < 32; ++d) s += (double)out_w[n * 128 + 32 * h + d] * (double)Wv[(32 * h + d) * 128 + c];
            } else if (kk < 516) {        // c_h[n] = sum_d Wo[n][32h+d] bv[32h+d]
                const int h = kk - 512;
                for (int d = 0; d < 32; ++d) s += (double)out_w[n * 128 + 32 * h + d] * (double)in_b[256 + 32 * h + d];
            }
            Naug[j] = (float)s;
        }
    }
}
int launch_absorb_train(const float* in_w, const float* in_b, const float* out_w, float* M, float* mvec, float* Naug,
                        hipStream_t stream) {
    hipLaunchKernelGGL(absorb_train_kernel, dim3(512), dim3(256), 0, stream, in_w, in_b, out_w, M, mvec, Naug);
    S3D_LAUNCH_CHECK();
    return 0;
}
// chain rule from (dM, dm, dNaug) to in_proj (weight [384][128], bias [384]) and out_proj.weight [128][128]; every element
// of the three gradients is written (the key bias's gradient is zero)
__global__ void absorb_grad_kernel(const float* __restrict__ in_w, const float* __restrict__ in_b,
                                   const float* __restrict__ out_w, const float* __restrict__ dM,
                                   const float* __restrict__ dm, const float* __restrict__ dNaug,
                                   float* __restrict__ g_in_w, float* __restrict__ g_in_b, float* __restrict__ g_out_w) {
    const float* Wq = in_w;
    const float* Wk = in_w + 128 * 128;
    const float* Wv = in_w + 256 * 128;
    const int total = 384 * 128 + 384 + 128 * 128;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        double s = 0.0;
        if (idx < 128 * 128) {                    // dWq[32h+d][k] = sum_c Wk[32h+d][c] dM[h*128+c][k]
            const int r = idx >> 7, k = idx & 127, h = r >> 5;
            for (int c = 0; c < 128; ++c) s += (double)Wk[r * 128 + c] * (double)dM[(h * 128 + c) * 128 + k];
            g_in_w[idx] = (float)s;
        } else if (idx < 256 * 128) {             // dWk[32h+d][c] = sum_k dM[h*128+c][k] Wq[32h+d][k] + dm[h*128+c] bq[32h+d]
            const int j = idx - 128 * 128, r = j >> 7, c = j & 127, h = r >> 5;
            for (int k = 0; k < 128; ++k) s += (double)dM[(h * 128 + c) * 128 + k] * (double)Wq[r * 128 + k];
            s += (double)dm[h * 128 + c] * (double)in_b[r];
            g_in_w[idx] = (float)s;
        } else if (idx < 384 * 128) {             // dWv[32h+d][c] = sum_n Wo[n][32h+d] dN[n][h*128+c]
            const int j = idx - 256 * 128, r = j >> 7, c = j & 127, h = r >> 5;
            for (int n = 0; n < 128; ++n) s += (double)out_w[n * 128 + r] * (double)dNaug[n * S3D_ABS_NA + h * 128 + c];
            g_in_w[idx] = (float)s;
        } else if (idx < 384 * 128 + 384) {
            const int r = idx - 384 * 128;
            if (r < 128) {                        // dbq[32h+d] = sum_c Wk[32h+d][c] dm[h*128+c]
                const int h = r >> 5;
                for (int c = 0; c < 128; ++c) s += (double)Wk[r * 128 + c] * (double)dm[h * 128 + c];
            } else if (r >= 256) {                // dbv[32h+d] = sum_n Wo[n][32h+d] dc_h[n]
                const int rv = r - 256, h = rv >> 5;
                for (int n = 0; n < 128; ++n) s += (double)out_w[n * 128 + rv] * (double)dNaug[n * S3D_ABS_NA + 512 + h];
            }
            g_in_b[r] = (float)s;
        } else {                                  // dWo[n][32h+d] = sum_c dN[n][h*128+c] Wv[32h+d][c] + dc_h[n] bv[32h+d]
            const int j = idx - 384 * 128 - 384, n = j >> 7, r = j & 127, h = r >> 5;
            for (int c = 0; c < 128; ++c) s += (double)dNaug[n * S3D_ABS_NA + h * 128 + c] * (double)Wv[r * 128 + c];
            s += (double)dNaug[n * S3D_ABS_NA + 512 + h] * (double)in_b[256 + r];
            g_out_w[j] = (float)s;
        }
    }
}
int launch_absorb_grad(const float* in_w, const float* in_b, const float* out_w, const float* dM, const float* dm,
                       const float* dNaug, float* g_in_w, float* g_in_b, float* g_out_w, hipStream_t stream) {
    hipLaunchKernelGGL(absorb_grad_kernel, dim3(512), dim3(256), 0, stream, in_w, in_b, out_w, dM, dm, dNaug, g_in_w, g_in_b,
                       g_out_w);
    S3D_LAUNCH_CHECK();
    return 0;
}

