// mesh.hip — device-side MISE refinement and marching cubes.
//
//   MISE: replaces reg_slices/src_convonet/utils/libmise/mise.pyx:33-368 (query / update / to_dense) as driven by
//         Generator3D.generate_from_latent (reconstruct.py:148-173).  The octree is kept as dense per-level state
//         grids in HBM; points / values never leave the device between rounds (the reference — and this package's
//         host C++ port in csrc_mesh/ — move every round's points H2D and values D2H in float64).
//   marching cubes: replaces libmcubes.marching_cubes (libmcubes/pywrapper.cpp:90-127, marchingcubes.h:23-193) as
//         classify -> scan -> emit, with the reference's cell traversal order, vertex ownership (a cell creates the
//         vertices of its edges 6, 5, 10 and of the edges whose owner lies outside the grid), per-cell creation order
//         and interpolation expression, so vertices and faces come out bit-identical and in the same numbering.
//
// Equivalence notes (pinned by tests/test_gpu_mesh.py against goldens made with the reference's compiled libraries):
//   * which points a round queries does not depend on the ORDER the previous round's points were inserted in; the
//     device version returns a round's points in ascending grid index instead of the reference's insertion order —
//     the same SET every round, the same dense grid at the end.
//   * a leaf is split when the known points inside its closed box hold both (value >= thr) and (value <= thr)
//     (mise.pyx:182-232); leaves created by a split wait for the next update, as in the reference.
#include <math.h>
#include <string.h>
#include <mutex>

#include "common.h"

// =============================================================================================
// ordered stream compaction of a predicate over [0, n)  (used by MISE query and by nothing else)
// =============================================================================================
#define CMP_ITEMS 16              // consecutive items per thread
#define CMP_BLOCK 256
#define CMP_TILE (CMP_ITEMS * CMP_BLOCK)

__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* s_tmp, unsigned& total) {
    // s_tmp: CMP_BLOCK / 64 + 1 words
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_tmp[wave] = x;
    __syncthreads();
    unsigned base = 0, tot = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        const unsigned t = s_tmp[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return base + x - v;
}

// =============================================================================================
// MISE
// =============================================================================================
struct MiseDev {
    int res0, depth, res, r1;       // r1 = res + 1 points per axis
    double thr;
    long n_pts;                     // r1^3
    double* val;                    // [n_pts]
    unsigned char* pstate;          // [n_pts]  0 none, 1 exists (value unknown), 2 known
    unsigned char* vstate[13];      // per level l: [(res0 << l)^3]  0 none, 1 leaf, 2 split, 3 leaf marked for splitting
    unsigned* blk;                  // compaction scratch: per-tile counts / offsets, [n_tiles + 1]
    long n_tiles;
};

static size_t mise_layout(int res0, int depth, MiseDev* m, char* base) {
    size_t off = align_up(sizeof(MiseDev), 256);
    const long r1 = (long)(res0 << depth) + 1;
    const long n = r1 * r1 * r1;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    const size_t o_val = take((size_t)n * 8), o_ps = take((size_t)n);
    size_t o_vs[13];
    for (int l = 0; l <= depth; ++l) {
        const long s = (long)res0 << l;
        o_vs[l] = take((size_t)(s * s * s));
    }
    const long tiles = (n + CMP_TILE - 1) / CMP_TILE;
    const size_t o_blk = take((size_t)(tiles + 1) * 4);
    if (m) {
        m->res0 = res0; m->depth = depth; m->res = res0 << depth; m->r1 = (int)r1; m->n_pts = n; m->n_tiles = tiles;
        m->val = (double*)(base + o_val);
        m->pstate = (unsigned char*)(base + o_ps);
        for (int l = 0; l <= depth; ++l) m->vstate[l] = (unsigned char*)(base + o_vs[l]);
        m->blk = (unsigned*)(base + o_blk);
    }
    return off;
}

extern "C" size_t s3d_mise_dev_workspace_bytes(int resolution0, int depth) {
    if (resolution0 < 1 || depth < 0 || depth > 12 || ((long)resolution0 << depth) > 2047) return 0;
    return mise_layout(resolution0, depth, nullptr, nullptr);
}

__global__ void mise_init_points_kernel(MiseDev m) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m.n_pts) return;
    const int r1 = m.r1, vs0 = 1 << m.depth;
    const int z = (int)(i % r1), y = (int)((i / r1) % r1), x = (int)(i / ((long)r1 * r1));
    m.pstate[i] = (x % vs0 == 0 && y % vs0 == 0 && z % vs0 == 0) ? 1 : 0;
    m.val[i] = __longlong_as_double(0x7ff8000000000000LL);
}
__global__ void fill_bytes_kernel(unsigned char* p, long n, unsigned char v) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// Opaque handle = host-side descriptor of a caller-owned device workspace (the library keeps no global state).
extern "C" void* s3d_mise_dev_create(void* workspace, size_t workspace_bytes, int resolution0, int depth,
                                     double threshold, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const size_t need = s3d_mise_dev_workspace_bytes(resolution0, depth);
    if (!workspace || need == 0 || workspace_bytes < need) {
        s3d_set_error("mise_dev_create: resolution0=%d depth=%d, workspace %zu bytes (need %zu)", resolution0, depth,
                      workspace_bytes, need);
        return nullptr;
    }
    MiseDev* m = new MiseDev();
    memset(m, 0, sizeof(*m));
    mise_layout(resolution0, depth, m, (char*)workspace);
    m->thr = threshold;
    hipLaunchKernelGGL(mise_init_points_kernel, dim3((unsigned)((m->n_pts + 255) / 256)), dim3(256), 0, st, *m);
    for (int l = 0; l <= depth; ++l) {
        const long s = (long)resolution0 << l, n = s * s * s;
        hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m->vstate[l], n,
                           (unsigned char)(l == 0 ? 1 : 0));
    }
    if (hipGetLastError() != hipSuccess) {
        s3d_set_error("mise_dev_create: launch failed");
        delete m;
        return nullptr;
    }
    return m;
}
extern "C" void s3d_mise_dev_destroy(void* handle) { delete (MiseDev*)handle; }
extern "C" int s3d_mise_dev_resolution(void* handle) { return handle ? ((MiseDev*)handle)->res : 0; }
static MiseDev* mise_find(void* handle) { return (MiseDev*)handle; }

// ---- query: ordered compaction of {i : pstate[i] == 1} ----
__global__ __launch_bounds__(CMP_BLOCK) void mise_count_kernel(MiseDev m) {
    __shared__ unsigned s_tmp[CMP_BLOCK / 64 + 1];
    const long base = (long)blockIdx.x * CMP_TILE + (long)threadIdx.x * CMP_ITEMS;
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < CMP_ITEMS; ++k) {
        const long i = base + k;
        c += (i < m.n_pts && m.pstate[i] == 1) ? 1u : 0u;
    }
    unsigned total;
    block_exclusive_scan(c, s_tmp, total);
    if (threadIdx.x == 0) m.blk[blockIdx.x] = total;
}
// single block: exclusive scan of n words in place, total -> p[n]
__global__ __launch_bounds__(1024) void scan_words_kernel(unsigned* p, long n) {
    __shared__ unsigned s_tmp[1024 / 64 + 1];
    __shared__ unsigned s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (long b = 0; b < n; b += 1024) {
        const long i = b + threadIdx.x;
        const unsigned v = i < n ? p[i] : 0u;
        unsigned total;
        const unsigned ex = block_exclusive_scan(v, s_tmp, total);
        const unsigned carry = s_carry;
        if (i < n) p[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) p[n] = s_carry;
}
__global__ __launch_bounds__(CMP_BLOCK) void mise_emit_kernel(MiseDev m, int* idx_out, long capacity) {
    __shared__ unsigned s_tmp[CMP_BLOCK / 64 + 1];
    const long base = (long)blockIdx.x * CMP_TILE + (long)threadIdx.x * CMP_ITEMS;
    unsigned c = 0;
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < CMP_ITEMS; ++k) {
        const long i = base + k;
        const bool on = i < m.n_pts && m.pstate[i] == 1;
        bits |= (on ? 1u : 0u) << k;
        c += on ? 1u : 0u;
    }
    unsigned total;
    unsigned pos = m.blk[blockIdx.x] + block_exclusive_scan(c, s_tmp, total);
#pragma unroll
    for (int k = 0; k < CMP_ITEMS; ++k)
        if (bits & (1u << k)) {
            if ((long)pos < capacity) idx_out[pos] = (int)(base + k);
            ++pos;
        }
}

// Points of the next round: linear grid indices ((r1*x + y)*r1 + z), ascending.  Synchronises the stream to return the
// count; writes at most `capacity` indices (call again with a larger buffer if *n_out > capacity).
extern "C" int s3d_mise_dev_query(void* workspace, int* idx_out, long capacity, long* n_out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    MiseDev* mp = mise_find(workspace);
    S3D_CHECK_ARG(mp && n_out, "mise_dev_query: workspace not initialised");
    const MiseDev m = *mp;
    hipLaunchKernelGGL(mise_count_kernel, dim3((unsigned)m.n_tiles), dim3(CMP_BLOCK), 0, st, m);
    hipLaunchKernelGGL(scan_words_kernel, dim3(1), dim3(1024), 0, st, m.blk, m.n_tiles);
    unsigned total = 0;
    hipError_t e = hipMemcpyAsync(&total, m.blk + m.n_tiles, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        s3d_set_error("mise_dev_query: %s", hipGetErrorString(e));
        return (int)e;
    }
    *n_out = (long)total;
    if (total && idx_out && capacity > 0)
        hipLaunchKernelGGL(mise_emit_kernel, dim3((unsigned)m.n_tiles), dim3(CMP_BLOCK), 0, st, m, idx_out, capacity);
    S3D_LAUNCH_CHECK();
    return 0;
}

// query coordinates of grid points, exactly as reconstruct.py:160-161 forms them in float32:
//   pointsf = box_size * (points.astype(float32) / resolution - 0.5)
__global__ void mise_points_kernel(const int* __restrict__ idx, long n, int r1, float res, float box,
                                   float* __restrict__ qry) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int li = idx[i];
    const int z = li % r1, y = (li / r1) % r1, x = li / (r1 * r1);
    qry[3 * i + 0] = box * (__fdiv_rn((float)x, res) - 0.5f);
    qry[3 * i + 1] = box * (__fdiv_rn((float)y, res) - 0.5f);
    qry[3 * i + 2] = box * (__fdiv_rn((float)z, res) - 0.5f);
}
extern "C" int s3d_mise_dev_points(void* workspace, const int* idx, long n, float box, float* qry_out, void* stream) {
    MiseDev* mp = mise_find(workspace);
    S3D_CHECK_ARG(mp && idx && qry_out, "mise_dev_points: bad argument");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mise_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, n,
                       mp->r1, (float)mp->res, box, qry_out);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---- update: scatter the round's values, then one refinement step ----
template <typename T>
__global__ void mise_scatter_kernel(MiseDev m, const int* __restrict__ idx, const T* __restrict__ v, long n, int* bad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long li = idx[i];
    if (li < 0 || li >= m.n_pts || m.pstate[li] == 0) {   // not a grid point of the octree (the reference raises)
        *bad = 1;
        return;
    }
    m.val[li] = (double)v[i];
    m.pstate[li] = 2;
}
// leaves of level l (l < depth) whose closed box holds known values on both sides of the threshold -> state 3
__global__ void mise_flag_kernel(MiseDev m, int l) {
    const int s = m.res0 << l;                       // voxels per axis at this level
    const long nv = (long)s * s * s;
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    unsigned char* vs = m.vstate[l];
    if (vs[v] != 1) return;
    const int size = 1 << (m.depth - l);             // edge length in finest cells
    const int vz = (int)(v % s), vy = (int)((v / s) % s), vx = (int)(v / ((long)s * s));
    const int x0 = vx * size, y0 = vy * size, z0 = vz * size;
    const long r1 = m.r1;
    unsigned f = 0;
    for (int i = 0; i <= size && f != 3; ++i)
        for (int j = 0; j <= size && f != 3; ++j) {
            const long row = ((x0 + i) * r1 + (y0 + j)) * r1 + z0;
            for (int k = 0; k <= size; ++k)
                if (m.pstate[row + k] == 2) {
                    const double val = m.val[row + k];
                    f |= (val >= m.thr ? 1u : 0u) | (val <= m.thr ? 2u : 0u);
                }
        }
    if (f == 3) vs[v] = 3;
}
// apply the marked splits: parent -> split, 8 children -> leaves, the 27 lattice points of the parent exist
__global__ void mise_split_kernel(MiseDev m, int l) {
    const int s = m.res0 << l;
    const long nv = (long)s * s * s;
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    unsigned char* vs = m.vstate[l];
    if (vs[v] != 3) return;
    vs[v] = 2;
    const int vz = (int)(v % s), vy = (int)((v / s) % s), vx = (int)(v / ((long)s * s));
    unsigned char* cs = m.vstate[l + 1];
    const long s2 = 2L * s;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) cs[((2L * vx + i) * s2 + (2 * vy + j)) * s2 + (2 * vz + k)] = 1;
    const int half = 1 << (m.depth - l - 1);
    const long r1 = m.r1;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                const long p = (((long)(2 * vx + i) * half) * r1 + (long)(2 * vy + j) * half) * r1 + (long)(2 * vz + k) * half;
                if (m.pstate[p] == 0) m.pstate[p] = 1;   // racing writers all store 1
            }
}

template <typename T>
static int mise_update_impl(void* workspace, const int* idx, const T* values, long n, hipStream_t st) {
    MiseDev* mp = mise_find(workspace);
    S3D_CHECK_ARG(mp && (n == 0 || (idx && values)), "mise_dev_update: bad argument");
    const MiseDev m = *mp;
    int* bad = (int*)(m.blk + m.n_tiles);      // the scan's total word doubles as the error flag between queries
    if (n > 0) {
        (void)hipMemsetAsync(bad, 0, 4, st);
        hipLaunchKernelGGL(mise_scatter_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m, idx, values,
                           n, bad);
    }
    for (int l = 0; l < m.depth; ++l) {
        const long s = (long)m.res0 << l, nv = s * s * s;
        hipLaunchKernelGGL(mise_flag_kernel, dim3((unsigned)((nv + 127) / 128)), dim3(128), 0, st, m, l);
    }
    for (int l = 0; l < m.depth; ++l) {
        const long s = (long)m.res0 << l, nv = s * s * s;
        hipLaunchKernelGGL(mise_split_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, m, l);
    }
    S3D_LAUNCH_CHECK();
    if (n > 0) {
        int h_bad = 0;
        hipError_t e = hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            s3d_set_error("mise_dev_update: %s", hipGetErrorString(e));
            return (int)e;
        }
        if (h_bad) {
            s3d_set_error("mise_dev_update: a point is not in the grid");
            return S3D_E_ARG;
        }
    }
    return 0;
}
// values: the network's logits at the points of `idx` (float32, as Generator3D.eval_points returns them)
extern "C" int s3d_mise_dev_update(void* workspace, const int* idx, const float* values, long n, void* stream) {
    return mise_update_impl<float>(workspace, idx, values, n, (hipStream_t)stream);
}
extern "C" int s3d_mise_dev_update_f64(void* workspace, const int* idx, const double* values, long n, void* stream) {
    return mise_update_impl<double>(workspace, idx, values, n, (hipStream_t)stream);
}

// ---- to_dense: unknown entries forward-filled along x, then y, then z (mise.pyx:131-163) ----
__global__ void mise_dense_copy_kernel(MiseDev m, double* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m.n_pts) out[i] = m.pstate[i] == 2 ? m.val[i] : __longlong_as_double(0x7ff8000000000000LL);
}
__global__ void fill_axis_kernel(double* out, int r1, int axis) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)r1 * r1) return;
    const int a = (int)(t / r1), b = (int)(t % r1);
    long base, step;
    if (axis == 0) { base = (long)a * r1 + b; step = (long)r1 * r1; }        // (j, k) lines along i
    else if (axis == 1) { base = (long)a * r1 * r1 + b; step = r1; }          // (i, k) lines along j
    else { base = ((long)a * r1 + b) * r1; step = 1; }                        // (i, j) lines along k
    double prev = out[base];
    for (int s = 1; s < r1; ++s) {
        const double v = out[base + s * step];
        if (v != v) out[base + s * step] = prev;
        else prev = v;
    }
}
extern "C" int s3d_mise_dev_to_dense(void* workspace, double* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    MiseDev* mp = mise_find(workspace);
    S3D_CHECK_ARG(mp && out, "mise_dev_to_dense: bad argument");
    const MiseDev m = *mp;
    hipLaunchKernelGGL(mise_dense_copy_kernel, dim3((unsigned)((m.n_pts + 255) / 256)), dim3(256), 0, st, m, out);
    const long lines = (long)m.r1 * m.r1;
    for (int axis = 0; axis < 3; ++axis)
        hipLaunchKernelGGL(fill_axis_kernel, dim3((unsigned)((lines + 63) / 64)), dim3(64), 0, st, out, m.r1, axis);
    S3D_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// marching cubes
// =============================================================================================
static const char* const kCasesHost[256] = {
#include "../csrc_mesh/mc_cases.inc"
};
struct McTables {
    signed char tri[256][16];     // edge ids, three per triangle, -1 terminated
    unsigned short emask[256];    // edges used by the case
    unsigned char ntri[256];
};
__constant__ McTables c_mc;
// cube corners, edges (a, b, axis, owner offset di dj dk, owner slot) and per-cell creation order: see csrc_mesh/mesh.cpp
__constant__ signed char c_corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
__constant__ signed char c_edge[12][7] = {{0, 1, 0, 0, -1, -1, 0}, {1, 2, 1, 0, 0, -1, 1}, {2, 3, 0, 0, 0, -1, 0}, {3, 0, 1, -1, 0, -1, 1},
                                          {4, 5, 0, 0, -1, 0, 0},  {5, 6, 1, 0, 0, 0, 1},  {6, 7, 0, 0, 0, 0, 0},  {7, 4, 1, -1, 0, 0, 1},
                                          {0, 4, 2, -1, -1, 0, 2}, {1, 5, 2, 0, -1, 0, 2}, {2, 6, 2, 0, 0, 0, 2},  {3, 7, 2, -1, 0, 0, 2}};
__constant__ signed char c_visit[12] = {6, 5, 10, 0, 1, 2, 3, 4, 7, 8, 9, 11};

// __constant__ memory is per device: the upload is tracked per device id, under a mutex (a process that drives two GPUs,
// or two threads, must not see a zeroed table on the second one)
static int mc_tables_ready() {
    static std::mutex mu;
    static int state[64] = {};   // per device: 0 not yet, 1 ok, <0 error
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    std::lock_guard<std::mutex> lock(mu);
    if (state[dev]) return state[dev];
    McTables t;
    memset(&t, 0, sizeof(t));
    for (int c = 0; c < 256; ++c) {
        int n = 0;
        for (const char* p = kCasesHost[c]; *p; ++p) {
            const int e = *p <= '9' ? *p - '0' : *p - 'a' + 10;
            t.tri[c][n++] = (signed char)e;
            t.emask[c] |= (unsigned short)(1u << e);
        }
        t.ntri[c] = (unsigned char)(n / 3);
        for (; n < 16; ++n) t.tri[c][n] = -1;
    }
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_mc), &t, sizeof(t));
    state[dev] = e == hipSuccess ? 1 : -1;
    return state[dev];
}

struct McDev {
    const void* grid;       // (gx, gy, gz) values, C order
    int is_f64;
    int gx, gy, gz;         // stored grid
    int pad;                // 0 / 1 layers of pad_value around it (np.pad(occ_hat, 1, constant_values=-1e6))
    double pad_value, iso;
    int nx, ny, nz;         // logical (padded) grid
    int cx, cy, cz;         // cells
    long n_cells, n_blocks;
    unsigned char* cube;    // [n_cells] case index
    unsigned* voff;         // [n_cells] first vertex id created by the cell
    unsigned* toff;         // [n_cells] first triangle of the cell
    unsigned* bsum;         // [2][n_blocks + 1] per-block vertex / triangle counts -> offsets
};
#define MC_BLOCK 256

__device__ __forceinline__ double mc_at(const McDev& d, int x, int y, int z) {
    x -= d.pad; y -= d.pad; z -= d.pad;
    if (x < 0 || y < 0 || z < 0 || x >= d.gx || y >= d.gy || z >= d.gz) return d.pad_value;
    const long i = ((long)x * d.gy + y) * d.gz + z;
    return d.is_f64 ? ((const double*)d.grid)[i] : (double)((const float*)d.grid)[i];
}
// vertices the cell creates itself: edges of its case that it owns (6, 5, 10) or whose owner lies outside the grid
__device__ __forceinline__ unsigned mc_created_mask(unsigned em, int i, int j, int k) {
    unsigned m = 0;
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        const bool owned = !c_edge[e][3] && !c_edge[e][4] && !c_edge[e][5];
        const bool boundary = (c_edge[e][3] && i == 0) || (c_edge[e][4] && j == 0) || (c_edge[e][5] && k == 0);
        if ((em >> e & 1u) && (owned || boundary)) m |= 1u << e;
    }
    return m;
}
__global__ __launch_bounds__(MC_BLOCK) void mc_classify_kernel(McDev d) {
    __shared__ unsigned s_tmp[MC_BLOCK / 64 + 1];
    const long c = (long)blockIdx.x * MC_BLOCK + threadIdx.x;
    unsigned nv = 0, nt = 0;
    if (c < d.n_cells) {
        const int k = (int)(c % d.cz), j = (int)((c / d.cz) % d.cy), i = (int)(c / ((long)d.cz * d.cy));
        unsigned cube = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (mc_at(d, i + c_corner[q][0], j + c_corner[q][1], k + c_corner[q][2]) <= d.iso) cube |= 1u << q;
        d.cube[c] = (unsigned char)cube;
        const unsigned em = c_mc.emask[cube];
        if (em) {
            nv = __popc(mc_created_mask(em, i, j, k));
            nt = c_mc.ntri[cube];
        }
    }
    unsigned tv, tt;
    const unsigned ev = block_exclusive_scan(nv, s_tmp, tv);
    const unsigned et = block_exclusive_scan(nt, s_tmp, tt);
    if (c < d.n_cells) {
        d.voff[c] = ev;      // block-local for now; mc_offsets_kernel adds the block's base
        d.toff[c] = et;
    }
    if (threadIdx.x == 0) {
        d.bsum[blockIdx.x] = tv;
        d.bsum[d.n_blocks + 1 + blockIdx.x] = tt;
    }
}
__global__ void mc_offsets_kernel(McDev d) {
    const long c = (long)blockIdx.x * MC_BLOCK + threadIdx.x;
    if (c >= d.n_cells) return;
    d.voff[c] += d.bsum[blockIdx.x];
    d.toff[c] += d.bsum[d.n_blocks + 1 + blockIdx.x];
}

__global__ __launch_bounds__(MC_BLOCK) void mc_emit_kernel(McDev d, double* __restrict__ verts, long long* __restrict__ tris) {
#pragma clang fp contract(off)
    const long c = (long)blockIdx.x * MC_BLOCK + threadIdx.x;
    if (c >= d.n_cells) return;
    const unsigned cube = d.cube[c];
    const unsigned em = c_mc.emask[cube];
    if (!em) return;
    const int k = (int)(c % d.cz), j = (int)((c / d.cz) % d.cy), i = (int)(c / ((long)d.cz * d.cy));
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = mc_at(d, i + c_corner[q][0], j + c_corner[q][1], k + c_corner[q][2]);
    const unsigned created = mc_created_mask(em, i, j, k);
    const long v0 = d.voff[c];
    long long id[12];
    unsigned rank = 0;
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        const int e = c_visit[t];
        if (!(em >> e & 1u)) continue;
        const int a = c_edge[e][0], b = c_edge[e][1], axis = c_edge[e][2];
        if (!(created >> e & 1u)) {   // created by the owner cell: its edges 6, 5, 10 come first in its creation order
            const long oc = ((long)(i + c_edge[e][3]) * d.cy + (j + c_edge[e][4])) * d.cz + (k + c_edge[e][5]);
            const unsigned oem = c_mc.emask[d.cube[oc]];
            const int slot = c_edge[e][6];
            const unsigned r = slot == 0 ? 0u : slot == 1 ? (oem >> 6 & 1u) : (oem >> 6 & 1u) + (oem >> 5 & 1u);
            id[e] = (long long)d.voff[oc] + r;
            continue;
        }
        id[e] = v0 + rank;
        double p[3] = {i + c_corner[a][0] + 0.5, j + c_corner[a][1] + 0.5, k + c_corner[a][2] + 0.5};
        const double x1 = p[axis];
        const double x2 = (axis == 0 ? i : axis == 1 ? j : k) + c_corner[b][axis] + 0.5;
        const double f1 = v[a], f2 = v[b];
        p[axis] = f2 == f1 ? (x2 + x1) / 2 : (x2 - x1) * (d.iso - f1) / (f2 - f1) + x1;
        double* o = verts + 3 * (v0 + rank);
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        ++rank;
    }
    long long* t = tris + 3 * (long)d.toff[c];
    for (int n = 0; n < 15 && c_mc.tri[cube][n] >= 0; ++n) t[n] = id[c_mc.tri[cube][n]];
}

static size_t mc_layout(int nx, int ny, int nz, McDev* d, char* base) {
    const long cells = (long)(nx - 1) * (ny - 1) * (nz - 1);
    const long blocks = (cells + MC_BLOCK - 1) / MC_BLOCK;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    const size_t o_cube = take((size_t)cells), o_v = take((size_t)cells * 4), o_t = take((size_t)cells * 4);
    const size_t o_b = take((size_t)(2 * (blocks + 1)) * 4);
    if (d) {
        d->n_cells = cells; d->n_blocks = blocks;
        d->cube = (unsigned char*)(base + o_cube);
        d->voff = (unsigned*)(base + o_v);
        d->toff = (unsigned*)(base + o_t);
        d->bsum = (unsigned*)(base + o_b);
    }
    return off;
}
// nx, ny, nz: the STORED grid; pad != 0 adds one layer of pad_value on every side (reconstruct.py:189)
extern "C" size_t s3d_mc_dev_workspace_bytes(int nx, int ny, int nz, int pad) {
    const int p = pad ? 2 : 0;
    if (nx + p < 2 || ny + p < 2 || nz + p < 2) return 256;
    return mc_layout(nx + p, ny + p, nz + p, nullptr, nullptr);
}

static int mc_make(McDev& d, const void* grid, int is_f64, int nx, int ny, int nz, int pad, double pad_value, double iso,
                   void* workspace, size_t workspace_bytes) {
    S3D_CHECK_ARG(grid && workspace && nx >= 1 && ny >= 1 && nz >= 1, "mc_dev: bad argument");
    S3D_CHECK_ARG((long)nx * ny * nz < (1L << 31), "mc_dev: grid too large");
    if (workspace_bytes < s3d_mc_dev_workspace_bytes(nx, ny, nz, pad)) {
        s3d_set_error("mc_dev: workspace %zu < %zu bytes", workspace_bytes, s3d_mc_dev_workspace_bytes(nx, ny, nz, pad));
        return S3D_E_WORKSPACE;
    }
    memset(&d, 0, sizeof(d));
    d.grid = grid; d.is_f64 = is_f64; d.gx = nx; d.gy = ny; d.gz = nz; d.pad = pad ? 1 : 0;
    d.pad_value = pad_value; d.iso = iso;
    d.nx = nx + 2 * d.pad; d.ny = ny + 2 * d.pad; d.nz = nz + 2 * d.pad;
    d.cx = d.nx - 1; d.cy = d.ny - 1; d.cz = d.nz - 1;
    if (d.cx < 1 || d.cy < 1 || d.cz < 1) {
        d.n_cells = 0;
        return 0;
    }
    mc_layout(d.nx, d.ny, d.nz, &d, (char*)workspace);
    return 0;
}

// Pass 1: classify + scan.  Returns the vertex / triangle counts (synchronises the stream); the workspace then holds
// what s3d_mc_dev_emit needs.  grid: device pointer, float32 (is_f64 = 0) or float64.
extern "C" int s3d_mc_dev_count(const void* grid, int is_f64, int nx, int ny, int nz, int pad, double pad_value,
                                double iso, void* workspace, size_t workspace_bytes, long* n_vertices,
                                long* n_triangles, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    S3D_CHECK_ARG(n_vertices && n_triangles, "mc_dev_count: null output");
    S3D_CHECK_ARG(mc_tables_ready() == 1, "mc_dev_count: case tables could not be uploaded");
    McDev d;
    TRY_RET(mc_make(d, grid, is_f64, nx, ny, nz, pad, pad_value, iso, workspace, workspace_bytes));
    *n_vertices = *n_triangles = 0;
    if (d.n_cells == 0) return 0;
    hipLaunchKernelGGL(mc_classify_kernel, dim3((unsigned)d.n_blocks), dim3(MC_BLOCK), 0, st, d);
    hipLaunchKernelGGL(scan_words_kernel, dim3(1), dim3(1024), 0, st, d.bsum, d.n_blocks);
    hipLaunchKernelGGL(scan_words_kernel, dim3(1), dim3(1024), 0, st, d.bsum + d.n_blocks + 1, d.n_blocks);
    hipLaunchKernelGGL(mc_offsets_kernel, dim3((unsigned)d.n_blocks), dim3(MC_BLOCK), 0, st, d);
    S3D_LAUNCH_CHECK();
    unsigned tot[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(&tot[0], d.bsum + d.n_blocks, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&tot[1], d.bsum + 2 * d.n_blocks + 1, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        s3d_set_error("mc_dev_count: %s", hipGetErrorString(e));
        return (int)e;
    }
    *n_vertices = tot[0];
    *n_triangles = tot[1];
    return 0;
}
// Pass 2 (same arguments as pass 1): vertices (V,3) float64 in index units + 0.5 of the padded grid (libmcubes'
// convention, undone by the caller: reconstruct.py:199-201), triangles (F,3) int64 — device buffers of the sizes
// pass 1 returned.
extern "C" int s3d_mc_dev_emit(const void* grid, int is_f64, int nx, int ny, int nz, int pad, double pad_value,
                               double iso, void* workspace, size_t workspace_bytes, double* vertices,
                               long long* triangles, void* stream) {
    McDev d;
    TRY_RET(mc_make(d, grid, is_f64, nx, ny, nz, pad, pad_value, iso, workspace, workspace_bytes));
    if (d.n_cells == 0) return 0;
    S3D_CHECK_ARG(vertices && triangles, "mc_dev_emit: null output");
    hipLaunchKernelGGL(mc_emit_kernel, dim3((unsigned)d.n_blocks), dim3(MC_BLOCK), 0, (hipStream_t)stream, d, vertices,
                       triangles);
    S3D_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// Dataset staging: the tensor contract of Slice3DDataset.__getitem__
// (reg_slices/src/datasets.py:89-179) from PRE-PACKED uint8 shards (slice3d_amd/shards.py), on the device.
// PNG decode, alpha compositing and the PIL resize were done once at pack time with the reference's own operations;
// what is left per sample is T.ToTensor() + T.Normalize(.5,.5) (datasets.py:31-34) and the query subset.
// =============================================================================================
// u8 (n_img, S, S, 3) HWC  ->  out (n_img*3, S, S) CHW float:  ((u8 / 255) - 0.5) / 0.5 in the reference's fp32 steps
__global__ void u8_hwc_to_norm_chw_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, long n_img,
                                          int S) {
    const long px = (long)S * S;
    const long total = n_img * px;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long img = i / px, p = i % px;
        const unsigned char* s = in + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = __fdiv_rn((float)s[c], 255.0f);
            out[(img * 3 + c) * px + p] = __fdiv_rn(t - 0.5f, 0.5f);
        }
    }
}
// imgs: (B, 1 + n_slices, S, S, 3) uint8, image 0 = the input view, 1.. = the slices in the reference's order
// -> img_input (B,3,S,S), img_slices (B,3*n_slices,S,S)
extern "C" int s3d_dataset_images_fwd(const unsigned char* imgs, float* img_input, float* img_slices, int batch,
                                      int n_slices, int size, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    S3D_CHECK_ARG(imgs && img_input && img_slices && batch >= 1 && n_slices >= 1 && size >= 1, "dataset_images: bad argument");
    const size_t per = (size_t)size * size * 3;
    for (int b = 0; b < batch; ++b) {
        const unsigned char* src = imgs + (size_t)b * (1 + n_slices) * per;
        hipLaunchKernelGGL(u8_hwc_to_norm_chw_kernel, dim3(64), dim3(256), 0, st, src, img_input + (size_t)b * per, 1L, size);
        hipLaunchKernelGGL(u8_hwc_to_norm_chw_kernel, dim3(256), dim3(256), 0, st, src + per,
                           img_slices + (size_t)b * n_slices * per, (long)n_slices, size);
    }
    S3D_LAUNCH_CHECK();
    return 0;
}
// pts (N,4) = (x, y, z, sdf) as the reference's float tensors hold them, idx (n) -> qry (n,3), sdf (n), occ (n) = sdf <= 0
__global__ void gather_points_kernel(const float* __restrict__ pts, const int* __restrict__ idx, long n,
                                     float* __restrict__ qry, float* __restrict__ sdf, float* __restrict__ occ) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f32x4 p = ld4(pts + 4L * idx[i]);
    qry[3 * i + 0] = p[0]; qry[3 * i + 1] = p[1]; qry[3 * i + 2] = p[2];
    sdf[i] = p[3];
    if (occ) occ[i] = p[3] <= 0.f ? 1.f : 0.f;
}
extern "C" int s3d_dataset_points_fwd(const float* pts, const int* idx, long n, float* qry, float* sdf, float* occ,
                                      void* stream) {
    S3D_CHECK_ARG(pts && idx && qry && sdf && n >= 0, "dataset_points: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(gather_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts, idx,
                       n, qry, sdf, occ);
    S3D_LAUNCH_CHECK();
    return 0;
}
