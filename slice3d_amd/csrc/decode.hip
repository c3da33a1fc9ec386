// decode.hip — per-query decoder of Slices3DRegModel (gfx950, fp32 MFMA).
//
// Replaces reg_slices/src/models.py:53-84: query rotation/flip, project_coord (models.py:28-36),
// 5x sample_from_planes (models.py:38-46,69-78), fc_p/fc_s (models.py:79-80), the 3-layer post-LN
// nn.TransformerEncoder (d=128, 4 heads, FFN 2048, relu; models.py:18-19,83) and fc_out (models.py:84).
//
// Token tensor layout in HBM:  X[group][token t][16 queries][128]  (group = 16 consecutive queries of
// one batch item; token 0 = point token, token 1+s = slice s), so every 16x128 block is one MFMA row
// tile of ONE token index and the token-0 rows needed by the pruned last layer are whole tiles.
//
// All GEMMs use the swapped form of common.h: a lane (m = l&15, g = l>>4) holds, for activation row m,
// the 4-channel groups {16j + 4g .. +3}.  The same registers serve as B operand of the next GEMM, as
// residual input and as the layout of the LayerNorm reductions (4 lanes per row -> 2 shuffles).
#include "decode.h"
#include "attn_last.h"

#define LN_EPS 1e-5f

// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------
struct Tap4 {          // bilinear footprint of one query in one level (grid_sample, align_corners=True)
    int off[4];        // element offset of the tap pixel (before * C), clamped in-bounds
    float w[4];        // tap weight, 0 for out-of-bounds taps (padding_mode='zeros')
};

__device__ __forceinline__ Tap4 make_taps(float gx, float gy, int W, int H) {
#pragma clang fp contract(off)   // the weights must not depend on which products hipcc decides to fuse in a given caller (round 6: two
                                 // instantiations of the token builder must produce the same bits)
    // ATen grid_sampler_unnormalize (align_corners): ((coord + 1) / 2) * (size - 1)
    const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float xe = x0f + 1.f, ye = y0f + 1.f;
    Tap4 t;
    const float wnw = (xe - ix) * (ye - iy), wne = (ix - x0f) * (ye - iy);
    const float wsw = (xe - ix) * (iy - y0f), wse = (ix - x0f) * (iy - y0f);
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
    const bool vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    t.off[0] = cy0 * W + cx0; t.w[0] = (vx0 && vy0) ? wnw : 0.f;
    t.off[1] = cy0 * W + cx1; t.w[1] = (vx1 && vy0) ? wne : 0.f;
    t.off[2] = cy1 * W + cx0; t.w[2] = (vx0 && vy1) ? wsw : 0.f;
    t.off[3] = cy1 * W + cx1; t.w[3] = (vx1 && vy1) ? wse : 0.f;
    return t;
}

// project_coord (models.py:28-36): [x y z 1] @ T(4x3); uv = XY / Z; 2(uv - .5); clamp [-1,1]
__device__ __forceinline__ void project(const float* T, float x, float y, float z, float& gx, float& gy) {
    const float X = x * T[0] + y * T[3] + z * T[6] + T[9];
    const float Y = x * T[1] + y * T[4] + z * T[7] + T[10];
    const float Z = x * T[2] + y * T[5] + z * T[8] + T[11];
    gx = fminf(fmaxf(2.f * (X / Z - 0.5f), -1.f), 1.f);
    gy = fminf(fmaxf(2.f * (Y / Z - 0.5f), -1.f), 1.f);
}

// torch.linspace(start,end,steps)[i] as ATen computes it (symmetric halves)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    const float step = (end - start) / (float)(steps - 1);
    return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

// LayerNorm over the 128 channels of a row held as y[8] (f32x4) by the 4 lanes sharing l&15
__device__ __forceinline__ void layer_norm_row(f32x4 (&y)[8], const float* gamma, const float* beta, int g) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (y[j][0] + y[j][1]) + (y[j][2] + y[j][3]);
    const float mean = quad_sum(s) * (1.f / 128.f);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = y[j][i] - mean;
            v += d * d;
        }
    const float rstd = 1.f / sqrtf(quad_sum(v) * (1.f / 128.f) + LN_EPS);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f32x4 ga = ld4(gamma + 16 * j + 4 * g), be = ld4(beta + 16 * j + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[j][i] = (y[j][i] - mean) * rstd * ga[i] + be[i];
    }
}

// ---------------------------------------------------------------------------------------------
// K1: query prologue + pyramid sampling + fc_p/fc_s  ->  X[group][T][16][128]
//     levels 0-2: fc_s already folded into 128-channel maps -> pure weighted gather-accumulate
//     levels 3-4: 96 raw channels sampled straight into MFMA B fragments, times Ws34 (LDS) on MFMA
// ---------------------------------------------------------------------------------------------
// F16 (split-precision modes): the K = 96 product with Ws34 runs on the f16x3 MFMA like every other GEMM of the path.  Cycle
// stamps (round 4, tools/patches/sample_tokens_stamps.patch) showed the fp32 form — 192 dependent-chain v_mfma_f32_16x16x4
// of 8 passes per (16 queries, slice) task, the matrix pipe's fp32 rate — taking 7 000 of a task's 20 000 cycles; 72 MFMAs
// of the 32-deep f16 instruction take 1 200.  The fine levels are then read so that a lane owns 8 consecutive channels of
// each 32-channel block (the k-slots 8g + t of the f16 fragment image, pack_frag f16 form).
//
// Shared footprints (round 6).  Until round 5 every lane gathered the four 512-byte tap rows of its own query in each of the
// three folded levels: 96 loads of 16 B per lane and task, 96 KB through the CU's L1 per (16 queries, slice) — 36 GB per
// step, the kernel's bound (one wave per SIMD, 448 registers of staging, the load path 38 % busy).  But the 16 queries of a
// group are neighbours in the image (s3d_query_sort: Morton order of the projected pixel), so in a folded level (16^2 ..
// 64^2 pixels at S = 256) their 64 taps fall on a handful of pixels.  When the group's floor cells span at most 3 x 3 cells
// (taps inside a 4 x 4 pixel window: nearly always in sorted order), the level is evaluated as ONE small GEMM on the fp32
// MFMA:   out^T [128 ch][16 q] += R^T [128 ch][16 window pixels] * Wt^T [16 pixels][16 q],
// A = the window's rows, gathered ONCE per group (lane (ch, k) loads pixel 4 s + k, channel 16 j + ch: 32 dword loads, 8 KB
// per level instead of 32 KB), B = the sparse bilinear weights (four non-zeros per query, computed per lane from its own
// footprint).  v_mfma_f32_16x16x4_f32 is bitwise a k-ordered fmaf chain and exact zeros in the chain change nothing
// (tools/unit/t_mfma_f32_chain.hip, run on the GPU), and the window's pixel order (row-major) visits a query's taps in the
// order NW, NE, SW, SE of the per-lane form — so both forms give the SAME BITS, a group that does not fit (unsorted queries,
// a jump of the Morton curve) simply takes the per-lane form for that level, and a query's result does not depend on its
// group mates (tests/test_gpu_parity.py::test_shared_footprint_sampler_is_bit_identical; s3d_decode_set_shared_footprint(0)
// forces the per-lane form).  With the staging arrays gone the kernel fits 256 registers: two workgroups per CU.
struct Foot {            // bilinear footprint of one query in one level, factorised (grid_sample, align_corners=True)
    int x0, y0;          // floor cell; >= 0 because the grid is clamped to [-1, 1]
    float wx0, wx1, wy0, wy1;  // column / row factors of the tap weights (make_taps: w = wx * wy), 0 for out-of-bounds columns / rows
                               // (scalars, not arrays: hipcc turns `g == dx ? wx[0] : g == dx + 1 ? wx[1] : 0` into an indexed scratch read)
};
__device__ __forceinline__ Foot make_foot(float gx, float gy, int W, int H) {
#pragma clang fp contract(off)   // (see make_taps)
    const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    Foot f;
    f.x0 = (int)x0f; f.y0 = (int)y0f;
    const int x1 = f.x0 + 1, y1 = f.y0 + 1;
    const float xe = x0f + 1.f, ye = y0f + 1.f;
    f.wx0 = (f.x0 >= 0 && f.x0 < W) ? xe - ix : 0.f;
    f.wx1 = (x1 >= 0 && x1 < W) ? ix - x0f : 0.f;
    f.wy0 = (f.y0 >= 0 && f.y0 < H) ? ye - iy : 0.f;
    f.wy1 = (y1 >= 0 && y1 < H) ? iy - y0f : 0.f;
    return f;
}
// min / max over the 16 lanes of a DPP row (the 16 queries of the group; every row of the wave holds the same 16)
#define S3D_ROW16_REDUCE(v, OP)                                                                         \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false));  /* quad_perm [1,0,3,2] */     \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));  /* quad_perm [2,3,0,1] */     \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false)); /* row_half_mirror */         \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false)); /* row_mirror */
__device__ __forceinline__ int row16_min(int v) { S3D_ROW16_REDUCE(v, min) return v; }
__device__ __forceinline__ int row16_max(int v) { S3D_ROW16_REDUCE(v, max) return v; }

// a * w + c per element with ONE rounding, spelled out: the two forms of a task must not depend on hipcc's contraction choices
__device__ __forceinline__ f32x4 fma4(const f32x4 a, float w, const f32x4 c) {
    return __builtin_elementwise_fma(a, f32x4{w, w, w, w}, c);
}
// One slice-token task: the 16 queries of a group in one slice image -> acc (swapped-form accumulators: lane (m, g) holds
// channels 16 j + 4 g .. + 3 of query m), starting from the fc_s bias.  WINDOW: the three folded levels through the group's
// shared 4 x 4 pixel windows on the fp32 MFMA; else the per-lane form (same bits).  Two separate instantiations instead of a
// per-level choice inside one body: with conditionally issued loads hipcc builds phi webs over the staging arrays and
// spills 200 - 300 registers.
template <bool F16, bool WINDOW>
__device__ __forceinline__ void slice_token_task(const SampleArgs& a, const float* s_ws34, f32x4 (&acc)[8], long img, int S, float gx,
                                                 float gy, const int (&bx)[3], const int (&by)[3], int tz, float* raw_row, int lane) {
    // (every fused multiply-add below is an explicit fma4 / MFMA: nothing is left to hipcc's contraction choices)
    const int m = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = ld4(a.fcs_b + 16 * j + 4 * g + tz);
    // levels 3,4 (raw 64 + 32 channels): per lane, requested under the folded levels' work.  F16: lane (m, g) owns channels
    // 8g .. 8g+7 of every 32-channel block (two 16-byte loads); fp32 form: channels 4g .. 4g+3 of every 16-channel block
    const Tap4 tpf = make_taps(gx, gy, S >> 1, S >> 1), tpq = make_taps(gx, gy, S, S);
    f32x4 vf[2][4], v4[4][2], braw[6];
    const char* bf = reinterpret_cast<const char*>(a.fine[0] + img * (long)(S >> 1) * (S >> 1) * 64);
    const char* bq = reinterpret_cast<const char*>(a.fine[1] + img * (long)S * S * 32);
    // staged so that at most 64 registers of raw taps are in flight: level 4 first, then level 3 in two tap pairs
    auto issue_level4 = [&]() {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                v4[kk][u] = *reinterpret_cast<const f32x4*>(bq + ((unsigned)(tpq.off[kk] * 32 + (F16 ? 8 : 4) * g) * 4u + 4u * (F16 ? 4 * u : 16 * u)));
    };
    auto issue_level3 = [&](int kp) {   // taps 2 kp, 2 kp + 1
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                vf[k2][u] = *reinterpret_cast<const f32x4*>(bf + ((unsigned)(tpf.off[2 * kp + k2] * 64 + (F16 ? 8 : 4) * g) * 4u + 4u * (F16 ? 32 * (u >> 1) + 4 * (u & 1) : 16 * u)));
    };
    auto blend_level4 = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            braw[4 + u] = fma4(v4[3][u], tpq.w[3], fma4(v4[2][u], tpq.w[2], fma4(v4[1][u], tpq.w[1], v4[0][u] * tpq.w[0])));
    };
    auto blend_level3 = [&](int kp) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (kp == 0) braw[u] = fma4(vf[1][u], tpf.w[1], vf[0][u] * tpf.w[0]);
            else braw[u] = fma4(vf[1][u], tpf.w[3], fma4(vf[0][u], tpf.w[2], braw[u]));
        }
    };
    if (WINDOW) {
        // the sparse weight matrix first (B operands: query m's weight of window pixel 4 s + g = wx(g - dx) * wy(s - dy)), so that
        // the footprints are dead before the loads go out
        float wt[3][4];
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const Foot f = make_foot(gx, gy, S >> (4 - l), S >> (4 - l));
            const int dx = f.x0 - bx[l], dy = f.y0 - by[l];
            const float wxq = g == dx ? f.wx0 : g == dx + 1 ? f.wx1 : 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) wt[l][s] = wxq * (s == dy ? f.wy0 : s == dy + 1 ? f.wy1 : 0.f);
        }
        // window rows as MFMA A operands: lane (ch = m, k = g) loads pixel (by + s, bx + g), channel 16 j + ch
        float wa[2][4][8];   // two windows in flight: level l uses buffer l & 1
        auto issue_window = [&](int l) {
            const int W = S >> (4 - l);
            const char* pl = reinterpret_cast<const char*>(a.proj[l] + img * (long)W * W * 128);   // scalar base + 32-bit lane offsets
            const int cx = min(bx[l] + g, W - 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int cy = min(by[l] + s, W - 1);
                const unsigned off = (unsigned)((cy * W + cx) * 128 + m) * 4u;
#pragma unroll
                for (int j = 0; j < 8; ++j) wa[l & 1][s][j] = *reinterpret_cast<const float*>(pl + (off + 64u * j));
            }
        };
        auto consume_window = [&](int l) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[l & 1][s][j], wt[l][s], acc[j], 0, 0, 0);
        };
        // (Requesting all three windows and both fine levels up front — one exposed round trip instead of three — measured SLOWER:
        // 2.11 against 1.70 ms; the 96 + 64 staging registers push the task into spills whose reloads queue behind the loads.)
        issue_window(0);
        issue_window(1);
        issue_level4();
        __builtin_amdgcn_sched_barrier(0);
        consume_window(0);
        __builtin_amdgcn_sched_barrier(0);
        issue_window(2);
        issue_level3(0);
        __builtin_amdgcn_sched_barrier(0);
        blend_level4();
        consume_window(1);
        __builtin_amdgcn_sched_barrier(0);
        blend_level3(0);
        __builtin_amdgcn_sched_barrier(0);
        issue_level3(1);
        __builtin_amdgcn_sched_barrier(0);
        consume_window(2);
        __builtin_amdgcn_sched_barrier(0);
        blend_level3(1);
    } else {
        // per-lane form: the lane's own four tap rows of each level, two taps (16 loads) per round (the taps in the order
        // NW, NE, SW, SE: the order the window form's pixel list visits them in)
        const Foot ft[3] = {make_foot(gx, gy, S >> 4, S >> 4), make_foot(gx, gy, S >> 3, S >> 3), make_foot(gx, gy, S >> 2, S >> 2)};
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const int W = S >> (4 - l);
            const char* base = reinterpret_cast<const char*>(a.proj[l] + img * (long)W * W * 128);
            const int cx0 = min(max(ft[l].x0, 0), W - 1), cx1 = min(max(ft[l].x0 + 1, 0), W - 1);
            const int cy0 = min(max(ft[l].y0, 0), W - 1), cy1 = min(max(ft[l].y0 + 1, 0), W - 1);
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const int cy = kp ? cy1 : cy0;
                const unsigned o0 = (unsigned)((cy * W + cx0) * 128 + 4 * g) * 4u, o1 = (unsigned)((cy * W + cx1) * 128 + 4 * g) * 4u;
                f32x4 v0[8], v1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v0[j] = *reinterpret_cast<const f32x4*>(base + (o0 + 64u * j));
                    v1[j] = *reinterpret_cast<const f32x4*>(base + (o1 + 64u * j));
                }
                if (l == 0 && kp == 0) issue_level4();
                if (l == 1 && kp == 0) { blend_level4(); issue_level3(0); }
                if (l == 2 && kp == 0) { blend_level3(0); __builtin_amdgcn_sched_barrier(0); issue_level3(1); }
                __builtin_amdgcn_sched_barrier(0);   // all requests out before the first use
                const float wyk = kp ? ft[l].wy1 : ft[l].wy0;
                const float w0 = ft[l].wx0 * wyk, w1 = ft[l].wx1 * wyk;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fma4(v0[j], w0, acc[j]);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fma4(v1[j], w1, acc[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        blend_level3(1);
    }
    // raw samples of levels 3, 4 = the B operand of the K = 96 product with Ws34
    if (F16) {
        const _Float16* sw = reinterpret_cast<const _Float16*>(s_ws34) + tz;
        s3d_half8 bh[3], bl[3];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {   // k-slot 8g + t of step kk <-> raw channel 32 kk + 8g + t
            const float x8[8] = {braw[2 * kk][0], braw[2 * kk][1], braw[2 * kk][2], braw[2 * kk][3],
                                 braw[2 * kk + 1][0], braw[2 * kk + 1][1], braw[2 * kk + 1][2], braw[2 * kk + 1][3]};
            s3d_split8(x8, bh[kk], bl[kk]);
        }
        // the 24 fragment pairs are read one step ahead of their MFMAs (read in the step they are used, every step waited a full LDS
        // round trip in front of its three MFMAs: 24 x ~100 cycles per task)
        s3d_half8 fh[2], fl[2];
        fh[0] = *reinterpret_cast<const s3d_half8*>(sw + lane * 8);
        fl[0] = *reinterpret_cast<const s3d_half8*>(sw + 512 + lane * 8);
#pragma unroll
        for (int st = 0; st < 24; ++st) {
            const int kk = st >> 3, j = st & 7;
            if (st + 1 < 24) {
                const int kn = (st + 1) >> 3, jn = (st + 1) & 7;
                fh[(st + 1) & 1] = *reinterpret_cast<const s3d_half8*>(sw + (jn * 3 + kn) * 1024 + lane * 8);
                fl[(st + 1) & 1] = *reinterpret_cast<const s3d_half8*>(sw + (jn * 3 + kn) * 1024 + 512 + lane * 8);
            }
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[st & 1], bl[kk], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[st & 1], bh[kk], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[st & 1], bh[kk], acc[j], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = mfma4(ld4(s_ws34 + ((j * 6 + u) * 64 + lane) * 4 + tz), braw[u], acc[j]);
    }
    if (raw_row) {   // rows of 96 raw channels in channel order (braw[u]: see the loads above for the lane's channels)
        float* ro = raw_row + (F16 ? 8 : 4) * g;
#pragma unroll
        for (int u = 0; u < 6; ++u) st4(ro + (F16 ? 32 * (u >> 1) + 4 * (u & 1) : 16 * u), braw[u]);
    }
}

// Loop order (round 6, second step).  A wave owns the tokens t = wave, wave + 4, ...  Walking group by group and, inside a group,
// token by token sends every task of a wave to a DIFFERENT slice image: the windows / taps of consecutive groups (neighbours in the
// image) are re-used a whole group later, after the CU's other tasks have pushed 12 images' worth of rows through L1 and the XCD's
// L2 (64 workgroups x 12 slices x ~48 KB = 37 MB against 4 MB) — rocprofv3 counted 3.3 M KiB fetched per launch for 0.74 GB of
// pyramid.  Now a workgroup takes its chunk in blocks of ST_BLK groups: phase A computes the block's query prologue ONCE (rotation /
// flip, project_coord, the three folded levels' window origins) into LDS, phase B lets every wave walk the block token-major — all
// groups of one slice image, then the next image — so a task's rows are the rows its predecessor just touched.
#define ST_BLK 64   // groups per block: 1 024 queries x 20 B + 64 x 32 B of LDS
template <bool F16>
__global__ __launch_bounds__(256, 2) void sample_tokens_kernel(const SampleArgs a) {
    __shared__ __attribute__((aligned(16))) float s_ws34[8 * 6 * 256];  // 48 KiB: fp32 [8][6] fragments or f16 hi|lo [8][3]
    // fc_p (128 x 3) transposed + its bias: [x | y | z | b][128].  Read from global per use it was 128 stride-3 scalar loads
    // per lane and made the point token the longest task of a group (34 000 cycles against 14 000 for a slice token): wave 0,
    // which owns it, set the kernel's time
    __shared__ __attribute__((aligned(16))) float s_fcp[4 * 128];
    __shared__ float s_q[5][ST_BLK * S3D_GROUP];   // x | y | z (rotated / flipped) | gx | gy of the block's queries
    __shared__ int s_win[ST_BLK][8];               // bx[3] | by[3] | window form? of the block's groups
    {
        const float* wsrc = F16 ? a.ws34_16 : a.ws34;
        for (int i = threadIdx.x; i < 8 * 6 * 64; i += 256) st4(s_ws34 + 4 * i, ld4(wsrc + 4 * i));
        for (int i = threadIdx.x; i < 512; i += 256) {
            const int c = i & 127, kx = i >> 7;
            s_fcp[i] = kx < 3 ? a.fcp_w[c * 3 + kx] : a.fcp_b[c];
        }
    }

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the image bases are SGPR pairs)
    const int m = lane & 15, g = lane >> 4;
    const int T = a.n_slices + 1;
    const int S = a.size;

    // each block walks a CONTIGUOUS chunk of groups (queries may be locality-sorted), and the blocks that the
    // dispatcher places on one XCD (b % 8) get adjacent chunks, so neighbouring taps meet in that XCD's L2
    const long nb = gridDim.x;
    const long bb = (nb % 8 == 0) ? (long)(blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : (long)blockIdx.x;
    const long chunk = (a.g_count + nb - 1) / nb;
    const long g_lo = bb * chunk, g_hi = g_lo + chunk < a.g_count ? g_lo + chunk : a.g_count;
#pragma unroll 1
    for (long g0 = g_lo; g0 < g_hi; g0 += ST_BLK) {
        const int ng = (int)(g_hi - g0 < ST_BLK ? g_hi - g0 : ST_BLK);
        __syncthreads();   // the previous block's readers are done (first pass: the weight fill above is published)
        // ---- phase A.1: one thread per query of the block ----
        for (int qi = threadIdx.x; qi < ng * S3D_GROUP; qi += 256) {
            const long grp = a.g_begin + g0 + (qi >> 4);
            const int b = (int)(grp / a.groups_per_batch);
            long q = (grp % a.groups_per_batch) * S3D_GROUP + (qi & 15);
            if (q >= a.n_qry) q = a.n_qry - 1;  // padded rows recompute the last query; never stored to sdf
            if (a.perm) q = a.perm[(long)b * a.n_qry + q];
            float x, y, z;
            if (a.qry) {
                const float* p = a.qry + ((long)b * a.n_qry + q) * 3;
                x = p[0]; y = p[1]; z = p[2];
            } else {  // dense grid: x slowest, z fastest (common.py:145-164)
                const long nn = (long)a.nx * a.nx, ql = q + a.q_offset;
                const int ixg = (int)(ql / nn), iyg = (int)((ql / a.nx) % a.nx), izg = (int)(ql % a.nx);
                x = a.box * linspace_at(-0.5f, 0.5f, a.nx, ixg);
                y = a.box * linspace_at(-0.5f, 0.5f, a.nx, iyg);
                z = a.box * linspace_at(-0.5f, 0.5f, a.nx, izg);
            }
            if (a.flip_yz) {  // mode='test' (models.py:53-56)
                y = -y; z = -z;
            } else if (a.rot) {  // qry @ obj_rot_mat (models.py:58-60)
                const float* R = a.rot + b * 9;
                const float rx = x * R[0] + y * R[3] + z * R[6];
                const float ry = x * R[1] + y * R[4] + z * R[7];
                const float rz = x * R[2] + y * R[5] + z * R[8];
                x = rx; y = ry; z = rz;
            }
            float gx, gy;
            project(a.trans + b * 12, x, y, z, gx, gy);
            s_q[0][qi] = x; s_q[1][qi] = y; s_q[2][qi] = z; s_q[3][qi] = gx; s_q[4][qi] = gy;
        }
        __syncthreads();
        // ---- phase A.2: the groups' pixel windows in the three folded levels (the same for every slice of the object); 16 lanes per group ----
        for (int gl = threadIdx.x >> 4; gl < ng; gl += 16) {
            const float gx = s_q[3][gl * S3D_GROUP + m], gy = s_q[4][gl * S3D_GROUP + m];
            int ok = !a.lane_footprints;
            int org[6];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const int W = S >> (4 - l);
                const Foot f = make_foot(gx, gy, W, W);
                const int mnx = row16_min(f.x0), mny = row16_min(f.y0);
                const int ex = row16_max(f.x0) - mnx, ey = row16_max(f.y0) - mny;
                org[l] = mnx; org[3 + l] = mny;
                ok = ok && ex <= 2 && ey <= 2 && mnx >= 0 && mny >= 0;
            }
            if (m == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) s_win[gl][i] = org[i];
                s_win[gl][6] = ok;
            }
        }
        __syncthreads();
        // ---- phase B: token-major walk of the block ----
        // full 128-byte lines per query row (s3d_full_line_pair, common.h): tiles 2J, 2J + 1 of query m are exchanged
        // with lane m ^ 8; one instruction then writes queries 0-7, the next queries 8-15
        auto store_rows = [&](const f32x4 (&acc)[8], long gi, int t) {
            float* o = a.X + ((gi * T + t) * S3D_GROUP + (m & 7)) * 128 + 16 * (m >> 3) + 4 * g;
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                f32x4 va, vb;
                s3d_full_line_pair(acc[2 * J], acc[2 * J + 1], m, va, vb);
                st4(o + 32 * J, va);
                st4(o + 8 * 128 + 32 * J, vb);
            }
        };
#pragma unroll 1
        for (int t = wave; t < T; t += 4) {
#pragma unroll 1
            for (int gl = 0; gl < ng; ++gl) {
                const long gi = g0 + gl;
                f32x4 acc[8];
                // Per task: an opaque zero added to the index of every loop-invariant operand read (fc_p / Ws34 in LDS, the fc_s bias)
                // and to the LDS index of the prologue values.  Left visible, the optimiser hoists those reads and everything
                // derived from the footprints out of the loops and keeps > 150 registers of constants live; the window loads' lane
                // offsets are formed outside the token loop as 64-bit values, spilled, and every reload's s_waitcnt vmcnt(0)
                // serialises the loads behind it.
                int tz = 0;
                asm volatile("" : "+v"(tz));
                const int qi = gl * S3D_GROUP + m + tz;
                if (t == 0) {  // fc_p (models.py:79)
                    const float x = s_q[0][qi], y = s_q[1][qi], z = s_q[2][qi];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int c = 16 * j + 4 * g + tz;
                        const f32x4 wx = ld4(s_fcp + c), wy = ld4(s_fcp + 128 + c), wz = ld4(s_fcp + 256 + c), wb = ld4(s_fcp + 384 + c);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[j][i] = wx[i] * x + wy[i] * y + wz[i] * z + wb[i];   // same expression order as before
                    }
                    if (a.raw_out) {
                        float* ro = a.raw_out + ((gi * T) * S3D_GROUP + m) * 96 + 4 * g;
#pragma unroll
                        for (int u = 0; u < 6; ++u) st4(ro + 16 * u, zero4());
                    }
                } else {
                    const float tgx = s_q[3][qi], tgy = s_q[4][qi];
                    int tbx[3], tby[3];
#pragma unroll
                    for (int l = 0; l < 3; ++l) {
                        tbx[l] = __builtin_amdgcn_readfirstlane(s_win[gl][l]);
                        tby[l] = __builtin_amdgcn_readfirstlane(s_win[gl][3 + l]);
                    }
                    const bool window = __builtin_amdgcn_readfirstlane(s_win[gl][6]) != 0;
                    asm volatile("" : "+s"(tbx[0]), "+s"(tbx[1]), "+s"(tbx[2]), "+s"(tby[0]), "+s"(tby[1]), "+s"(tby[2]));
                    const int b = (int)((a.g_begin + gi) / a.groups_per_batch);
                    const long img = (long)b * a.n_slices + (t - 1);
                    float* raw_row = a.raw_out ? a.raw_out + ((gi * T + t) * S3D_GROUP + m) * 96 : nullptr;
                    if (window)
                        slice_token_task<F16, true>(a, s_ws34, acc, img, S, tgx, tgy, tbx, tby, tz, raw_row, lane);
                    else
                        slice_token_task<F16, false>(a, s_ws34, acc, img, S, tgx, tgy, tbx, tby, tz, raw_row, lane);
                }
                store_rows(acc, gi, t);
            }
        }
    }
}

static int sampler_cu_count() { return s3d_cu_count(); }   // one workgroup per CU of the CURRENT device
int launch_sample_tokens(const SampleArgs& a, hipStream_t stream) {
    S3D_CHECK_ARG(a.size % 16 == 0 && a.size >= 16, "sample: size %d", a.size);
    S3D_CHECK_ARG(a.n_slices >= 1 && a.n_slices + 1 <= S3D_N_TOKENS_MAX, "sample: n_slices %d", a.n_slices);
    // two persistent workgroups per CU (256 registers since round 6; with 2048 workgroups of 3 - 7 groups each the 48 KiB
    // weight fill, the cold tap rows and the last, partly filled round cost 22 %: 0.80 -> 0.62 ms per 100 k queries, round 4)
    const long cap = 2L * sampler_cu_count();
    long blocks = a.g_count < cap ? a.g_count : cap;
    if (blocks <= 0) return 0;
    if (blocks >= 8) blocks -= blocks % 8;
    if (a.ws34_16)
        hipLaunchKernelGGL(sample_tokens_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(sample_tokens_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K2: self-attention block of one encoder layer:  X <- LN1(X + out_proj(MHA(X)))
//     one workgroup (4 waves) per group of 16 queries; wave w owns token tiles t = w, w+4, w+8, w+12.
//     per head: QKV_h on MFMA -> LDS -> per-(query,token) softmax attention on VALU -> LDS ->
//     out_proj partial sums on MFMA (accumulated over heads in registers).
// ---------------------------------------------------------------------------------------------
#define QKV_LD 100   // floats per row of the per-head QKV buffer (96 + pad: conflict-free b128 rows)
#define OH_LD 36     // floats per row of the per-head attention output buffer (32 + pad)
#define ATT_MAXT 4   // token tiles per wave (T <= 16)

template <bool LAST>
__global__ __launch_bounds__(256) void attn_layer_kernel(float* X, float* x0_out, long groups, int T,
                                                         const LayerPtrs w) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_qkv = smem;                                   // [T*16][QKV_LD]
    float* s_oh = smem + S3D_N_TOKENS_MAX * 16 * QKV_LD;   // [T*16][OH_LD]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const float scale = 0.17677669529663687f;  // 1/sqrt(32)

    for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        float* Xg = X + grp * T * S3D_GROUP * 128;
        // B fragments of the owned token tiles
        f32x4 xb[ATT_MAXT][8];
#pragma unroll
        for (int ti = 0; ti < ATT_MAXT; ++ti) {
            const int t = wave + 4 * ti;
            if (t < T) {
                const float* p = Xg + (t * S3D_GROUP + m) * 128 + 4 * g;
#pragma unroll
                for (int u = 0; u < 8; ++u) xb[ti][u] = ld4(p + 16 * u);
            }
        }
        f32x4 acc_o[LAST ? 1 : ATT_MAXT][8];
#pragma unroll
        for (int ti = 0; ti < (LAST ? 1 : ATT_MAXT); ++ti)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc_o[ti][j] = zero4();

#pragma unroll 1
        for (int h = 0; h < 4; ++h) {
            // ---- QKV_h = X Wqkv_h^T + b  (6 column tiles of 16: Q0 Q1 K0 K1 V0 V1) ----
#pragma unroll
            for (int ti = 0; ti < ATT_MAXT; ++ti) {
                const int t = wave + 4 * ti;
                if (t >= T) continue;
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) {
                    if (LAST && jj < 2 && t != 0) continue;  // Q only for token 0 in the last layer
                    const int jg = (jj >> 1) * 8 + h * 2 + (jj & 1);
                    f32x4 c = zero4();
#pragma unroll
                    for (int u = 0; u < 8; ++u) c = mfma4(ld4(frag_ptr(w.inw, jg, u, 8, lane)), xb[ti][u], c);
                    c += ld4(w.inb + 16 * jg + 4 * g);
                    st4(s_qkv + (t * S3D_GROUP + m) * QKV_LD + 16 * jj + 4 * g, c);
                }
            }
            __syncthreads();
            // ---- softmax(q k^T / sqrt(32)) v  per (query ql, token tq): one thread each ----
            {
                const int ql = threadIdx.x & 15, tq = threadIdx.x >> 4;
                if (tq < (LAST ? 1 : T)) {
                    const float* qrow = s_qkv + (tq * S3D_GROUP + ql) * QKV_LD;
                    f32x4 qv[8];
#pragma unroll
                    for (int d = 0; d < 8; ++d) qv[d] = ld4(qrow + 4 * d);
                    float sc[S3D_N_TOKENS_MAX];
                    float mx = -1e30f;
#pragma unroll
                    for (int tk = 0; tk < S3D_N_TOKENS_MAX; ++tk) {
                        if (tk < T) {
                            const float* krow = s_qkv + (tk * S3D_GROUP + ql) * QKV_LD + 32;
                            float s = 0.f;
#pragma unroll
                            for (int d = 0; d < 8; ++d) {
                                const f32x4 kv = ld4(krow + 4 * d);
                                s += qv[d][0] * kv[0] + qv[d][1] * kv[1] + qv[d][2] * kv[2] + qv[d][3] * kv[3];
                            }
                            sc[tk] = s * scale;
                            mx = fmaxf(mx, sc[tk]);
                        }
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
                            const float* vrow = s_qkv + (tk * S3D_GROUP + ql) * QKV_LD + 64;
                            const float p = sc[tk] * inv;
#pragma unroll
                            for (int d = 0; d < 8; ++d) ov[d] += ld4(vrow + 4 * d) * p;
                        }
                    float* orow = s_oh + (tq * S3D_GROUP + ql) * OH_LD;
#pragma unroll
                    for (int d = 0; d < 8; ++d) st4(orow + 4 * d, ov[d]);
                }
            }
            __syncthreads();
            // ---- out_proj partial: acc_o += Wo[:, 32h:32h+32] * O_h^T ----
#pragma unroll
            for (int ti = 0; ti < (LAST ? 1 : ATT_MAXT); ++ti) {
                const int t = wave + 4 * ti;
                if (t >= (LAST ? 1 : T)) continue;
#pragma unroll
                for (int u2 = 0; u2 < 2; ++u2) {
                    const f32x4 ob = ld4(s_oh + (t * S3D_GROUP + m) * OH_LD + 16 * u2 + 4 * g);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        acc_o[ti][j] = mfma4(ld4(frag_ptr(w.outw, j, h * 2 + u2, 8, lane)), ob, acc_o[ti][j]);
                }
            }
            // the next head's s_qkv writes are fenced from this head's attention reads by the barrier
            // above; its s_oh writes are fenced from these reads by the next head's first barrier.
        }
        // ---- residual + LayerNorm1, store ----
#pragma unroll
        for (int ti = 0; ti < (LAST ? 1 : ATT_MAXT); ++ti) {
            const int t = wave + 4 * ti;
            if (t >= (LAST ? 1 : T)) continue;
            f32x4 y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = acc_o[ti][j] + ld4(w.outb + 16 * j + 4 * g) + xb[ti][j];
            layer_norm_row(y, w.ln1g, w.ln1b, g);
            float* o = LAST ? x0_out + (grp * S3D_GROUP + m) * 128 + 4 * g
                            : Xg + (t * S3D_GROUP + m) * 128 + 4 * g;
#pragma unroll
            for (int j = 0; j < 8; ++j) st4(o + 16 * j, y[j]);
        }
        __syncthreads();  // LDS reuse by the next group
    }
}

int launch_attn_layer(float* X, float* x0_out, long groups, int T, const LayerPtrs& w, hipStream_t stream) {
    S3D_CHECK_ARG(T >= 2 && T <= S3D_N_TOKENS_MAX, "attn: T %d", T);
    // the last layer (token 0 only) always runs the absorbed form (launch_attn_last_mix); the LAST instantiation of this
    // kernel is no longer built
    S3D_CHECK_ARG(x0_out == nullptr, "attn: the token-0-only form of this kernel was retired (use the absorbed last layer)");
    if (groups <= 0) return 0;
    const size_t lds = (size_t)S3D_N_TOKENS_MAX * 16 * (QKV_LD + OH_LD) * sizeof(float);
    static std::atomic<unsigned long long> attr_done{0};
    TRY_RET(s3d_set_max_lds(attr_done, {(const void*)attn_layer_kernel<false>}, lds));
    const long blocks = groups < 4096 ? groups : 4096;
    hipLaunchKernelGGL(attn_layer_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, stream, X, x0_out, groups, T, w);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K3: feed-forward block:  X <- LN2(X + W2 relu(W1 X + b1) + b2)   [+ fc_out when FINAL]
//     rows are independent: a workgroup takes 256 rows (4 waves x 4 row tiles).  The activation row
//     tile lives in registers as B fragments for the whole kernel; W1/W2 stream through LDS in
//     32-hidden-unit chunks (double buffered); the hidden tile never leaves the register file: the D
//     registers of GEMM1 (relu'd) are the B operand of GEMM2.
// ---------------------------------------------------------------------------------------------
#define FFN_R 4
#define FFN_CHUNK_FLOATS 8192   // 16 KiB of W1 fragments + 16 KiB of W2 fragments

template <bool FINAL>
__global__ __launch_bounds__(256) void ffn_layer_kernel(const float* X, float* Yout, float* Uout, long rows,
                                                        const LayerPtrs w, const float* fco_w, const float* fco_b,
                                                        float* sdf_out, float sign, long groups_per_batch,
                                                        long n_qry, long g_begin, const DropCfg dh, const DropCfg dq,
                                                        const int* perm) {
    __shared__ __attribute__((aligned(16))) float s_w[2][FFN_CHUNK_FLOATS];  // 64 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const long row0 = ((long)blockIdx.x * 4 + wave) * (FFN_R * 16);

    f32x4 xb[FFN_R][8], acc[FFN_R][8];
#pragma unroll
    for (int r = 0; r < FFN_R; ++r) {
        long row = row0 + r * 16 + m;
        if (row >= rows) row = rows - 1;
        const float* p = X + row * 128 + 4 * g;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            xb[r][u] = ld4(p + 16 * u);
            acc[r][u] = zero4();
        }
    }
    // stage chunk 0
    f32x4 pre[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pre[i] = ld4(w.w1 + (i * 256 + threadIdx.x) * 4);
        pre[4 + i] = ld4(w.w2 + (i * 256 + threadIdx.x) * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        st4(&s_w[0][(i * 256 + threadIdx.x) * 4], pre[i]);
        st4(&s_w[0][4096 + (i * 256 + threadIdx.x) * 4], pre[4 + i]);
    }
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < S3D_FFN_NCHUNK; ++c) {
        const float* sw = s_w[c & 1];
        if (c + 1 < S3D_FFN_NCHUNK) {
            const float* g1 = w.w1 + (size_t)(c + 1) * 4096;
            const float* g2 = w.w2 + (size_t)(c + 1) * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pre[i] = ld4(g1 + (i * 256 + threadIdx.x) * 4);
                pre[4 + i] = ld4(g2 + (i * 256 + threadIdx.x) * 4);
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            f32x4 hd[FFN_R];
#pragma unroll
            for (int r = 0; r < FFN_R; ++r) hd[r] = zero4();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 wa = ld4(sw + ((a * 8 + u) * 64 + lane) * 4);
#pragma unroll
                for (int r = 0; r < FFN_R; ++r) hd[r] = mfma4(wa, xb[r][u], hd[r]);
            }
            const f32x4 b1 = ld4(w.b1 + c * S3D_FFN_CHUNK + 16 * a + 4 * g);
#pragma unroll
            for (int r = 0; r < FFN_R; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) hd[r][i] = fmaxf(hd[r][i] + b1[i], 0.f);
            if (dh.p > 0.f) {   // train-mode dropout on the hidden activations (index = row*2048 + unit)
#pragma unroll
                for (int r = 0; r < FFN_R; ++r) {
                    const unsigned long long base =
                        (unsigned long long)(row0 + r * 16 + m) * S3D_FFN + c * S3D_FFN_CHUNK + 16 * a + 4 * g;
                    float mk4[4];
                    s3d_drop4(dh, base, mk4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) hd[r][i] *= mk4[i];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 wb = ld4(sw + 4096 + ((j * 2 + a) * 64 + lane) * 4);
#pragma unroll
                for (int r = 0; r < FFN_R; ++r) acc[r][j] = mfma4(wb, hd[r], acc[r][j]);
            }
        }
        if (c + 1 < S3D_FFN_NCHUNK) {
            float* dw = s_w[(c + 1) & 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                st4(dw + (i * 256 + threadIdx.x) * 4, pre[i]);
                st4(dw + 4096 + (i * 256 + threadIdx.x) * 4, pre[4 + i]);
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int r = 0; r < FFN_R; ++r) {
        const long row = row0 + r * 16 + m;
        f32x4 y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 f = acc[r][j] + ld4(w.b2 + 16 * j + 4 * g);
            if (dq.p > 0.f) {
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] *= s3d_drop(dq, (unsigned long long)row * 128 + 16 * j + 4 * g + i);
            }
            y[j] = f + xb[r][j];
        }
        if (!FINAL && Uout && row < rows) {
            float* uo = Uout + row * 128 + 4 * g;
#pragma unroll
            for (int j = 0; j < 8; ++j) st4(uo + 16 * j, y[j]);
        }
        layer_norm_row(y, w.ln2g, w.ln2b, g);
        if (FINAL) {  // fc_out (models.py:84) on the token-0 row of query (group, m)
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 wo = ld4(fco_w + 16 * j + 4 * g);
                s += y[j][0] * wo[0] + y[j][1] * wo[1] + y[j][2] * wo[2] + y[j][3] * wo[3];
            }
            s = quad_sum(s) + fco_b[0];
            if (g == 0 && row < rows) {
                const long grp = g_begin + row / S3D_GROUP;
                const long b = grp / groups_per_batch;
                const long q = (grp % groups_per_batch) * S3D_GROUP + (row % S3D_GROUP);
                if (q < n_qry) sdf_out[b * n_qry + (perm ? perm[b * n_qry + q] : q)] = sign * s;
            }
        } else if (row < rows) {
            float* o = Yout + row * 128 + 4 * g;
#pragma unroll
            for (int j = 0; j < 8; ++j) st4(o + 16 * j, y[j]);
        }
    }
}

int launch_ffn_layer(float* X, long rows, const LayerPtrs& w, const float* fco_w, const float* fco_b,
                     float* sdf_out, float sign, long groups_per_batch, long n_qry, long g_begin, int prec,
                     const int* perm, hipStream_t stream, bool pre_ln1) {
    if (prec == S3D_PREC_F16X3 || prec == S3D_PREC_F16 || prec == S3D_PREC_BF16)
        return launch_ffn_layer_f16x3(X, rows, w, prec == S3D_PREC_BF16 ? w.wfb16 : w.wf16, fco_w, fco_b, sdf_out, sign,
                                      groups_per_batch, n_qry, g_begin, perm, stream, prec != S3D_PREC_F16X3, pre_ln1,
                                      prec == S3D_PREC_BF16);
    S3D_CHECK_ARG(prec == S3D_PREC_F32 && !pre_ln1, "ffn: precision mode %d not built (LayerNorm prologue: split precision only)", prec);
    if (rows <= 0) return 0;
    const long blocks = (rows + 4 * FFN_R * 16 - 1) / (4 * FFN_R * 16);
    if (sdf_out)
        hipLaunchKernelGGL(ffn_layer_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, X, X, nullptr, rows,
                           w, fco_w, fco_b, sdf_out, sign, groups_per_batch, n_qry, g_begin, make_drop(0, 0.f, 0),
                           make_drop(0, 0.f, 0), perm);
    else
        hipLaunchKernelGGL(ffn_layer_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, X, X, nullptr, rows,
                           w, fco_w, fco_b, sdf_out, sign, groups_per_batch, n_qry, g_begin, make_drop(0, 0.f, 0),
                           make_drop(0, 0.f, 0), perm);
    S3D_LAUNCH_CHECK();
    return 0;
}

int launch_ffn_layer_train(const float* Xin, float* Yout, float* Uout, long rows, const LayerPtrs& w,
                           const DropCfg& drop_hidden, const DropCfg& drop_out, hipStream_t stream) {
    if (rows <= 0) return 0;
    const long blocks = (rows + 4 * FFN_R * 16 - 1) / (4 * FFN_R * 16);
    hipLaunchKernelGGL(ffn_layer_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, Xin, Yout, Uout, rows,
                       w, nullptr, nullptr, nullptr, 1.f, 1L, 1L, 0L, drop_hidden, drop_out, nullptr);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Token-0 attention of the last layer, absorbed form (see decode.h).  16 lanes per query (8 channels each), 4 queries
// per wave, one 256-thread workgroup per group of 16 queries.  The 13 token rows of the query stay in registers
// (104 floats per lane) for both passes (scores, then the probability-weighted sum); per head the 16-lane dot
// products are all-reduced with four DPP steps (quad xor 1, quad xor 2, half mirror, row mirror).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_last_mix_kernel(const float* __restrict__ X, const float* __restrict__ qt,
                                                            float* __restrict__ xbar, long groups, int T) {
    const int mq = threadIdx.x >> 4, c0 = (threadIdx.x & 15) * 8;   // query of the group, first channel of the lane
    for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const float* xg = X + (grp * T * S3D_GROUP + mq) * 128 + c0;
        f32x4 xa[S3D_N_TOKENS_MAX], xb[S3D_N_TOKENS_MAX];
#pragma unroll
        for (int t = 0; t < S3D_N_TOKENS_MAX; ++t) {
            const int tc = t < T ? t : T - 1;
            xa[t] = ld4(xg + (long)tc * S3D_GROUP * 128);
            xb[t] = ld4(xg + (long)tc * S3D_GROUP * 128 + 4);
        }
        const long row = grp * S3D_GROUP + mq;
#pragma unroll 1
        for (int h = 0; h < 4; ++h) {
            const f32x4 qa = ld4(qt + row * 512 + h * 128 + c0), qb = ld4(qt + row * 512 + h * 128 + c0 + 4);
            f32x4 oa, ob;
            al_mix_head(xa, xb, qa, qb, T, oa, ob);   // attn_last.h: shared with the fused kernel (decode_last.hip)
            st4(xbar + row * 512 + h * 128 + c0, oa);
            st4(xbar + row * 512 + h * 128 + c0 + 4, ob);
        }
    }
}
int launch_attn_last_mix(const float* X, const float* qt, float* xbar, long groups, int T, hipStream_t stream) {
    if (groups <= 0) return 0;
    S3D_CHECK_ARG(T >= 1 && T <= S3D_N_TOKENS_MAX, "attn_last_mix: T %d", T);
    const long blocks = groups < 8192 ? groups : 8192;
    hipLaunchKernelGGL(attn_last_mix_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, X, qt, xbar, groups, T);
    S3D_LAUNCH_CHECK();
    return 0;
}

// pack-time products of the absorbed form (double accumulation, rounded once to fp32)
__global__ void absorb_last_kernel(const float* __restrict__ in_w, const float* __restrict__ in_b,
                                   const float* __restrict__ out_w, const float* __restrict__ out_b,
                                   float* __restrict__ out, float* __restrict__ bias_out, int kind) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // 65536 matrix elements
    if (idx >= 512 * 128) return;
    const float* Wq = in_w;
    const float* Wk = in_w + 128 * 128;
    const float* Wv = in_w + 256 * 128;
    if (kind == 0) {   // M[h*128 + c][k] = sum_d Wk[32h+d][c] Wq[32h+d][k]
        const int r = idx >> 7, k = idx & 127, h = r >> 7, c = r & 127;
        double s = 0.0;
        for (int d = 0; d < 32; ++d) s += (double)Wk[(32 * h + d) * 128 + c] * (double)Wq[(32 * h + d) * 128 + k];
        out[idx] = (float)s;
        if (k == 0) {
            double v = 0.0;
            for (int d = 0; d < 32; ++d) v += (double)Wk[(32 * h + d) * 128 + c] * (double)in_b[32 * h + d];
            bias_out[r] = (float)v;
        }
    } else {           // N[n][h*128 + c] = sum_d Wo[n][32h+d] Wv[32h+d][c]
        const int n = idx >> 9, kk = idx & 511, h = kk >> 7, c = kk & 127;
        double s = 0.0;
        for (int d = 0; d < 32; ++d) s += (double)out_w[n * 128 + 32 * h + d] * (double)Wv[(32 * h + d) * 128 + c];
        out[idx] = (float)s;
        if (kk == 0) {
            double v = (double)out_b[n];
            for (int d = 0; d < 128; ++d) v += (double)out_w[n * 128 + d] * (double)in_b[256 + d];
            bias_out[n] = (float)v;
        }
    }
}
int launch_absorb_last(const float* in_w, const float* in_b, const float* out_w, const float* out_b, float* out,
                       float* bias_out, int kind, hipStream_t stream) {
    hipLaunchKernelGGL(absorb_last_kernel, dim3(256), dim3(256), 0, stream, in_w, in_b, out_w, out_b, out, bias_out, kind);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// image-space locality sort of the queries (counting sort, 65536 Morton bins of the projected pixel)
// ---------------------------------------------------------------------------------------------
#define QS_BINS 65536
__device__ __forceinline__ unsigned morton8(unsigned x, unsigned y) {
    auto spread = [](unsigned v) {
        v = (v | (v << 4)) & 0x0F0Fu;
        v = (v | (v << 2)) & 0x3333u;
        v = (v | (v << 1)) & 0x5555u;
        return v;
    };
    return spread(x) | (spread(y) << 1);
}
__global__ void qsort_key_kernel(const float* qry, const float* rot, const float* trans, int flip_yz, int batch,
                                 long n_qry, int* keys, int* hist) {
    const long total = (long)batch * n_qry;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / n_qry);
        if (!trans) {   // already-projected coordinates: qry = (B, Q, 2) grid in [-1, 1]
            const float gx = qry[i * 2], gy = qry[i * 2 + 1];
            const unsigned px = (unsigned)fminf(fmaxf((gx + 1.f) * 127.5f, 0.f), 255.f);
            const unsigned py = (unsigned)fminf(fmaxf((gy + 1.f) * 127.5f, 0.f), 255.f);
            const int key = (int)morton8(px, py);
            keys[i] = key;
            atomicAdd(&hist[(long)b * QS_BINS + key], 1);
            continue;
        }
        float x = qry[i * 3], y = qry[i * 3 + 1], z = qry[i * 3 + 2];
        if (flip_yz) {
            y = -y; z = -z;
        } else if (rot) {
            const float* R = rot + b * 9;
            const float rx = x * R[0] + y * R[3] + z * R[6];
            const float ry = x * R[1] + y * R[4] + z * R[7];
            const float rz = x * R[2] + y * R[5] + z * R[8];
            x = rx; y = ry; z = rz;
        }
        float gx, gy;
        project(trans + b * 12, x, y, z, gx, gy);
        const unsigned px = (unsigned)fminf(fmaxf((gx + 1.f) * 127.5f, 0.f), 255.f);
        const unsigned py = (unsigned)fminf(fmaxf((gy + 1.f) * 127.5f, 0.f), 255.f);
        const int key = (int)morton8(px, py);
        keys[i] = key;
        atomicAdd(&hist[(long)b * QS_BINS + key], 1);
    }
}
// exclusive scan of each batch item's 65536-bin histogram, in place (one 1024-thread block per item)
__global__ __launch_bounds__(1024) void qsort_scan_kernel(int* hist) {
    __shared__ int part[1024];
    int* h = hist + (long)blockIdx.x * QS_BINS;
    const int t = threadIdx.x;
    int local[64];
    int s = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        local[i] = s;
        s += h[t * 64 + i];
    }
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const int base = t ? part[t - 1] : 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) h[t * 64 + i] = base + local[i];
}
__global__ void qsort_scatter_kernel(const int* keys, int* offs, int batch, long n_qry, int* perm) {
    const long total = (long)batch * n_qry;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / n_qry);
        const int pos = atomicAdd(&offs[(long)b * QS_BINS + keys[i]], 1);
        perm[(long)b * n_qry + pos] = (int)(i - (long)b * n_qry);
    }
}

// The atomic scatter leaves the members of one bin in arbitrary order.  Restore ascending query index inside
// every bin so that the permutation is a pure function of the inputs (reproducible dropout streams, row-wise
// comparisons against the oracle): bins of <= 32 members (the common case: ~1.5 per bin) are insertion-sorted by
// one thread; larger bins (clamped border pixels) go to an overflow list and get a workgroup rank sort.
__global__ void qsort_fix_kernel(const int* __restrict__ ends, int batch, long n_qry, int* __restrict__ perm,
                                 int* __restrict__ ovf) {
    const long total = (long)batch * QS_BINS;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / QS_BINS), k = (int)(i % QS_BINS);
        const int beg = k ? ends[i - 1] : 0, n = ends[i] - beg;
        if (n <= 1) continue;
        int* p = perm + (long)b * n_qry + beg;
        if (n <= 32) {
            for (int u = 1; u < n; ++u) {
                const int v = p[u];
                int w = u - 1;
                while (w >= 0 && p[w] > v) {
                    p[w + 1] = p[w];
                    --w;
                }
                p[w + 1] = v;
            }
        } else {
            const int slot = atomicAdd(&ovf[0], 1);
            ovf[1 + slot] = (int)i;
        }
    }
}
__global__ __launch_bounds__(256) void qsort_fix_large_kernel(const int* __restrict__ ends, long n_qry,
                                                              int* __restrict__ perm, int* __restrict__ tmp,
                                                              const int* __restrict__ ovf) {
    const int cnt = ovf[0];
    for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
        const long i = ovf[1 + e];
        const int b = (int)(i / QS_BINS), k = (int)(i % QS_BINS);
        const int beg = k ? ends[i - 1] : 0, n = ends[i] - beg;
        int* p = perm + (long)b * n_qry + beg;
        int* t = tmp + (long)b * n_qry + beg;
        for (int u = threadIdx.x; u < n; u += 256) {
            const int v = p[u];
            int rank = 0;
            for (int w = 0; w < n; ++w) rank += p[w] < v;
            t[rank] = v;
        }
        __syncthreads();
        for (int u = threadIdx.x; u < n; u += 256) p[u] = t[u];
        __syncthreads();
    }
}

// workspace: hist/ends [batch][65536] | keys (later: rank-sort scratch) [batch][n_qry] | overflow list
size_t query_sort_ws_ints(int batch, long n_qry) {
    return (size_t)batch * QS_BINS + (size_t)batch * n_qry + (size_t)batch * (n_qry / 32 + 1) + 1;
}

// perm[b][pos] = original query index; on return ws[b*65536 + k] = END of bin k (start of bin k+1) in perm[b]
int launch_query_sort(const float* qry, const float* rot, const float* trans, int flip_yz, int batch, long n_qry,
                      int* perm, int* ws, hipStream_t stream) {
    int* hist = ws;
    int* keys = ws + (size_t)batch * QS_BINS;
    int* ovf = keys + (size_t)batch * n_qry;
    if (hipMemsetAsync(hist, 0, (size_t)batch * QS_BINS * sizeof(int), stream) != hipSuccess ||
        hipMemsetAsync(ovf, 0, sizeof(int), stream) != hipSuccess) {
        s3d_set_error("query_sort: memset failed");
        return (int)hipErrorUnknown;
    }
    const long total = (long)batch * n_qry;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(qsort_key_kernel, dim3(blocks), dim3(256), 0, stream, qry, rot, trans, flip_yz, batch, n_qry,
                       keys, hist);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(qsort_scan_kernel, dim3(batch), dim3(1024), 0, stream, hist);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(qsort_scatter_kernel, dim3(blocks), dim3(256), 0, stream, keys, hist, batch, n_qry, perm);
    S3D_LAUNCH_CHECK();
    const int fb = batch * QS_BINS / 256 < 4096 ? batch * QS_BINS / 256 : 4096;
    hipLaunchKernelGGL(qsort_fix_kernel, dim3(fb), dim3(256), 0, stream, hist, batch, n_qry, perm, ovf);
    S3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(qsort_fix_large_kernel, dim3(256), dim3(256), 0, stream, hist, n_qry, perm, keys, ovf);
    S3D_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// stand-alone helper ops of the module API
// ---------------------------------------------------------------------------------------------
__global__ void project_coord_kernel(const float* coords, const float* trans, float* out, int batch,
                                     long n_qry) {
    const long total = (long)batch * n_qry;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / n_qry);
        float gx, gy;
        project(trans + b * 12, coords[i * 3], coords[i * 3 + 1], coords[i * 3 + 2], gx, gy);
        out[i * 2] = gx;
        out[i * 2 + 1] = gy;
    }
}

int launch_project_coord(const float* coords, const float* trans, float* out, int batch, long n_qry,
                         hipStream_t stream) {
    const long total = (long)batch * n_qry;
    if (total <= 0) return 0;
    const long blocks = (total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096;
    hipLaunchKernelGGL(project_coord_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, coords, trans, out,
                       batch, n_qry);
    S3D_LAUNCH_CHECK();
    return 0;
}

// sample_from_planes on a channels-last plane: one thread per (point, 4 channels); consecutive lanes
// read consecutive 16 B of a tap's C-vector and write consecutive 16 B of the output row.
__global__ void sample_planes_kernel(const float* __restrict__ plane, const float* __restrict__ grid,
                                     float* __restrict__ out, int n, int h, int w, int c, long mpts) {
    const int c4 = c >> 2;
    const long total = (long)n * mpts * c4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % c4) * 4;
        const long pt = idx / c4;
        const int ni = (int)(pt / mpts);
        const Tap4 tp = make_taps(grid[pt * 2], grid[pt * 2 + 1], w, h);
        const float* base = plane + (long)ni * h * w * c + cc;
        f32x4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = ld4(base + (long)tp.off[k] * c);
        __builtin_amdgcn_sched_barrier(0);   // four taps in flight
        st4(out + pt * c + cc, (t[0] * tp.w[0] + t[1] * tp.w[1]) + (t[2] * tp.w[2] + t[3] * tp.w[3]));
    }
}

// Fused form of the reference's five sample_from_planes calls + torch.cat (models.py:66-73): every (slice
// image, point) gets its 992-channel row [512 | 256 | 128 | 64 | 32] of bilinear samples of the five pyramid
// levels.  Points are visited in image-space locality order (perm), each row is written once, contiguously:
// the kernel is bound by the 3 968-byte row write.  thread = (row, channel quad).
struct SamplePyrArgs {
    const float* level[5];
    const float* grid;   // (B, Q, 2)
    const int* perm;     // (B, Q) or null
    float4* pts;         // (B, Q) scratch: (gx, gy, original index as int bits, 0) in visiting order
    float* out;          // (B*ns, Q, 992)
    int size, n_slices, batch;
    long n_qry;
};
__global__ void sample_pyramid_pts_kernel(const SamplePyrArgs a) {
    const long total = (long)a.batch * a.n_qry;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / a.n_qry;
        const int q = a.perm ? a.perm[i] : (int)(i - b * a.n_qry);
        const float* gp = a.grid + (b * a.n_qry + q) * 2;
        a.pts[i] = make_float4(gp[0], gp[1], __int_as_float(q), 0.f);
    }
}
// A workgroup owns SP_CHUNK consecutive rows = points that follow each other in the image-space locality order of ONE
// image; thread = channel quad (248 of 256 threads: [512 | 256 | 128 | 64 | 32] channels).  Consecutive points share
// their tap pixels on the coarse levels almost always (a level-0 pixel cell spans 16x16 of the sort's 256^2 bins: a few
// hundred points in a row), so a thread keeps its four tap vectors in registers and re-reads them only when the cell of
// its level changes: the gather traffic through L1 / L2 — 4 taps x 3 968 B per row, four times the bytes written, and
// what bound the previous one-row-at-a-time kernel at 3.3 TB/s — drops to a few per cent, and the kernel is left with
// its 3 968-byte non-temporal row stores.
// The footprints (4 tap offsets + 4 weights per level) of the chunk's rows are computed ONCE, cooperatively, into LDS
// (128 rows x 5 levels x 32 B = 20 KiB); in the row loop a thread reads its level's record with two broadcast
// ds_read_b128 — with every lane recomputing make_taps per row the kernel was VALU-bound (60 instructions per row and
// wave), not write-bound.
#define SP_CHUNK 128
__global__ __launch_bounds__(256) void sample_pyramid_kernel(const SamplePyrArgs a) {
    __shared__ __attribute__((aligned(16))) int4 s_off[SP_CHUNK][5];      // off0, off1, off2, cell id
    __shared__ __attribute__((aligned(16))) float4 s_w[SP_CHUNK][5];
    __shared__ int s_q[SP_CHUNK];
    const int cq = threadIdx.x;
    const bool active = cq < 248;
    int l, c0;   // level and first channel inside the level
    if (cq < 128) { l = 0; c0 = 4 * cq; }
    else if (cq < 192) { l = 1; c0 = 4 * (cq - 128); }
    else if (cq < 224) { l = 2; c0 = 4 * (cq - 192); }
    else if (cq < 240) { l = 3; c0 = 4 * (cq - 224); }
    else { l = 4; c0 = 4 * (cq - 240); }
    const int C = 512 >> l, W = (a.size / 16) << l;
    const long chunks_per_img = (a.n_qry + SP_CHUNK - 1) / SP_CHUNK;
    const long n_img = (long)a.batch * a.n_slices;
    for (long ch = blockIdx.x; ch < n_img * chunks_per_img; ch += gridDim.x) {
        const long img = ch / chunks_per_img;
        const long p0 = (ch - img * chunks_per_img) * SP_CHUNK;
        const int nrows = (int)(p0 + SP_CHUNK < a.n_qry ? SP_CHUNK : a.n_qry - p0);
        const float4* pts = a.pts + (img / a.n_slices) * a.n_qry + p0;
        __syncthreads();   // the previous chunk's records are no longer read
        for (int i = threadIdx.x; i < nrows * 5; i += 256) {
            const int row = i / 5, lv = i - row * 5;
            const float4 pt = pts[row];
            const int Wl = (a.size / 16) << lv;
            const Tap4 tp = make_taps(pt.x, pt.y, Wl, Wl);
            const int id = tp.off[0] * 4 + (tp.off[1] != tp.off[0] ? 1 : 0) + (tp.off[2] != tp.off[0] ? 2 : 0);
            s_off[row][lv] = make_int4(tp.off[0], tp.off[1], tp.off[2], id);
            s_w[row][lv] = make_float4(tp.w[0], tp.w[1], tp.w[2], tp.w[3]);
            if (lv == 0) s_q[row] = __float_as_int(pt.z);
        }
        __syncthreads();
        if (!active) continue;
        const float* plane = a.level[l] + img * (long)W * W * C + c0;
        float* out = a.out + img * a.n_qry * 992 + 4 * cq;
        int cell = -1;
        f32x4 t0 = zero4(), t1 = zero4(), t2 = zero4(), t3 = zero4();
        for (int row = 0; row < nrows; ++row) {
            const int4 o = s_off[row][l];
            const float4 w = s_w[row][l];
            if (o.w != cell) {                                      // uniform over the lanes of a level
                cell = o.w;
                t0 = ld4(plane + (long)o.x * C);
                t1 = ld4(plane + (long)o.y * C);
                t2 = ld4(plane + (long)o.z * C);
                t3 = ld4(plane + (long)(o.y + o.z - o.x) * C);
            }
            __builtin_nontemporal_store((t0 * w.x + t1 * w.y) + (t2 * w.z + t3 * w.w),
                                        reinterpret_cast<f32x4*>(out + (long)s_q[row] * 992));
        }
    }
}
int launch_sample_pyramid_points(const float* grid, const int* perm, float* pts, int batch, long n_qry,
                                 hipStream_t stream) {
    SamplePyrArgs a = {};
    a.grid = grid; a.perm = perm; a.pts = reinterpret_cast<float4*>(pts); a.batch = batch; a.n_qry = n_qry;
    const long np = (long)batch * n_qry;
    if (np <= 0) return 0;
    hipLaunchKernelGGL(sample_pyramid_pts_kernel, dim3((unsigned)((np + 255) / 256 < 4096 ? (np + 255) / 256 : 4096)),
                       dim3(256), 0, stream, a);
    S3D_LAUNCH_CHECK();
    return 0;
}
int launch_sample_pyramid(const float* const* level, const float* grid, const int* perm, float* pts, float* out,
                          int batch, int n_slices, int size, long n_qry, hipStream_t stream) {
    SamplePyrArgs a;
    for (int l = 0; l < 5; ++l) a.level[l] = level[l];
    a.grid = grid; a.perm = perm; a.pts = reinterpret_cast<float4*>(pts); a.out = out;
    a.size = size; a.n_slices = n_slices; a.batch = batch; a.n_qry = n_qry;
    const long rows = (long)batch * n_slices * n_qry;
    if (rows <= 0) return 0;
    const long chunks = (long)batch * n_slices * ((n_qry + SP_CHUNK - 1) / SP_CHUNK);
    const long blocks = chunks < 65536 ? chunks : 65536;
    hipLaunchKernelGGL(sample_pyramid_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    S3D_LAUNCH_CHECK();
    return 0;
}

int launch_sample_planes(const float* plane, const float* grid, float* out, int n, int h, int w, int c, long m,
                         hipStream_t stream) {
    S3D_CHECK_ARG(c % 4 == 0 && c > 0, "sample_planes: C %d must be a multiple of 4", c);
    const long total = (long)n * m * (c / 4);
    if (total <= 0) return 0;
    const long blocks = (total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384;
    hipLaunchKernelGGL(sample_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, plane, grid, out, n,
                       h, w, c, m);
    S3D_LAUNCH_CHECK();
    return 0;
}


// =============================================================================================
// Slices3DGTModel front end (model_gt.py:77-96)
// =============================================================================================
__device__ __forceinline__ void gt_query_xyz(const float* qry, const float* rot, int flip_yz, const int* perm,
                                             int nx, float box, int b, long q, long n_qry, float& x, float& y,
                                             float& z) {
    if (perm) q = perm[(long)b * n_qry + q];
    if (qry) {
        const float* p = qry + ((long)b * n_qry + q) * 3;
        x = p[0]; y = p[1]; z = p[2];
    } else {
        const long nn = (long)nx * nx;
        const int ixg = (int)(q / nn), iyg = (int)((q / nx) % nx), izg = (int)(q % nx);
        x = box * linspace_at(-0.5f, 0.5f, nx, ixg);
        y = box * linspace_at(-0.5f, 0.5f, nx, iyg);
        z = box * linspace_at(-0.5f, 0.5f, nx, izg);
    }
    if (flip_yz) {
        y = -y; z = -z;
    } else if (rot) {
        const float* R = rot + b * 9;
        const float rx = x * R[0] + y * R[3] + z * R[6];
        const float ry = x * R[1] + y * R[4] + z * R[7];
        const float rz = x * R[2] + y * R[5] + z * R[8];
        x = rx; y = ry; z = rz;
    }
}

// F16: the K = 64 product with the raw conv1_2 level on the f16x3 MFMA (see sample_tokens_kernel: the fp32 form is 128 dependent
// 8-pass MFMAs per task); a lane then owns 8 consecutive channels of each 32-channel block
template <bool F16>
__global__ __launch_bounds__(256) void sample_tokens_gt_kernel(const SampleGtArgs a) {
    __shared__ __attribute__((aligned(16))) float s_w[8 * 4 * 256];  // 32 KiB: fc_local[0][:, :64] fragments (fp32 [8][4] or f16 hi|lo [8][2])
    {
        const float* wsrc = F16 ? a.wraw16 : a.wraw;
        for (int i = threadIdx.x; i < 8 * 4 * 64; i += 256) st4(s_w + 4 * i, ld4(wsrc + 4 * i));
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int T = a.n_slices + 1, S = a.size;
    const long nb = gridDim.x;
    const long bb = (nb % 8 == 0) ? (long)(blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : (long)blockIdx.x;
    const long chunk = (a.g_count + nb - 1) / nb;
    const long g_lo = bb * chunk, g_hi = g_lo + chunk < a.g_count ? g_lo + chunk : a.g_count;
    for (long gi = g_lo; gi < g_hi; ++gi) {
        const long grp = a.g_begin + gi;
        const int b = (int)(grp / a.groups_per_batch);
        long q = (grp % a.groups_per_batch) * S3D_GROUP + m;
        if (q >= a.n_qry) q = a.n_qry - 1;
        float x, y, z, gx, gy;
        gt_query_xyz(a.qry, a.rot, a.flip_yz, a.perm, a.nx, a.box, b, q, a.n_qry, x, y, z);
        project(a.trans + b * 12, x, y, z, gx, gy);
        for (int t = wave; t < T; t += 4) {
            f32x4 acc[8];
            f32x4 braw[4] = {zero4(), zero4(), zero4(), zero4()};
            if (t == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = zero4();
            } else {
                const long img = (long)b * a.n_slices + (t - 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = ld4(a.bias + 16 * j + 4 * g);
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    const int W = S >> (4 - l);
                    const Tap4 tp = make_taps(gx, gy, W, W);
                    const float* base = a.proj[l] + img * (long)W * W * 128 + 4 * g;
#pragma unroll
                    for (int kp = 0; kp < 2; ++kp) {
                        f32x4 v[2][8];
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2) {
                            const float* p = base + (long)tp.off[2 * kp + k2] * 128;
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[k2][j] = ld4(p + 16 * j);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2) {
                            const float w = tp.w[2 * kp + k2];
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[j] += v[k2][j] * w;
                        }
                    }
                }
                {
                    const Tap4 tp = make_taps(gx, gy, S, S);
                    const float* base = a.fine + img * (long)S * S * 64 + (F16 ? 8 : 4) * g;
                    f32x4 v[4][4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            v[k][u] = ld4(base + (long)tp.off[k] * 64 + (F16 ? 32 * (u >> 1) + 4 * (u & 1) : 16 * u));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        braw[u] = v[0][u] * tp.w[0] + v[1][u] * tp.w[1] + v[2][u] * tp.w[2] + v[3][u] * tp.w[3];
                }
                if (F16) {
                    const _Float16* sw = reinterpret_cast<const _Float16*>(s_w);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {   // k-slot 8g + t of step kk <-> raw channel 32 kk + 8g + t
                        const float x8[8] = {braw[2 * kk][0], braw[2 * kk][1], braw[2 * kk][2], braw[2 * kk][3],
                                             braw[2 * kk + 1][0], braw[2 * kk + 1][1], braw[2 * kk + 1][2], braw[2 * kk + 1][3]};
                        s3d_half8 bh, bl;
                        s3d_split8(x8, bh, bl);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const s3d_half8 fh = *reinterpret_cast<const s3d_half8*>(sw + (j * 2 + kk) * 1024 + lane * 8);
                            const s3d_half8 fl = *reinterpret_cast<const s3d_half8*>(sw + (j * 2 + kk) * 1024 + 512 + lane * 8);
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, bl, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, bh, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, bh, acc[j], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            acc[j] = mfma4(ld4(s_w + ((j * 4 + u) * 64 + lane) * 4), braw[u], acc[j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[j][i] = fmaxf(acc[j][i], 0.f);
            }
            // full 128-byte lines per query row (s3d_full_line_pair, common.h): tiles 2J, 2J + 1 of query m are exchanged
            // with lane m ^ 8; one instruction then writes queries 0-7, the next queries 8-15
            float* o = a.X + ((gi * T + t) * S3D_GROUP + (m & 7)) * 128 + 16 * (m >> 3) + 4 * g;
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                f32x4 va, vb;
                s3d_full_line_pair(acc[2 * J], acc[2 * J + 1], m, va, vb);
                st4(o + 32 * J, va);
                st4(o + 8 * 128 + 32 * J, vb);
            }
            if (a.raw_out) {
                float* ro = a.raw_out + ((gi * T + t) * S3D_GROUP + m) * 64 + (F16 ? 8 : 4) * g;
#pragma unroll
                for (int u = 0; u < 4; ++u) st4(ro + (F16 ? 32 * (u >> 1) + 4 * (u & 1) : 16 * u), braw[u]);
            }
        }
    }
}

int launch_sample_tokens_gt(const SampleGtArgs& a, hipStream_t stream) {
    S3D_CHECK_ARG(a.size % 16 == 0 && a.size >= 16, "sample_gt: size %d", a.size);
    S3D_CHECK_ARG(a.n_slices >= 1 && a.n_slices + 1 <= S3D_N_TOKENS_MAX, "sample_gt: n_slices %d", a.n_slices);
    const long cap = sampler_cu_count();   // one workgroup per CU, as launch_sample_tokens
    long blocks = a.g_count < cap ? a.g_count : cap;
    if (blocks <= 0) return 0;
    if (blocks >= 8) blocks -= blocks % 8;
    if (a.wraw16)
        hipLaunchKernelGGL(sample_tokens_gt_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(sample_tokens_gt_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    S3D_LAUNCH_CHECK();
    return 0;
}

// pts_feat_extractor (model_gt.py:24-31): thread = (query row, 4 output channels); the three small weight
// matrices sit in LDS (3*32 + 32*64 + 64*128 floats = 41 KiB)
__global__ __launch_bounds__(256) void gt_point_tokens_kernel(const GtPointArgs a) {
    __shared__ float s_w0[32 * 3], s_b0[32], s_w1[64 * 32], s_b1[64], s_w2[128 * 64], s_b2[128];
    __shared__ float s_h1[8][32], s_h2[8][64];
    for (int i = threadIdx.x; i < 96; i += 256) s_w0[i] = a.w0[i];
    for (int i = threadIdx.x; i < 32; i += 256) s_b0[i] = a.b0[i];
    for (int i = threadIdx.x; i < 2048; i += 256) s_w1[i] = a.w1[i];
    for (int i = threadIdx.x; i < 64; i += 256) s_b1[i] = a.b1[i];
    for (int i = threadIdx.x; i < 8192; i += 256) s_w2[i] = a.w2[i];
    for (int i = threadIdx.x; i < 128; i += 256) s_b2[i] = a.b2[i];
    __syncthreads();
    const int T = a.n_slices + 1;
    const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;   // 8 query rows per pass, 32 threads each
    const long rows = a.g_count * S3D_GROUP;
    for (long r0 = (long)blockIdx.x * 8; r0 < rows; r0 += (long)gridDim.x * 8) {
        const long r = r0 + sub;
        const bool ok = r < rows;
        const long gi = (ok ? r : rows - 1) / S3D_GROUP;
        const int m = (int)((ok ? r : rows - 1) % S3D_GROUP);
        const long grp = a.g_begin + gi;
        const int b = (int)(grp / a.groups_per_batch);
        long q = (grp % a.groups_per_batch) * S3D_GROUP + m;
        if (q >= a.n_qry) q = a.n_qry - 1;
        float x, y, z;
        gt_query_xyz(a.qry, a.rot, a.flip_yz, a.perm, a.nx, a.box, b, q, a.n_qry, x, y, z);
        s_h1[sub][l] = fmaxf(s_w0[l * 3] * x + s_w0[l * 3 + 1] * y + s_w0[l * 3 + 2] * z + s_b0[l], 0.f);
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int c = l + 32 * o;
            float s = s_b1[c];
#pragma unroll 8
            for (int k = 0; k < 32; ++k) s += s_w1[c * 32 + k] * s_h1[sub][k];
            s_h2[sub][c] = fmaxf(s, 0.f);
        }
        __syncthreads();
        f32x4 out;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 4 * l + i;
            float s = s_b2[c];
#pragma unroll 8
            for (int k = 0; k < 64; ++k) s += s_w2[c * 64 + k] * s_h2[sub][k];
            out[i] = fmaxf(s, 0.f);
        }
        if (ok) st4(a.X + ((gi * T) * S3D_GROUP + m) * 128 + 4 * l, out);
        if (ok && a.h1_out) {
            a.h1_out[r * 32 + l] = s_h1[sub][l];
            a.h2_out[r * 64 + l] = s_h2[sub][l];
            a.h2_out[r * 64 + 32 + l] = s_h2[sub][32 + l];
        }
        __syncthreads();
    }
}

int launch_gt_point_tokens(const GtPointArgs& a, hipStream_t stream) {
    if (a.g_count <= 0) return 0;
    const long rows = a.g_count * S3D_GROUP;
    const long blocks = (rows + 7) / 8 < 2048 ? (rows + 7) / 8 : 2048;
    hipLaunchKernelGGL(gt_point_tokens_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    S3D_LAUNCH_CHECK();
    return 0;
}
