// common.h — shared device helpers for libslice3d_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/slice3d_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// MFMA  v_mfma_f32_16x16x4_f32  (exact fp32: bitwise a k-ordered fmaf chain).
// Operand maps (cdna_hip_programming.md section 3):  lane l:  A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15],  C/D[row = (l>>4)*4 + reg][col = l&15].
// Every GEMM in this library is written in the "swapped" form  D^T = W * X^T :
//   A <- weight rows (output channel r = l&15, k-slot g = l>>4)
//   B <- activation rows (row / pixel / query m = l&15, k-slot g = l>>4)
//   D[reg i] = out[m = l&15][n = 4*g + i]
// so a lane owns 4 CONSECUTIVE output channels of ONE activation row (16-byte stores), and the D
// registers of one GEMM are directly the B operand of the next one (k-slot g <-> channel 4*g+i).
// A 16-deep K chunk is consumed by 4 MFMA steps e = 0..3 with k-slot g <-> k = 4*g + e, so each
// operand fragment of a chunk is one 16-byte (f32x4) load per lane.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma4(const f32x4 a, const f32x4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// sum over the 4 lanes that share l&15 (the 4 k-slot groups of one activation row)
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Packed weight fragment image: W[N][K] (N, K multiples of 16) is stored as
//   P[(j*KU + u)*256 + lane*4 + e] = W[16*j + (lane&15)][16*u + 4*(lane>>4) + e],   KU = K/16
// so the A fragment of (row tile j, k chunk u) is one contiguous, lane-linear 1 KiB block:
// coalesced from global, conflict-free from LDS.
__device__ __forceinline__ const float* frag_ptr(const float* packed, int j, int u, int KU, int lane) {
    return packed + ((size_t)(j * KU + u) * 64 + lane) * 4;
}

// ---------------------------------------------------------------------------------------------
// host-side error plumbing
// ---------------------------------------------------------------------------------------------
void s3d_set_error(const char* fmt, ...);
#define S3D_CHECK_ARG(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            s3d_set_error(__VA_ARGS__);     \
            return S3D_E_ARG;               \
        }                                   \
    } while (0)
#define S3D_LAUNCH_CHECK()                                            \
    do {                                                              \
        hipError_t e__ = hipGetLastError();                           \
        if (e__ != hipSuccess) {                                      \
            s3d_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return (int)e__;                                          \
        }                                                             \
    } while (0)

#define TRY_RET(x)               \
    do {                         \
        int rc__ = (x);          \
        if (rc__) return rc__;   \
    } while (0)

// hi/lo split of a pair for the split-precision MFMA paths: one v_cvt_pk_f16_f32 + two v_fma_mix{lo,hi}_f16
// (lo = f16(x - f32(hi)): the subtraction is exact, one rounding — the same value a scalar convert - subtract - convert
// produces), 1.5 VALU per value.  The halves are written by 16-bit partial-register asm ops whose write -> MFMA-read
// spacing the compiler does not pad: put S3D_SPLIT_SETTLE() between a group of splits and MFMAs that read them straight
// from registers (values that go through LDS need nothing).
typedef _Float16 s3d_half2 __attribute__((ext_vector_type(2)));
typedef _Float16 s3d_half4 __attribute__((ext_vector_type(4)));
typedef _Float16 s3d_half8 __attribute__((ext_vector_type(8)));
typedef float s3d_float2 __attribute__((ext_vector_type(2)));
typedef unsigned s3d_uint2 __attribute__((ext_vector_type(2)));
typedef unsigned s3d_uint4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void s3d_split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(s3d_float2{a, b}, s3d_half2));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
}
__device__ __forceinline__ void s3d_split8(const float (&x)[8], s3d_half8& hi, s3d_half8& lo) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    s3d_split2(x[0], x[1], h0, l0);
    s3d_split2(x[2], x[3], h1, l1);
    s3d_split2(x[4], x[5], h2, l2);
    s3d_split2(x[6], x[7], h3, l3);
    hi = __builtin_bit_cast(s3d_half8, s3d_uint4{h0, h1, h2, h3});
    lo = __builtin_bit_cast(s3d_half8, s3d_uint4{l0, l1, l2, l3});
}
__device__ __forceinline__ void s3d_split4(const f32x4 a, s3d_half4& hi, s3d_half4& lo) {
    unsigned h0, h1, l0, l1;
    s3d_split2(a[0], a[1], h0, l0);
    s3d_split2(a[2], a[3], h1, l1);
    hi = __builtin_bit_cast(s3d_half4, s3d_uint2{h0, h1});
    lo = __builtin_bit_cast(s3d_half4, s3d_uint2{l0, l1});
}
// v if bit `pos` of `bits` is set, else +0: the 1-bit signed field is the AND mask itself, 2 VALU (asm: hipcc rewrites the
// builtin form into and + compare + select)
__device__ __forceinline__ float s3d_gate_bit(float v, unsigned bits, int pos) {
    unsigned mk;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mk) : "v"(bits), "v"(pos));
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & mk);
}
__device__ __forceinline__ float s3d_gate_bit_imm(float v, unsigned bits, int pos) {   // pos: a compile-time constant
    unsigned mk;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mk) : "v"(bits), "n"(pos));
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & mk);
}
#define S3D_SPLIT_SETTLE()                          \
    __builtin_amdgcn_sched_barrier(0);              \
    asm volatile("s_nop 15" ::: "memory");          \
    __builtin_amdgcn_sched_barrier(0);

// Full-line stores out of the swapped-form accumulator layout.  Lane (m = row, g) of a D tile holds 4 consecutive output
// channels, so one store instruction writes 64-byte runs of 16 rows, and the two tiles nt = 0, 1 that complete a row's
// 128-byte line go out in different instructions.  Exchanging the tiles between lanes m and m ^ 8 (a rotation by 8 inside
// the 16-lane DPP row) gives instruction A rows 0-7 and instruction B rows 8-15 of the tile pair, each row as one full
// line: lanes m < 8 carry channels 0-15, lanes m >= 8 channels 16-31 of row m & 7 (+ 8).  Measured on the 128 -> 384
// row-linear layer: 3.10 -> 2.59 ms (tools/lin_abl.sh).
// (issued as asm: with __builtin_amdgcn_update_dpp on the four elements hipcc emitted ONE v_mov_b32_dpp and used its result
// for all four — seen in the ISA and as scrambled rows on the GPU; s_nop 1 = the VALU-write -> DPP-read wait states the
// compiler would have added)
__device__ __forceinline__ f32x4 s3d_row_ror8(const f32x4 v) {
    float r0, r1, r2, r3;
    asm volatile("s_nop 1\n\t"
        "v_mov_b32_dpp %0, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %2, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %3, %7 row_ror:8 row_mask:0xf bank_mask:0xf"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    return f32x4{r0, r1, r2, r3};
}
__device__ __forceinline__ unsigned s3d_row_ror8_u32(unsigned v) {
    unsigned r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(v));
    return r;
}
// (v0, v1) = tiles nt = 0, 1 of row m  ->  (a, b) = this lane's 16 bytes of row (m & 7) and of row (m & 7) + 8, both at
// channel offset 16 (m >> 3) + 4 g of the 32-channel pair
__device__ __forceinline__ void s3d_full_line_pair(const f32x4 v0, const f32x4 v1, int m, f32x4& a, f32x4& b) {
    const f32x4 t0 = s3d_row_ror8(v0), t1 = s3d_row_ror8(v1);
    const bool lo = m < 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = lo ? v0[i] : t1[i];
        b[i] = lo ? t0[i] : v1[i];
    }
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Per-device one-time kernel attributes.  hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: a
// process that drives a second GPU must raise that device's limit too, so the "done" state is a bit per device ordinal
// (setting the attribute twice from two racing threads is harmless; the bit is published only after the calls returned).
// Returns 0 or the hipError_t of the failing call.
#include <atomic>
#include <initializer_list>
static inline int s3d_set_max_lds(std::atomic<unsigned long long>& done, std::initializer_list<const void*> kernels, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    for (const void* k : kernels) {
        const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            s3d_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) on device %d: %s", bytes, dev, hipGetErrorString(e));
            return (int)e;
        }
    }
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}
// CU count of the current device (cached per device ordinal)
static inline int s3d_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int v = cache[dev & 63].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cache[dev & 63].store(v, std::memory_order_relaxed);
    }
    return v;
}

// Workgroup barrier that publishes LDS-DMA (global_load_lds) data: a wave must see ITS OWN requests land before it
// arrives, because the other waves read those bytes right after the barrier.  __syncthreads() alone does not wait
// on vmcnt (the compiler only guards a wave's own later LDS reads), which left a window in which a wave could read
// a chunk another wave's DMA had not delivered yet (seen as rare garbage rows once nothing else delayed the reads).
__device__ __forceinline__ void dma_publish_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
