// attnq.h — shared pieces of the query-major attention kernels (decode_attnq.hip: inference / training forward;
// train_attnq.hip: fused training backward): split-precision operand helpers, lane-swap reductions, hand-issued LDS reads
// and the weight ring's barrier.  See decode_attnq.hip for the layout story.
#pragma once
#include "decode.h"
#include "dropout.h"

typedef _Float16 half8q __attribute__((ext_vector_type(8)));

#define AQ_WIN_HALFS (24 * 1024)   // per head: 24 fragment pairs (q0,q1,k0,k1,v0,v1) x 4 k-steps, hi|lo = 48 KiB
#define AQ_WO_HALFS (8 * 1024)     // per head: 8 fragment pairs = 16 KiB

__device__ __forceinline__ half8q ldq8(const _Float16* p) { return *reinterpret_cast<const half8q*>(p); }
// BF (S3D_PREC_BF16): single pass on the bf16 MFMA; the 16-bit lanes of the "hi" operands then hold bf16 bit patterns
typedef __bf16 bf8q __attribute__((ext_vector_type(8)));
typedef short short4q __attribute__((ext_vector_type(4)));
template <bool SINGLE, bool BF = false>
__device__ __forceinline__ f32x4 mfma3q(const half8q ah, const half8q al, const half8q bh, const half8q bl, f32x4 c) {
    if (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8q, ah), __builtin_bit_cast(bf8q, bh), c, 0, 0, 0);
    if (!SINGLE) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
    return c;
}
// hi/lo split of a pair: one v_cvt_pk_f16_f32 + two v_fma_mix{lo,hi}_f16 (lo = f16(x - f32(hi)), the subtraction is
// exact, one rounding: the same value a scalar convert - subtract - convert produces)
typedef _Float16 half2q __attribute__((ext_vector_type(2)));
typedef float float2q __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2q(float a, float b, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(float2q{a, b}, half2q));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
}
typedef unsigned uint4q __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8pk(const f32x4 a, const f32x4 b, half8q& hi, half8q& lo) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split2q(a[0], a[1], h0, l0);
    split2q(a[2], a[3], h1, l1);
    split2q(b[0], b[1], h2, l2);
    split2q(b[2], b[3], h3, l3);
    hi = __builtin_bit_cast(half8q, uint4q{h0, h1, h2, h3});
    lo = __builtin_bit_cast(half8q, uint4q{l0, l1, l2, l3});
}
// bf16 forms of the two splits (BF mode: only the high halves are operands; the low halves mirror them and are never read)
typedef __bf16 bf2q __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16_pair_q(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(float2q{a, b}, bf2q));
}
template <bool BF>
__device__ __forceinline__ void split8x(const f32x4 a, const f32x4 b, half8q& hi, half8q& lo) {
    if (BF) {
        hi = __builtin_bit_cast(half8q, uint4q{bf16_pair_q(a[0], a[1]), bf16_pair_q(a[2], a[3]), bf16_pair_q(b[0], b[1]), bf16_pair_q(b[2], b[3])});
        lo = hi;
    } else {
        split8pk(a, b, hi, lo);
    }
}
// four values -> the A / B operand of the 16-deep MFMA (v_mfma_f32_16x16x16_f16: lane group g carries k = 4g..4g+3)
typedef _Float16 half4q __attribute__((ext_vector_type(4)));
typedef unsigned uint2q __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4pk(const f32x4 a, half4q& hi, half4q& lo) {
    unsigned h0, h1, l0, l1;
    split2q(a[0], a[1], h0, l0);
    split2q(a[2], a[3], h1, l1);
    hi = __builtin_bit_cast(half4q, uint2q{h0, h1});
    lo = __builtin_bit_cast(half4q, uint2q{l0, l1});
}
template <bool BF>
__device__ __forceinline__ void split4x(const f32x4 a, half4q& hi, half4q& lo) {
    if (BF) {
        hi = __builtin_bit_cast(half4q, uint2q{bf16_pair_q(a[0], a[1]), bf16_pair_q(a[2], a[3])});
        lo = hi;
    } else {
        split4pk(a, hi, lo);
    }
}
template <bool SINGLE, bool BF = false>
__device__ __forceinline__ f32x4 mfma3h(const half4q ah, const half4q al, const half4q bh, const half4q bl, f32x4 c) {
    if (BF) return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4q, ah), __builtin_bit_cast(short4q, bh), c, 0, 0, 0);
    if (!SINGLE) {
        c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, c, 0, 0, 0);
    return c;
}
// reductions over the 4 lane groups g of one column (l & 15) with the gfx950 lane-swap instructions (no LDS crossbar):
// v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second, so two copies
// of v become {lo, lo} and {hi, hi}; v_permlane16_swap does the same with odd / even rows of 16.  Issued as asm: the
// __builtin_amdgcn_permlane*_swap builtins of this hipcc return the FIRST result for both elements (checked on the
// GPU with build/t-style unit kernels); s_nop 1 = the VALU-write -> permlane hazard the compiler would have padded.
__device__ __forceinline__ void lane_swap32(float& x, float& y) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
__device__ __forceinline__ void lane_swap16(float& x, float& y) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
__device__ __forceinline__ float colsum16(float v) {
    float x = v, y = v;
    lane_swap32(x, y);
    x += y;
    y = x;
    lane_swap16(x, y);
    return x + y;
}
__device__ __forceinline__ float colmax16(float v) {
    float x = v, y = v;
    lane_swap32(x, y);
    x = fmaxf(x, y);
    y = x;
    lane_swap16(x, y);
    return fmaxf(x, y);
}

// ---------------------------------------------------------------------------------------------
// Four waves per workgroup, TWO workgroups per CU (one computes while the other sits at a barrier); a workgroup owns
// half a group (8 queries), a wave two of them.  The rows of the wave's two queries are loaded and split ONCE per
// item and stay in registers as f16 hi / lo fragments for the four heads (round 3: the high halves used to be parked in a
// wave-private LDS region and re-read with every k-step — 2 of 6 fragment reads).  The weight ring holds QUARTER-head
// slots of 16 KiB (q | k | v | out_proj fragments), FOUR of them, filled by LDS-DMA three phases ahead; four barriers per
// head; LDS = 4 x 16 KiB ring + 3 KiB of small vectors.
// (Earlier versions: eight waves / whole-head slots / rows re-read per head: 1.16 ms per layer; four waves / half-head
// slots: 1.13 ms; quarter-head slots, two-slot ring: 0.94 ms.  Bench stage, 4 launches + the last layer: 7.41 ms with the
// two-slot ring, 6.9 with full-line stores, 6.86 with the rows in registers, 6.74 with the four-slot ring.)
// ---------------------------------------------------------------------------------------------
// LDS reads and their counted waits are issued by hand (same finding as in decode_f16.hip's pipelined FFN): with an
// LDS-DMA refill in flight hipcc turns every LDS wait of this single-LDS-object kernel into lgkmcnt(0), i.e. it waits
// for the fragment reads it has just issued for the NEXT k-step (all 42 waits of the previous build were lgkmcnt(0)).
#define AQ_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define AQ_WAIT6(n, a, b, c, d, e, f) \
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(n))
#define AQ_WAIT4(n, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(n))
#define AQ_READ32(dst, addr, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
// Four-slot ring, DMA three phases ahead: at a barrier the slot of the phase that starts must have landed, and at most two
// newer DMA sets (2 x 4 instructions per wave) have been issued since — whatever else is in flight (row loads, stores of
// the previous item) is older or only makes the wait stricter.  vmcnt retires in order.  Raw s_barrier: the kernel has no
// compiler-visible LDS access after its prologue, so nothing needs the fence __syncthreads() carries (which would drain vmcnt).
#define AQ_BARRIER()                                       \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       \
    asm volatile("s_barrier" ::: "memory")
// The same barrier with the EXACT number of vector-memory operations a wave has issued after the DMA set it waits for (round
// 4).  vmcnt(8) is always safe — at least the two newer DMA sets are younger — but it also waits for all but eight of whatever
// else was issued since: the previous item's row stores and the next rows' loads at the first barriers of an item (32 - 48
// operations), the per-head stores of the training kernels.  n must not exceed the real count (a larger n lets the barrier
// pass before the set has landed), so the callers use these counts only where every counted operation is issued
// unconditionally (T >= 9: both halves of every full-line store pair have active lanes) and fall back to 8 otherwise.
// The counts are hand-derived from the source and hold only while hipcc emits exactly the counted operations (a dropped or
// merged one would let a slot be read before its DMA lands: silently wrong rows).  -DS3D_AQ_SAFE_BARRIERS builds every
// counted barrier as the always-safe vmcnt(8) form; `make` also builds that variant (libslice3d_hip_safe.so) and
// tests/test_gpu_barriers.py holds the two libraries' outputs bit-identical.
#ifdef S3D_AQ_SAFE_BARRIERS
#define AQ_BARRIER_N(n) AQ_BARRIER()
#else
#define AQ_BARRIER_N(n)                                          \
    asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");        \
    asm volatile("s_barrier" ::: "memory")
#endif
#define AQ3_SLOT_HALFS (8 * 1024)    // 8 fragment pairs = 16 KiB
