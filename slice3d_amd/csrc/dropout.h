// dropout.h — counter-based dropout masks (no stored masks: forward and backward regenerate them).
// keep(seed, site, idx) is a pure function; value = keep ? 1/(1-p) : 0.  Sites: 4*layer + {0 attention
// probabilities, 1 attention-block output, 2 FFN hidden, 3 FFN output} (nn.TransformerEncoderLayer's
// dropout, dropout1, dropout, dropout2 — reference models.py:18-19 uses the default p = 0.1).
#pragma once
#include <stdint.h>

struct DropCfg {
    unsigned long long seed;
    float p;          // 0 = off
    float scale;      // 1/(1-p)
    unsigned thresh;  // keep iff the element's 16-bit hash field >= thresh
    int site;
};

// Stream key of a (seed, site) pair: both words of the seed and the site go through full avalanche rounds BEFORE they
// meet the element index, so the streams of different steps / sites / ranks are not XOR re-indexings of one table
// (seed ^ idx alone would make stream(seed a)[i] == stream(seed b)[i ^ a ^ b]).  Uniform per launch: the compiler
// hoists it out of the element loops (scalar ALU).
__host__ __device__ inline unsigned s3d_mix32(unsigned x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__host__ __device__ inline unsigned s3d_stream_key(unsigned long long seed, int site) {
    unsigned k = s3d_mix32((unsigned)seed + 0x9E3779B9u);
    k = s3d_mix32(k ^ (unsigned)(seed >> 32)) + (unsigned)(site + 1) * 0xC2B2AE3Du;
    return s3d_mix32(k);
}
// 32-bit mixing only (three multiply / xor-shift rounds of the murmur3 finaliser on the keyed counter; the key is
// ADDED, the high index word multiplied in): the FFN forward draws 2 176 masks per token row, 64-bit multiplies there
// cost a millisecond per layer.
__host__ __device__ inline unsigned s3d_hash32_rounds(unsigned x) {
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    return x;
}
// the keyed counter the rounds start from; a kernel that walks counters of the form base | small (no carry into base) can
// precompute (unsigned)base ^ hi(base) * K once and XOR the small part in: s3d_hash32_rounds((that ^ small) + key)
__host__ __device__ inline unsigned s3d_hash32_fold(unsigned long long idx) {
    return (unsigned)idx ^ ((unsigned)(idx >> 32) * 0x9E3779B1u);
}
__host__ __device__ inline unsigned s3d_hash32(unsigned long long seed, int site, unsigned long long idx) {
    return s3d_hash32_rounds(s3d_hash32_fold(idx) + s3d_stream_key(seed, site));
}
// One full hash serves FOUR consecutive elements (idx >> 2): two 16-bit fields of the hash word and two of a cheap
// remix of it.  keep iff field >= thresh (thresh = round(p * 65536): p = 0.1 -> 6554 / 65536).  s3d_drop4 draws the four
// masks of an aligned quad at once — the FFN forward's 2 048 hidden masks per row cost 512 hashes instead of 2 048.
__host__ __device__ inline unsigned s3d_drop_remix(unsigned h) {
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    return h;
}
__host__ __device__ inline float s3d_drop(const DropCfg& d, unsigned long long idx) {
    unsigned h = s3d_hash32(d.seed, d.site, idx >> 2);
    if (idx & 2) h = s3d_drop_remix(h);
    const unsigned f = (idx & 1) ? h >> 16 : h & 0xFFFFu;
    return f >= d.thresh ? d.scale : 0.f;
}
// masks of elements base .. base+3, base % 4 == 0 (the same values s3d_drop gives one by one)
__host__ __device__ inline void s3d_drop4(const DropCfg& d, unsigned long long base, float (&out)[4]) {
    const unsigned h = s3d_hash32(d.seed, d.site, base >> 2), h2 = s3d_drop_remix(h);
    out[0] = (h & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
    out[1] = (h >> 16) >= d.thresh ? d.scale : 0.f;
    out[2] = (h2 & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
    out[3] = (h2 >> 16) >= d.thresh ? d.scale : 0.f;
}
static inline DropCfg make_drop(unsigned long long seed, float p, int site) {
    DropCfg d;
    d.seed = seed; d.p = p; d.site = site;
    d.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    d.thresh = p > 0.f ? (unsigned)((double)p * 65536.0 + 0.5) : 0u;
    return d;
}
