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
    unsigned thresh;  // keep iff hash >= thresh
    int site;
};

__host__ __device__ inline unsigned s3d_hash32(unsigned long long seed, int site, unsigned long long idx) {
    unsigned long long x = idx * 0x9E3779B97F4A7C15ull + seed + (unsigned long long)(site + 1) * 0xD1B54A32D192ED03ull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return (unsigned)x;
}
__host__ __device__ inline float s3d_drop(const DropCfg& d, unsigned long long idx) {
    return s3d_hash32(d.seed, d.site, idx) >= d.thresh ? d.scale : 0.f;
}
static inline DropCfg make_drop(unsigned long long seed, float p, int site) {
    DropCfg d;
    d.seed = seed; d.p = p; d.site = site;
    d.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    d.thresh = p > 0.f ? (unsigned)((double)p * 4294967296.0) : 0u;
    return d;
}
