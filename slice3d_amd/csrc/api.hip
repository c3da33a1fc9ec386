// api.hip — exported C ABI (include/slice3d_hip.h): weight packing + kernel orchestration.
#include <dlfcn.h>
#include <stdarg.h>
#include <string.h>

#include <vector>
#include <mutex>

#include <math.h>

#include "conv.h"
#include "decode.h"
#include "train.h"
#include "ldm_ops.h"

static thread_local char g_err[512] = "";

void s3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int s3d_version(void) { return S3D_VERSION; }
extern "C" const char* s3d_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// roctx ranges (tracing): every tracked stage of the inference path and every phase of the train
// step is bracketed by roctxRangePush / roctxRangePop, so a `rocprofv3 --marker-trace --kernel-trace` run can be read by
// phase.  The roctx library is looked up at run time (no link-time dependency; without it the ranges are no-ops).
// ---------------------------------------------------------------------------------------------
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        // rocprofv3 (rocprofiler-sdk) records the ranges of ITS roctx library; the roctracer-era libroctx64 is the fallback
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_GLOBAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_LAZY | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
static const Roctx& roctx() {
    static const Roctx r;
    return r;
}
// one open range at a time: next(name) closes the previous phase and opens the next; the destructor closes the last one
// (also on the early returns of TRY)
struct RangeSeq {
    bool open = false;
    void next(const char* name) {
        const Roctx& r = roctx();
        if (!r.push) return;
        if (open) r.pop();
        r.push(name);
        open = true;
    }
    ~RangeSeq() {
        if (open) roctx().pop();
    }
};
static const char* const kProfRangeNames[] = {"s3d:unet_encode", "s3d:latent_build", "s3d:sample_tokens", "s3d:attn_layer",
                                               "s3d:ffn_layer", "s3d:ffn_final", "s3d:vgg_loss", "s3d:sample_pyramid"};

// ---------------------------------------------------------------------------------------------
// optional event profiler: brackets the tracked launches with hipEvents ON THE CALLER'S STREAM so
// bench.py can report per-kernel durations measured live inside its timed region.
// ---------------------------------------------------------------------------------------------
static bool g_prof = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_ev[S3D_PROF_N];

struct ProfScope {
    int id;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    RangeSeq range;
    ProfScope(int id_, hipStream_t st_) : id(id_), st(st_) {
        range.next(kProfRangeNames[id]);
        if (!g_prof) return;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, st);
        g_prof_ev[id].push_back({a, b});
    }
};

extern "C" int s3d_range_push(const char* name) {
    const Roctx& r = roctx();
    return r.push && name ? (r.push(name), 1) : 0;
}
extern "C" int s3d_range_pop(void) {
    const Roctx& r = roctx();
    return r.pop ? (r.pop(), 1) : 0;
}

extern "C" int s3d_prof_enable(int on) {
    for (int i = 0; i < S3D_PROF_N; ++i) {
        for (auto& e : g_prof_ev[i]) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        g_prof_ev[i].clear();
    }
    g_prof = on != 0;
    return 0;
}

extern "C" int s3d_prof_read(int id, double* total_ms, long* count) {
    S3D_CHECK_ARG(id >= 0 && id < S3D_PROF_N && total_ms && count, "prof_read: bad argument");
    double tot = 0;
    for (auto& e : g_prof_ev[id]) {
        (void)hipEventSynchronize(e.second);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e.first, e.second);
        tot += ms;
    }
    *total_ms = tot;
    *count = (long)g_prof_ev[id].size();
    return 0;
}

#define TRY(x)              \
    do {                    \
        int rc__ = (x);     \
        if (rc__) return rc__; \
    } while (0)

// =============================================================================================
// U-Net
// =============================================================================================
static const int kEncCin[13] = {3, 64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512};
static const int kEncCout[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
static const bool kEncTap[13] = {false, true, false, true, false, false, true,
                                 false, false, true, false, false, true};
static const int kUpC[4] = {512, 256, 128, 64};  // input channels of up1..4; Ct = C/2

struct PackedConv {
    size_t w, scale, shift;  // float offsets
    size_t w16;              // split-precision (f16 hi/lo) image, same size as w
    int cout_pad, KU;
};
struct UNetLayout {
    PackedConv enc[13];
    size_t pool_scale[4], pool_shift[4];  // BN that follows tap convs 1,3,6,9 (applied by the pool kernel)
    PackedConv trans_c, trans_up[4], up_t[4], up_c1[4], up_c2[4], outc;
    size_t emds;
    size_t enc0_raw;   // conv1_1's weight as is, (64,3,3,3): the 3-channel first layer runs a direct fp32 kernel on the NCHW image
    size_t total;
};

static int pad16(int c) { return (c + 15) / 16 * 16; }

static UNetLayout unet_layout(int n_slices) {
    UNetLayout L;
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t o = off;
        off += (n + 3) / 4 * 4;
        return o;
    };
    auto conv = [&](PackedConv& pc, int cout_pad, int KU) {
        pc.cout_pad = cout_pad;
        pc.KU = KU;
        pc.w = take((size_t)cout_pad * KU * 16);
        pc.w16 = take((size_t)cout_pad * KU * 16);
        pc.scale = take(cout_pad);
        pc.shift = take(cout_pad);
    };
    for (int i = 0; i < 13; ++i) conv(L.enc[i], kEncCout[i], 9 * pad16(kEncCin[i]) / 16);
    const int tapc[4] = {64, 128, 256, 512};
    for (int i = 0; i < 4; ++i) {
        L.pool_scale[i] = take(tapc[i]);
        L.pool_shift[i] = take(tapc[i]);
    }
    conv(L.trans_c, 512, 640 / 16);
    for (int i = 0; i < 4; ++i) {
        const int C = kUpC[i], Ct = C / 2;
        conv(L.trans_up[i], Ct, C / 16);
        conv(L.up_t[i], 4 * Ct, C / 16);
        conv(L.up_c1[i], Ct, 2 * 9 * Ct / 16);
        conv(L.up_c2[i], Ct, 9 * Ct / 16);
    }
    conv(L.outc, 16, 32 / 16);
    L.emds = take((size_t)n_slices * 128);
    L.enc0_raw = take(64 * 27);
    L.total = off;
    return L;
}

extern "C" size_t s3d_unet_packed_bytes(int n_slices) { return unet_layout(n_slices).total * sizeof(float); }

static int pack_conv3(const float* w, float* dst, int cout, int cout_pad, int cin_tot, int cin_begin, int cseg,
                      int KU_total, int u_off, int taps, hipStream_t st, int f16 = 0) {
    PackArgs a = {};
    a.f16 = f16;
    a.src = w; a.dst = dst; a.kind = S3D_PACK_CONV;
    a.n_valid = cout; a.n_pad = cout_pad;
    a.KU_total = KU_total; a.u_off = u_off;
    a.cseg = pad16(cseg); a.cseg_valid = cseg; a.cin_tot = cin_tot; a.cin_begin = cin_begin; a.taps = taps;
    a.ku_seg = taps * a.cseg / 16;
    return launch_pack(a, st);
}

static int pack_linear(const float* w, float* dst, int n, int n_pad, int k, int ld, int chunk_ku, hipStream_t st,
                       int f16 = 0) {
    PackArgs a = {};
    a.f16 = f16;
    a.src = w; a.dst = dst; a.kind = S3D_PACK_LINEAR;
    a.n_valid = n; a.n_pad = n_pad;
    a.KU_total = pad16(k) / 16; a.u_off = 0; a.ku_seg = a.KU_total;
    a.ld = ld; a.k_valid = k; a.chunk_ku = chunk_ku;
    return launch_pack(a, st);
}

extern "C" int s3d_unet_pack(const S3dUNetParams* P, void* packed, size_t packed_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    S3D_CHECK_ARG(P && packed, "unet_pack: null argument");
    S3D_CHECK_ARG(P->n_slices >= 1 && P->n_slices <= 12, "unet_pack: n_slices %d", P->n_slices);
    const UNetLayout L = unet_layout(P->n_slices);
    if (packed_bytes < L.total * sizeof(float)) {
        s3d_set_error("unet_pack: packed buffer %zu < %zu bytes", packed_bytes, L.total * sizeof(float));
        return S3D_E_WORKSPACE;
    }
    float* base = (float*)packed;
    for (int i = 0; i < 13; ++i) {
        const PackedConv& pc = L.enc[i];
        TRY(pack_conv3(P->enc[i].w, base + pc.w, kEncCout[i], pc.cout_pad, kEncCin[i], 0, kEncCin[i], pc.KU, 0, 9,
                       st));
        if (kEncCin[i] % 32 == 0)
            TRY(pack_conv3(P->enc[i].w, base + pc.w16, kEncCout[i], pc.cout_pad, kEncCin[i], 0, kEncCin[i], pc.KU, 0,
                           9, st, 1));
        if (kEncTap[i])  // raw conv output is the skip tensor: bias only
            TRY(launch_fold_bn(P->enc[i].b, nullptr, base + pc.scale, base + pc.shift, kEncCout[i], pc.cout_pad, 1,
                               0, st));
        else
            TRY(launch_fold_bn(P->enc[i].b, P->enc[i].bn, base + pc.scale, base + pc.shift, kEncCout[i],
                               pc.cout_pad, 1, 1, st));
    }
    if (hipMemcpyAsync(base + L.enc0_raw, P->enc[0].w, 64 * 27 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        s3d_set_error("unet_pack: copy of conv1_1's weight failed");
        return S3D_E_ARG;
    }
    const int tapi[4] = {1, 3, 6, 9};
    for (int i = 0; i < 4; ++i)
        TRY(launch_fold_bn(nullptr, P->enc[tapi[i]].bn, base + L.pool_scale[i], base + L.pool_shift[i],
                           kEncCout[tapi[i]], kEncCout[tapi[i]], 1, 0, st));
    TRY(pack_linear(P->trans_c.w, base + L.trans_c.w, 512, 512, 640, 640, 0, st));
    TRY(pack_linear(P->trans_c.w, base + L.trans_c.w16, 512, 512, 640, 640, 0, st, 1));
    TRY(launch_fold_bn(P->trans_c.b, nullptr, base + L.trans_c.scale, base + L.trans_c.shift, 512, 512, 1, 0, st));
    for (int i = 0; i < 4; ++i) {
        const int C = kUpC[i], Ct = C / 2;
        TRY(pack_linear(P->trans_up[i].w, base + L.trans_up[i].w, Ct, Ct, C, C, 0, st));
        TRY(pack_linear(P->trans_up[i].w, base + L.trans_up[i].w16, Ct, Ct, C, C, 0, st, 1));
        TRY(launch_fold_bn(P->trans_up[i].b, nullptr, base + L.trans_up[i].scale, base + L.trans_up[i].shift, Ct,
                           Ct, 1, 0, st));
        {  // ConvTranspose2d weight [Cin][Ct][2][2] -> GEMM rows n = q*Ct + co
            PackArgs a = {};
            a.src = P->up_t[i].w; a.dst = base + L.up_t[i].w; a.kind = S3D_PACK_CONVT;
            a.n_valid = 4 * Ct; a.n_pad = 4 * Ct; a.KU_total = C / 16; a.u_off = 0; a.ku_seg = C / 16;
            a.k_valid = C; a.ct = Ct;
            TRY(launch_pack(a, st));
            a.f16 = 1; a.dst = base + L.up_t[i].w16;
            TRY(launch_pack(a, st));
            TRY(launch_fold_bn(P->up_t[i].b, nullptr, base + L.up_t[i].scale, base + L.up_t[i].shift, Ct, 4 * Ct, 4,
                               0, st));
        }
        // DoubleConv conv0 on cat([skip_proj, up]) (unet_parts.py:73): two K segments
        TRY(pack_conv3(P->up_c1[i].w, base + L.up_c1[i].w, Ct, Ct, C, 0, Ct, L.up_c1[i].KU, 0, 9, st));
        TRY(pack_conv3(P->up_c1[i].w, base + L.up_c1[i].w, Ct, Ct, C, Ct, Ct, L.up_c1[i].KU, 9 * Ct / 16, 9, st));
        TRY(pack_conv3(P->up_c1[i].w, base + L.up_c1[i].w16, Ct, Ct, C, 0, Ct, L.up_c1[i].KU, 0, 9, st, 1));
        TRY(pack_conv3(P->up_c1[i].w, base + L.up_c1[i].w16, Ct, Ct, C, Ct, Ct, L.up_c1[i].KU, 9 * Ct / 16, 9, st, 1));
        TRY(launch_fold_bn(nullptr, P->up_c1[i].bn, base + L.up_c1[i].scale, base + L.up_c1[i].shift, Ct, Ct, 1, 0,
                           st));
        TRY(pack_conv3(P->up_c2[i].w, base + L.up_c2[i].w, Ct, Ct, Ct, 0, Ct, L.up_c2[i].KU, 0, 9, st));
        TRY(pack_conv3(P->up_c2[i].w, base + L.up_c2[i].w16, Ct, Ct, Ct, 0, Ct, L.up_c2[i].KU, 0, 9, st, 1));
        TRY(launch_fold_bn(nullptr, P->up_c2[i].bn, base + L.up_c2[i].scale, base + L.up_c2[i].shift, Ct, Ct, 1, 0,
                           st));
    }
    TRY(pack_linear(P->outc.w, base + L.outc.w, 3, 16, 32, 32, 0, st));
    TRY(pack_linear(P->outc.w, base + L.outc.w16, 3, 16, 32, 32, 0, st, 1));
    TRY(launch_fold_bn(P->outc.b, nullptr, base + L.outc.scale, base + L.outc.shift, 3, 16, 1, 0, st));
    hipError_t e = hipMemcpyAsync(base + L.emds, P->emds, (size_t)P->n_slices * 128 * sizeof(float),
                                  hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
        s3d_set_error("unet_pack: memcpy failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

#define S3D_SPLITK_FLOATS ((size_t)4718592)
struct UNetWs {
    size_t in16, a, b, x[5], p[4], proj, up, mid, splitk;
    size_t total;
};
static UNetWs unet_ws(int B, int S, int ns) {
    UNetWs W;
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t o = off;
        off += (n + 63) / 64 * 64;
        return o;
    };
    const size_t px = (size_t)B * S * S;
    W.in16 = take(px * 16);
    W.a = take(px * 64);   // ping-pong buffers for the non-tap encoder activations (largest: 64 ch @ S)
    W.b = take(px * 64 / 2);
    const int xc[5] = {64, 128, 256, 512, 512};
    for (int i = 0; i < 5; ++i) W.x[i] = take((px >> (2 * i)) * xc[i]);
    for (int i = 0; i < 4; ++i) W.p[i] = take((px >> (2 * (i + 1))) * xc[i]);
    W.proj = take(px * 32);                 // largest skip projection: (B,S,S,32)
    W.up = take(px * ns * 32);              // largest ConvT output: (B*ns,S,S,32)
    W.mid = take(px * ns * 32);
    W.splitk = take(S3D_SPLITK_FLOATS);   // split-K partials of the few-pixel encoder layers
    W.total = off;
    return W;
}

extern "C" size_t s3d_unet_workspace_bytes(int batch, int size, int n_slices) {
    return unet_ws(batch, size, n_slices).total * sizeof(float);
}

static int g_desc_prec = S3D_PREC_F32;   // set by the entry point that builds descriptors (host, single stream)
static ConvLaunch conv_desc(const float* base, const PackedConv& pc, int N, int H, int W, int ks, int act) {
    ConvLaunch c = {};
    c.wpk16 = g_desc_prec != S3D_PREC_F32 ? (const void*)(base + pc.w16) : nullptr;
    c.single_pass = g_desc_prec == S3D_PREC_F16;
    c.N = N; c.H = H; c.W = W; c.ks = ks;
    c.CoutPad = pc.cout_pad; c.wpk = base + pc.w; c.KU = pc.KU;
    c.scale = base + pc.scale; c.shift = base + pc.shift;
    c.act = act; c.out_mode = S3D_OUT_NHWC; c.cout_store = pc.cout_pad; c.out_cstride = pc.cout_pad;
    return c;
}
static ConvSrc plain_src(const float* p, int C) { return ConvSrc{p, C, 1, 0, 0}; }

extern "C" int s3d_unet_encode_fwd(const void* packed, const float* img, const S3dPyramid* out,
                                   float* slices_rec, int B, int S, int ns, int prec, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    S3D_CHECK_ARG(prec == S3D_PREC_F32 || prec == S3D_PREC_F16X3 || prec == S3D_PREC_F16 || prec == S3D_PREC_BF16,
                  "unet_encode: precision mode %d", prec);
    if (prec == S3D_PREC_BF16) prec = S3D_PREC_F16;   // the bf16 mode covers the decoder's attention / FFN GEMMs; the conv engine runs single-pass f16
    struct PrecScope { PrecScope(int p) { g_desc_prec = p; } ~PrecScope() { g_desc_prec = S3D_PREC_F32; } } ps_(prec);
    S3D_CHECK_ARG(packed && img && out && workspace, "unet_encode: null argument");
    S3D_CHECK_ARG(B >= 1 && S >= 16 && S % 16 == 0, "unet_encode: B=%d S=%d (S must be a multiple of 16)", B, S);
    S3D_CHECK_ARG(ns >= 1 && ns <= 12, "unet_encode: n_slices %d", ns);
    S3D_CHECK_ARG(out->n_img == B * ns && out->size == S, "unet_encode: pyramid handle mismatch");
    const UNetLayout L = unet_layout(ns);
    const UNetWs W = unet_ws(B, S, ns);
    if (workspace_bytes < W.total * sizeof(float)) {
        s3d_set_error("unet_encode: workspace %zu < %zu bytes", workspace_bytes, W.total * sizeof(float));
        return S3D_E_WORKSPACE;
    }
    const float* base = (const float*)packed;
    float* ws = (float*)workspace;

    ProfScope prof_(S3D_PROF_UNET, st);
    // ---- VGG16-BN encoder (unet_custom.py:43-47) ----
    const float* cur = ws + W.in16;
    int curC = 16, res = S, tap_i = 0;
    float* pp[2] = {ws + W.a, ws + W.b};
    int flip = 0;
    for (int i = 0; i < 13; ++i) {
        if (i == 0) {   // 3 -> 64 straight from the NCHW image: a direct fp32 kernel bound by its 64-channel write
                        // (the padded implicit GEMM spent 100 us on 0.9 GFLOP); folded BN + ReLU in its epilogue
            TRY(launch_conv3x3_first(img, 3, base + L.enc0_raw, nullptr, pp[flip], B, S, S, st, base + L.enc[0].scale,
                                     base + L.enc[0].shift, 1));
            cur = pp[flip];
            curC = kEncCout[0];
            flip ^= 1;
            continue;
        }
        ConvLaunch c = conv_desc(base, L.enc[i], B, res, res, 3, kEncTap[i] ? S3D_ACT_NONE : S3D_ACT_RELU);
        c.nsrc = 1;
        c.src[0] = plain_src(cur, curC);
        c.splitk_ws = ws + W.splitk; c.splitk_floats = S3D_SPLITK_FLOATS;
        float* dst = kEncTap[i] ? ws + W.x[tap_i] : pp[flip];
        c.out = dst;
        TRY(launch_conv(c, st));
        cur = dst;
        curC = kEncCout[i];
        if (!kEncTap[i]) flip ^= 1;
        if (kEncTap[i]) {
            if (tap_i < 4) {  // BN + ReLU + MaxPool open the next block
                TRY(launch_bn_relu_pool(ws + W.x[tap_i], base + L.pool_scale[tap_i], base + L.pool_shift[tap_i],
                                        ws + W.p[tap_i], B, res, res, curC, st));
                cur = ws + W.p[tap_i];
                res /= 2;
            }
            ++tap_i;
        }
    }
    // ---- latent = trans_c(cat[tile(x5), emb])  (unet_custom.py:50-58) ----
    const int r5 = S / 16;
    {
        ConvLaunch c = conv_desc(base, L.trans_c, B * ns, r5, r5, 1, S3D_ACT_NONE);
        c.nsrc = 2;
        c.src[0] = ConvSrc{ws + W.x[4], 512, ns, 0, 0};
        c.src[1] = ConvSrc{base + L.emds, 128, 1, ns, 1};
        c.out = out->level[0];
        TRY(launch_conv(c, st));
    }
    // ---- up1..up4 (unet_custom.py:60-67, unet_parts.py:55-75) ----
    const float* prev = out->level[0];
    int rp = r5;
    for (int i = 0; i < 4; ++i) {
        const int C = kUpC[i], Ct = C / 2, ro = rp * 2;
        {  // skip projection, once per image: trans_up(expand_bs(x)) == expand_bs(trans_up(x))
            ConvLaunch c = conv_desc(base, L.trans_up[i], B, ro, ro, 1, S3D_ACT_NONE);
            c.nsrc = 1;
            c.src[0] = plain_src(ws + W.x[3 - i], C);
            c.out = ws + W.proj;
            TRY(launch_conv(c, st));
        }
        {  // ConvTranspose2d 2x2 s2 as a 1x1 GEMM with N = 4*Ct and a quadrant-scatter store
            ConvLaunch c = conv_desc(base, L.up_t[i], B * ns, rp, rp, 1, S3D_ACT_NONE);
            c.nsrc = 1;
            c.src[0] = plain_src(prev, C);
            c.out = ws + W.up;
            c.out_mode = S3D_OUT_CONVT;
            c.cout_store = Ct;
            TRY(launch_conv(c, st));
        }
        {
            ConvLaunch c = conv_desc(base, L.up_c1[i], B * ns, ro, ro, 3, S3D_ACT_RELU);
            c.nsrc = 2;
            c.src[0] = ConvSrc{ws + W.proj, Ct, ns, 0, 0};
            c.src[1] = plain_src(ws + W.up, Ct);
            c.out = ws + W.mid;
            TRY(launch_conv(c, st));
        }
        {
            ConvLaunch c = conv_desc(base, L.up_c2[i], B * ns, ro, ro, 3, S3D_ACT_RELU);
            c.nsrc = 1;
            c.src[0] = plain_src(ws + W.mid, Ct);
            c.out = out->level[i + 1];
            TRY(launch_conv(c, st));
        }
        prev = out->level[i + 1];
        rp = ro;
    }
    if (slices_rec) {  // OutConv 1x1 + tanh -> NCHW (unet_parts.py:78-84)
        ConvLaunch c = conv_desc(base, L.outc, B * ns, S, S, 1, S3D_ACT_TANH);
        c.nsrc = 1;
        c.src[0] = plain_src(prev, 32);
        c.out = slices_rec;
        c.out_mode = S3D_OUT_NCHW;
        c.cout_store = 3;
        TRY(launch_conv(c, st));
    }
    return 0;
}

// =============================================================================================
// VGG19 perceptual loss
// =============================================================================================
static const int kVggCin[14] = {3, 64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512};
static const int kVggCout[14] = {64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512};
static const int kVggPoolAfter[14] = {0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0};  // after convs 2,7,16,25
static const int kVggTap[14] = {0, 1, 0, 2, 0, 3, 0, 0, 0, 4, 0, 0, 0, 5};        // convs 2,7,12,21,30
static const float kVggW[5] = {1.0f / 2.6f, 1.0f / 4.8f, 1.0f / 3.7f, 1.0f / 5.6f, 10.0f / 1.5f};

struct VggLayout {
    PackedConv conv[14];
    size_t mean, stdv, total;
};
static VggLayout vgg_layout() {
    VggLayout L;
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t o = off;
        off += (n + 3) / 4 * 4;
        return o;
    };
    for (int i = 0; i < 14; ++i) {
        L.conv[i].cout_pad = kVggCout[i];
        L.conv[i].KU = 9 * pad16(kVggCin[i]) / 16;
        L.conv[i].w = take((size_t)kVggCout[i] * L.conv[i].KU * 16);
        L.conv[i].w16 = 0;
        L.conv[i].scale = take(kVggCout[i]);
        L.conv[i].shift = take(kVggCout[i]);
    }
    L.mean = take(4);
    L.stdv = take(4);
    L.total = off;
    return L;
}

extern "C" size_t s3d_vgg_packed_bytes(void) { return vgg_layout().total * sizeof(float); }

extern "C" int s3d_vgg_pack(const S3dVggParams* P, void* packed, size_t packed_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    S3D_CHECK_ARG(P && packed, "vgg_pack: null argument");
    const VggLayout L = vgg_layout();
    if (packed_bytes < L.total * sizeof(float)) {
        s3d_set_error("vgg_pack: packed buffer %zu < %zu bytes", packed_bytes, L.total * sizeof(float));
        return S3D_E_WORKSPACE;
    }
    float* base = (float*)packed;
    for (int i = 0; i < 14; ++i) {
        TRY(pack_conv3(P->conv[i].w, base + L.conv[i].w, kVggCout[i], kVggCout[i], kVggCin[i], 0, kVggCin[i],
                       L.conv[i].KU, 0, 9, st));
        TRY(launch_fold_bn(P->conv[i].b, nullptr, base + L.conv[i].scale, base + L.conv[i].shift, kVggCout[i],
                           kVggCout[i], 1, 0, st));
    }
    if (hipMemcpyAsync(base + L.mean, P->mean, 3 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(base + L.stdv, P->std, 3 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        s3d_set_error("vgg_pack: memcpy failed");
        return (int)hipErrorUnknown;
    }
    return 0;
}

struct VggWs {
    size_t in16, a, b, partial, total;
};
static VggWs vgg_ws(int n_img, int S) {
    VggWs W;
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t o = off;
        off += (n + 63) / 64 * 64;
        return o;
    };
    const size_t px = (size_t)2 * n_img * S * S;
    W.in16 = take(px * 16);
    W.a = take(px * 64);
    W.b = take(px * 64);
    W.partial = take(4096);
    W.total = off;
    return W;
}
extern "C" size_t s3d_vgg_workspace_bytes(int n_img, int size) { return vgg_ws(n_img, size).total * sizeof(float); }

extern "C" int s3d_vgg_loss_fwd(const void* packed, const float* pred, const float* target, int n_img, int S,
                                float* loss_out, void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    S3D_CHECK_ARG(packed && pred && target && loss_out && workspace, "vgg_loss: null argument");
    S3D_CHECK_ARG(n_img >= 1 && S >= 16 && S % 16 == 0, "vgg_loss: n_img=%d S=%d", n_img, S);
    const VggLayout L = vgg_layout();
    const VggWs W = vgg_ws(n_img, S);
    if (workspace_bytes < W.total * sizeof(float)) {
        s3d_set_error("vgg_loss: workspace %zu < %zu bytes", workspace_bytes, W.total * sizeof(float));
        return S3D_E_WORKSPACE;
    }
    ProfScope prof_(S3D_PROF_VGG, st);
    const float* base = (const float*)packed;
    float* ws = (float*)workspace;
    if (hipMemsetAsync(loss_out, 0, sizeof(float), st) != hipSuccess) return (int)hipErrorUnknown;
    TRY(launch_vgg_prep(pred, target, base + L.mean, base + L.stdv, ws + W.in16, n_img, S, st));
    const int N2 = 2 * n_img;
    const float* cur = ws + W.in16;
    int curC = 16, res = S, flip = 0;
    float* pp[2] = {ws + W.a, ws + W.b};
    for (int i = 0; i < 14; ++i) {
        ConvLaunch c = conv_desc(base, L.conv[i], N2, res, res, 3, i == 13 ? S3D_ACT_NONE : S3D_ACT_RELU);
        c.nsrc = 1;
        c.src[0] = plain_src(cur, curC);
        c.out = pp[flip];
        TRY(launch_conv(c, st));
        cur = pp[flip];
        curC = kVggCout[i];
        flip ^= 1;
        if (kVggTap[i]) {
            const long half = (long)n_img * res * res * curC;
            TRY(launch_l1_diff(cur, cur + half, half, 0.001f * kVggW[kVggTap[i] - 1] / (float)half, ws + W.partial,
                               loss_out, st));
        }
        if (kVggPoolAfter[i]) {
            TRY(launch_bn_relu_pool(cur, nullptr, nullptr, pp[flip], N2, res, res, curC, st));
            cur = pp[flip];
            flip ^= 1;
            res /= 2;
        }
    }
    return 0;
}

// =============================================================================================
// head
// =============================================================================================
HeadLayout head_layout() {
    HeadLayout H;
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t o = off;
        off += (n + 3) / 4 * 4;
        return o;
    };
    H.fcp_w = take(128 * 3);
    H.fcp_b = take(128);
    H.fcs_b = take(128);
    const int lc[3] = {512, 256, 128};
    for (int l = 0; l < 3; ++l) H.wproj[l] = take((size_t)128 * lc[l]);
    for (int l = 0; l < 3; ++l) H.wproj16[l] = take((size_t)128 * lc[l]);
    H.ws34 = take(128 * 96);
    H.ws34_16 = take(128 * 96);
    for (int l = 0; l < S3D_N_LAYERS; ++l) {
        H.L[l].inw = take(384 * 128);
        H.L[l].inb = take(384);
        H.L[l].outw = take(128 * 128);
        H.L[l].outb = take(128);
        H.L[l].ln1g = take(128);
        H.L[l].ln1b = take(128);
        H.L[l].w1 = take((size_t)S3D_FFN * 128);
        H.L[l].b1 = take(S3D_FFN);
        H.L[l].w2 = take((size_t)128 * S3D_FFN);
        H.L[l].b2 = take(128);
        H.L[l].ln2g = take(128);
        H.L[l].ln2b = take(128);
        H.L[l].wf16 = take((size_t)S3D_FFN_NCHUNK * 8192);
        H.L[l].aq16 = take((size_t)(96 + 32) * 512);
        H.L[l].wfb16 = take((size_t)S3D_FFN_NCHUNK * 8192);      // bf16 twins of the two images (S3D_PREC_BF16)
        H.L[l].aqb16 = take((size_t)(96 + 32) * 512);
    }
    H.fco_w = take(128);
    H.fco_b = take(4);
    H.last.wm = take(512 * 128); H.last.wm16 = take(512 * 128); H.last.bm = take(512);
    H.last.wn = take(128 * 512); H.last.wn16 = take(128 * 512); H.last.bn = take(128);
    H.last.dense = take(512 * 128);
    H.total = off;
    return H;
}

extern "C" size_t s3d_head_packed_bytes(void) { return head_layout().total * sizeof(float); }

static int copy_vec(float* dst, const float* src, size_t n, hipStream_t st) {
    hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
        s3d_set_error("head_pack: memcpy failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// bf16_twins: also pack the bf16 images of the FFN / attention fragments (S3D_PREC_BF16, an inference mode of
// Slices3DRegModel).  The GT head (api_gt.inc rejects S3D_PREC_BF16) and the trainer's head (repacked with the weights every
// step, never decoded in bf16) skip them: 6.9 MB and 6 pack launches per head.
static int pack_head_layers(const S3dLayerParams* layers, float* b, const HeadLayout& H, hipStream_t st, bool bf16_twins) {
    for (int l = 0; l < S3D_N_LAYERS; ++l) {
        const S3dLayerParams& p = layers[l];
        TRY(pack_linear(p.in_proj_w, b + H.L[l].inw, 384, 384, 128, 128, 0, st));
        TRY(copy_vec(b + H.L[l].inb, p.in_proj_b, 384, st));
        TRY(pack_linear(p.out_proj_w, b + H.L[l].outw, 128, 128, 128, 128, 0, st));
        TRY(copy_vec(b + H.L[l].outb, p.out_proj_b, 128, st));
        TRY(copy_vec(b + H.L[l].ln1g, p.norm1_w, 128, st));
        TRY(copy_vec(b + H.L[l].ln1b, p.norm1_b, 128, st));
        TRY(pack_linear(p.lin1_w, b + H.L[l].w1, S3D_FFN, S3D_FFN, 128, 128, 0, st));
        TRY(copy_vec(b + H.L[l].b1, p.lin1_b, S3D_FFN, st));
        TRY(pack_linear(p.lin2_w, b + H.L[l].w2, 128, 128, S3D_FFN, S3D_FFN, S3D_FFN_CHUNK / 16, st));
        TRY(copy_vec(b + H.L[l].b2, p.lin2_b, 128, st));
        TRY(copy_vec(b + H.L[l].ln2g, p.norm2_w, 128, st));
        TRY(copy_vec(b + H.L[l].ln2b, p.norm2_b, 128, st));
        TRY(launch_pack_ffn_f16x3(p.lin1_w, p.lin2_w, b + H.L[l].wf16, st));
        TRY(launch_pack_attn_q_f16x3(p.in_proj_w, p.out_proj_w, b + H.L[l].aq16, st));
        if (bf16_twins) {
            TRY(launch_pack_ffn_f16x3(p.lin1_w, p.lin2_w, b + H.L[l].wfb16, st, 1));
            TRY(launch_pack_attn_q_f16x3(p.in_proj_w, p.out_proj_w, b + H.L[l].aqb16, st, 1));
        }
    }
    {   // absorbed token-0 attention of the last layer (launch_attn_last_mix)
        PackBatchSuspend now;   // `dense` is reused: these packs must run between the two absorb launches
        const S3dLayerParams& p = layers[S3D_N_LAYERS - 1];
        float* dense = b + H.last.dense;
        TRY(launch_absorb_last(p.in_proj_w, p.in_proj_b, p.out_proj_w, p.out_proj_b, dense, b + H.last.bm, 0, st));
        TRY(pack_linear(dense, b + H.last.wm, 512, 512, 128, 128, 0, st));
        TRY(pack_linear(dense, b + H.last.wm16, 512, 512, 128, 128, 0, st, 1));
        TRY(launch_absorb_last(p.in_proj_w, p.in_proj_b, p.out_proj_w, p.out_proj_b, dense, b + H.last.bn, 1, st));
        TRY(pack_linear(dense, b + H.last.wn, 128, 128, 512, 512, 0, st));
        TRY(pack_linear(dense, b + H.last.wn16, 128, 128, 512, 512, 0, st, 1));
    }
    return 0;
}

static int head_pack_impl(const S3dHeadParams* P, void* packed, size_t packed_bytes, void* stream, bool bf16_twins) {
    hipStream_t st = (hipStream_t)stream;
    S3D_CHECK_ARG(P && packed, "head_pack: null argument");
    const HeadLayout H = head_layout();
    if (packed_bytes < H.total * sizeof(float)) {
        s3d_set_error("head_pack: packed buffer %zu < %zu bytes", packed_bytes, H.total * sizeof(float));
        return S3D_E_WORKSPACE;
    }
    float* b = (float*)packed;
    TRY(copy_vec(b + H.fcp_w, P->fc_p_w, 128 * 3, st));
    TRY(copy_vec(b + H.fcp_b, P->fc_p_b, 128, st));
    TRY(copy_vec(b + H.fcs_b, P->fc_s_b, 128, st));
    const int lc[3] = {512, 256, 128}, lo[3] = {0, 512, 768};
    for (int l = 0; l < 3; ++l) TRY(pack_linear(P->fc_s_w + lo[l], b + H.wproj[l], 128, 128, lc[l], 992, 0, st));
    for (int l = 0; l < 3; ++l) TRY(pack_linear(P->fc_s_w + lo[l], b + H.wproj16[l], 128, 128, lc[l], 992, 0, st, 1));
    TRY(pack_linear(P->fc_s_w + 896, b + H.ws34, 128, 128, 96, 992, 0, st));
    TRY(pack_linear(P->fc_s_w + 896, b + H.ws34_16, 128, 128, 96, 992, 0, st, 1));
    TRY(pack_head_layers(P->layer, b, H, st, bf16_twins));
    TRY(copy_vec(b + H.fco_w, P->fc_out_w, 128, st));
    TRY(copy_vec(b + H.fco_b, P->fc_out_b, 1, st));
    return 0;
}
extern "C" int s3d_head_pack(const S3dHeadParams* P, void* packed, size_t packed_bytes, void* stream) {
    return head_pack_impl(P, packed, packed_bytes, stream, true);
}

extern "C" int s3d_latent_build(const void* head_packed, const S3dPyramid* pyr, const S3dLatent* out, int prec,
                                void* stream) {
    hipStream_t st = (hipStream_t)stream;
    S3D_CHECK_ARG(head_packed && pyr && out, "latent_build: null argument");
    S3D_CHECK_ARG(out->n_img == pyr->n_img && out->size == pyr->size, "latent_build: handle mismatch");
    S3D_CHECK_ARG(out->fine[0] == pyr->level[3] && out->fine[1] == pyr->level[4],
                  "latent_build: latent.fine must alias pyramid levels 3,4");
    const HeadLayout H = head_layout();
    const float* b = (const float*)head_packed;
    if (prec == S3D_PREC_BF16) prec = S3D_PREC_F16;   // (see s3d_unet_encode_fwd)
    ProfScope prof_(S3D_PROF_LATENT, st);
    const int lc[3] = {512, 256, 128};
    for (int l = 0; l < 3; ++l) {
        const int r = (pyr->size / 16) << l;
        ConvLaunch c = {};
        c.N = pyr->n_img; c.H = r; c.W = r; c.ks = 1;
        c.CoutPad = 128; c.wpk = b + H.wproj[l]; c.KU = lc[l] / 16;
        c.wpk16 = prec != S3D_PREC_F32 ? (const void*)(b + H.wproj16[l]) : nullptr;
        c.single_pass = prec == S3D_PREC_F16;
        c.scale = nullptr; c.shift = nullptr;  // identity epilogue: fc_s bias is added by the sampler
        c.act = S3D_ACT_NONE; c.out_mode = S3D_OUT_NHWC; c.cout_store = 128; c.out_cstride = 128;
        c.nsrc = 1;
        c.src[0] = plain_src(pyr->level[l], lc[l]);
        c.out = out->proj[l];
        TRY(launch_conv(c, st));
    }
    return 0;
}

// =============================================================================================
// decode
// =============================================================================================
#ifndef S3D_CHUNK_GROUPS
#define S3D_CHUNK_GROUPS 32768  // 524288 queries per pass: X = 32768*13*16*128*4 B = 3.49 GB
#endif
#define S3D_CHUNK_GROUPS_MIN 4096   // smallest pass a caller's workspace may force (s3d_decode_workspace_bytes_min)
// The FFN kernel is launched on at most this many groups at a time.  One pass over the bench's 400 k queries instead of two
// saves the attention kernel, the token builder and the final layer a launch tail each (-0.23 ms per step); the FFN kernel
// itself ran 0.5 % slower per row on a 3.5 GB launch than on a 1.7 GB one (tools/chunk_groups.sh), so it keeps the old size.
#define S3D_FFN_LAUNCH_GROUPS 16384

struct DecodeWs {
    size_t X, X0, perm, sortws, last, total;
    long chunk;   // groups per pass
};
// scratch of the absorbed last-layer attention per query row: x0 | u (128 each), qt | xbar (512 each)
#define S3D_LAST_ROW_FLOATS (2 * 128 + 2 * 512)
#define S3D_SORT_MIN_QUERIES 4096   // below this the sort costs more than the locality buys
static DecodeWs decode_ws(int batch, long n_qry, int ns, long chunk_groups = S3D_CHUNK_GROUPS) {
    const long gpb = (n_qry + S3D_GROUP - 1) / S3D_GROUP;
    long g = gpb * batch;
    if (g > chunk_groups) g = chunk_groups;
    DecodeWs W;
    W.chunk = chunk_groups;
    W.X = 0;
    W.X0 = (size_t)g * (ns + 1) * S3D_GROUP * 128;
    W.perm = W.X0 + (size_t)g * S3D_GROUP * 128;
    W.sortws = W.perm + (size_t)batch * n_qry;
    W.last = (W.sortws + query_sort_ws_ints(batch, n_qry) + 63) / 64 * 64;
    W.total = W.last + (size_t)g * S3D_GROUP * S3D_LAST_ROW_FLOATS;
    return W;
}
// the pass size a workspace of `bytes` allows: the preferred one, else halved down to S3D_CHUNK_GROUPS_MIN (a decode of
// >= 524 288 queries needs 6.4 GB at the preferred size, 0.8 GB at the smallest; smaller passes cost launch tails only)
static bool decode_ws_fit(int batch, long n_qry, int ns, size_t bytes, DecodeWs& W) {
    for (long c = S3D_CHUNK_GROUPS; c >= S3D_CHUNK_GROUPS_MIN; c >>= 1) {
        W = decode_ws(batch, n_qry, ns, c);
        if (bytes >= W.total * sizeof(float)) return true;
    }
    return false;
}

extern "C" size_t s3d_decode_workspace_bytes(int batch, long n_qry, int n_slices) {
    return decode_ws(batch, n_qry, n_slices).total * sizeof(float);
}
extern "C" size_t s3d_decode_workspace_bytes_min(int batch, long n_qry, int n_slices) {
    return decode_ws(batch, n_qry, n_slices, S3D_CHUNK_GROUPS_MIN).total * sizeof(float);
}

// ---------------------------------------------------------------------------------------------
// Two decode lanes (round 5, OPT-IN: measured time-neutral, profiles/r05_lanes_ab.md — 35.51 against 35.54 ms per step;
// every matrix kernel of the decoder already runs at the socket's power cap, so filling one kernel's idle matrix cycles with
// another's only lowers the clock).  The attention kernel (matrix pipe 52 % busy, latency-bound phases) and the FFN kernel (79 %
// busy, power-bound) each hold one wave per SIMD per workgroup, two workgroups per CU: run back to back the chip sees
// attention || attention, then FFN || FFN.  With the pass split into two halves whose layer chains run on two streams, offset
// by one attention launch, a CU holds one workgroup of each kind during every attention phase but the first — the attention
// waves' VALU / LDS phases fill the FFN waves' idle matrix cycles.  The second stream and its events are created lazily per
// device and live for the process; everything is ordered behind the caller's stream (fork after the token builder, join
// before the call returns its last launch), so the caller still sees one in-order stream.  Results are bit-identical: rows
// are independent and every kernel sees the rows it saw before.
// ---------------------------------------------------------------------------------------------
// Thread safety: the side stream and its two events are per-device state shared by every caller of that device.
// Two host threads decoding on one device with different caller streams must not interleave their fork / join records (A
// records stagger, B records stagger, A waits: A's side-stream work would be ordered behind B's token builder, not its own),
// so the whole fork-to-join section of a call holds the device's mutex (the enqueue only: microseconds; the GPU work of two
// callers then simply queues on the one side stream).  Creation is under the same mutex and all-or-nothing.
struct DecodeLanes {
    std::mutex mu;
    hipStream_t aux = nullptr;
    hipEvent_t stagger = nullptr, join = nullptr;
    bool ok = false;
};
static std::atomic<int> g_decode_lanes{-1};   // -1: not configured (env S3D_DECODE_LANES, default 1)
extern "C" int s3d_decode_set_lanes(int n) {
    S3D_CHECK_ARG(n == 1 || n == 2, "decode_set_lanes: %d (1 or 2)", n);
    g_decode_lanes.store(n, std::memory_order_relaxed);
    return 0;
}
static int decode_lanes() {
    int v = g_decode_lanes.load(std::memory_order_relaxed);
    if (v < 0) {   // (two racing first callers read the same environment: both store the same value)
        const char* e = getenv("S3D_DECODE_LANES");
        v = (e && atoi(e) == 2) ? 2 : 1;
        g_decode_lanes.store(v, std::memory_order_relaxed);
    }
    return v;
}
// the current device's lanes, or nullptr when they cannot be created; the caller locks L->mu around its fork-to-join section
// and calls decode_lanes_ready(L) under that lock
static DecodeLanes* decode_lanes_for_device() {
    static DecodeLanes lanes[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    return &lanes[dev & 63];
}
static bool decode_lanes_ready(DecodeLanes& L) {   // under L.mu
    if (L.ok) return true;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool made = hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess &&
                      hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess;
    if (!made) {   // nothing half-made survives: a later call starts from scratch instead of leaking a stream per call
        if (e1) (void)hipEventDestroy(e1);
        if (e0) (void)hipEventDestroy(e0);
        if (s) (void)hipStreamDestroy(s);
        (void)hipGetLastError();
        return false;
    }
    L.aux = s; L.stagger = e0; L.join = e1; L.ok = true;
    return true;
}
#define S3D_LANES_MIN_GROUPS 8192   // below 131 072 queries per pass the halves do not fill the chip

static LayerPtrs layer_ptrs(const float* b, const HeadLayout& H, int l) {
    LayerPtrs p;
    p.inw = b + H.L[l].inw; p.inb = b + H.L[l].inb; p.outw = b + H.L[l].outw; p.outb = b + H.L[l].outb;
    p.ln1g = b + H.L[l].ln1g; p.ln1b = b + H.L[l].ln1b; p.w1 = b + H.L[l].w1; p.b1 = b + H.L[l].b1;
    p.w2 = b + H.L[l].w2; p.b2 = b + H.L[l].b2; p.ln2g = b + H.L[l].ln2g; p.ln2b = b + H.L[l].ln2b;
    p.wf16 = b + H.L[l].wf16;
    p.aq16 = b + H.L[l].aq16;
    p.wfb16 = b + H.L[l].wfb16;
    p.aqb16 = b + H.L[l].aqb16;
    return p;
}

// rows x k  ->  rows x n  linear map on the conv engine (1x1 convolution over a row "image")
static int rows_linear(const float* b, size_t w32, size_t w16, int n, int k, const float* bias, const float* x,
                       long nrows, float* out, const float* residual, int prec, hipStream_t st) {
    ConvLaunch c = {};
    c.N = 1; c.H = 1; c.W = (int)nrows; c.ks = 1;
    c.CoutPad = n; c.wpk = b + w32; c.KU = k / 16;
    c.wpk16 = prec != S3D_PREC_F32 ? (const void*)(b + w16) : nullptr;
        c.single_pass = prec == S3D_PREC_F16;
    c.shift = bias; c.act = S3D_ACT_NONE;
    c.out_mode = S3D_OUT_NHWC; c.cout_store = n; c.out_cstride = n;
    c.nsrc = 1;
    c.src[0] = plain_src(x, k);
    c.out = out; c.residual = residual;
    return launch_conv(c, st);
}

// X [gc][T][16][128] (input of the last layer) -> X0 [gc*16][128] = LN1(x0 + out_proj(attention of token 0)),
// see launch_attn_last_mix.  scratch: gc*16*S3D_LAST_ROW_FLOATS floats.
// fold_ln: X0 receives the pre-LayerNorm sums u and the caller's final FFN kernel normalises them in its prologue
// (launch_ffn_layer(..., pre_ln1 = true): the ln_fwd launch and one round trip of the token-0 rows are gone)
static std::atomic<int> g_shared_footprint{-1};   // -1: not configured (env S3D_SHARED_FOOTPRINT, default 1)
extern "C" int s3d_decode_set_shared_footprint(int on) {
    S3D_CHECK_ARG(on == 0 || on == 1, "decode_set_shared_footprint: %d (0 or 1)", on);
    g_shared_footprint.store(on, std::memory_order_relaxed);
    return 0;
}
int s3d_shared_footprint() {   // (also read by the training entry point)
    int v = g_shared_footprint.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("S3D_SHARED_FOOTPRINT");
        v = (e && atoi(e) == 0) ? 0 : 1;
        g_shared_footprint.store(v, std::memory_order_relaxed);
    }
    return v;
}
static std::atomic<int> g_last_fused{-1};   // -1: not configured (env S3D_LAST_FUSED, default 1)
extern "C" int s3d_decode_set_last_fused(int on) {
    S3D_CHECK_ARG(on == 0 || on == 1, "decode_set_last_fused: %d (0 or 1)", on);
    g_last_fused.store(on, std::memory_order_relaxed);
    return 0;
}
static bool last_fused() {
    int v = g_last_fused.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("S3D_LAST_FUSED");
        v = (e && atoi(e) == 0) ? 0 : 1;
        g_last_fused.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
static int attn_last_layer(const float* b, const HeadLayout& H, const LayerPtrs& lp, const float* X, float* X0, long gc,
                           int T, float* scratch, int prec, hipStream_t st, bool fold_ln = false) {
    if (fold_ln && prec != S3D_PREC_F32 && last_fused())   // one kernel, same bits (decode_last.hip)
        return launch_attn_last_fused(X, X0, gc, T, b + H.last.wm16, b + H.last.bm, b + H.last.wn16, b + H.last.bn,
                                      prec == S3D_PREC_F16, st);
    const long rows0 = gc * S3D_GROUP;
    float* x0 = scratch;
    float* u = x0 + rows0 * 128;
    float* qt = u + rows0 * 128;
    float* xbar = qt + rows0 * 512;
    TRY(launch_tok0_copy(const_cast<float*>(X), x0, gc, T, 0, 128, st));
    TRY(rows_linear(b, H.last.wm, H.last.wm16, 512, 128, b + H.last.bm, x0, rows0, qt, nullptr, prec, st));
    TRY(launch_attn_last_mix(X, qt, xbar, gc, T, st));
    TRY(rows_linear(b, H.last.wn, H.last.wn16, 128, 512, b + H.last.bn, xbar, rows0, fold_ln ? X0 : u, x0, prec, st));
    return fold_ln ? 0 : launch_ln_fwd(u, lp.ln1g, lp.ln1b, X0, rows0, st);
}

static int decode_impl(const void* head_packed, const S3dLatent* lat, const float* qry, const float* rot,
                       const float* trans, int flip_yz, int nx, float box, float sign, float* out, int batch,
                       long n_qry, int ns, int prec, void* workspace, size_t workspace_bytes, hipStream_t st,
                       long q_offset = 0, float* stages = nullptr) {
    S3D_CHECK_ARG(head_packed && lat && trans && out && workspace, "decode: null argument");
    S3D_CHECK_ARG(batch >= 1 && n_qry >= 1, "decode: batch=%d n_qry=%ld", batch, n_qry);
    S3D_CHECK_ARG(ns >= 1 && ns <= 12, "decode: n_slices %d", ns);
    S3D_CHECK_ARG(lat->n_img == batch * ns, "decode: latent has %d images, expected %d", lat->n_img, batch * ns);
    S3D_CHECK_ARG(prec == S3D_PREC_F32 || prec == S3D_PREC_F16X3 || prec == S3D_PREC_F16 || prec == S3D_PREC_BF16,
                  "decode: precision mode %d not built", prec);
    const bool bf16 = prec == S3D_PREC_BF16;   // attention + FFN GEMMs on the bf16 MFMA; everything else as S3D_PREC_F16
    if (bf16) prec = S3D_PREC_F16;
    const int ffn_prec = bf16 ? S3D_PREC_BF16 : prec;
    DecodeWs W;
    if (!decode_ws_fit(batch, n_qry, ns, workspace_bytes, W)) {
        s3d_set_error("decode: workspace %zu < %zu bytes (s3d_decode_workspace_bytes_min)", workspace_bytes, W.total * sizeof(float));
        return S3D_E_WORKSPACE;
    }
    const long CHUNK = W.chunk;
    const HeadLayout H = head_layout();
    const float* b = (const float*)head_packed;
    float* X = (float*)workspace + W.X;
    float* X0 = (float*)workspace + W.X0;
    const int T = ns + 1;
    const long gpb = (n_qry + S3D_GROUP - 1) / S3D_GROUP;
    const long G = gpb * batch;
    const int* perm = nullptr;
    if (qry && n_qry >= S3D_SORT_MIN_QUERIES) {   // scattered queries: sort by projected pixel for L2 locality
        int* pm = (int*)((float*)workspace + W.perm);
        TRY(launch_query_sort(qry, flip_yz ? nullptr : rot, trans, flip_yz, batch, n_qry, pm,
                              (int*)((float*)workspace + W.sortws), st));
        perm = pm;
    }
    // stage capture (s3d_decode_points_stages_fwd): one pass, queries in caller order
    S3D_CHECK_ARG(!stages || (G <= CHUNK && !perm), "decode stages: at most %d unsorted queries per object",
                  (int)S3D_SORT_MIN_QUERIES - 1);
    const size_t rows_all = (size_t)G * T * S3D_GROUP * 128, rows0_all = (size_t)G * S3D_GROUP * 128;
    for (long g0 = 0; g0 < G; g0 += CHUNK) {
        const long gc = G - g0 < CHUNK ? G - g0 : CHUNK;
        SampleArgs sa = {};
        for (int l = 0; l < 3; ++l) sa.proj[l] = lat->proj[l];
        sa.fine[0] = lat->fine[0]; sa.fine[1] = lat->fine[1];
        sa.size = lat->size; sa.n_slices = ns;
        sa.fcp_w = b + H.fcp_w; sa.fcp_b = b + H.fcp_b; sa.fcs_b = b + H.fcs_b; sa.ws34 = b + H.ws34;
        sa.ws34_16 = prec != S3D_PREC_F32 ? b + H.ws34_16 : nullptr;
        sa.qry = qry; sa.rot = rot; sa.trans = trans; sa.flip_yz = flip_yz;
        sa.n_qry = n_qry; sa.groups_per_batch = gpb; sa.g_begin = g0; sa.g_count = gc;
        sa.nx = nx; sa.box = box; sa.q_offset = q_offset; sa.X = X; sa.perm = perm;
        sa.lane_footprints = !s3d_shared_footprint();
        {
            ProfScope prof_(S3D_PROF_SAMPLE, st);
            TRY(launch_sample_tokens(sa, st));
        }
        if (stages && hipMemcpyAsync(stages, X, rows_all * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
            s3d_set_error("decode stages: memcpy failed");
            return (int)hipErrorUnknown;
        }
        // the three encoder layers + fc_out on groups [gs, gs + gn) of this pass, enqueued on stream s
        const bool fold_ln = prec != S3D_PREC_F32;   // LayerNorm1 of the last layer in the final FFN kernel's prologue
        auto run_layers = [&](hipStream_t s, long gs, long gn, hipEvent_t after_first_attn) -> int {
            float* Xs = X + (size_t)gs * T * S3D_GROUP * 128;
            float* X0s = X0 + (size_t)gs * S3D_GROUP * 128;
            float* lasts = (float*)workspace + W.last + (size_t)gs * S3D_GROUP * S3D_LAST_ROW_FLOATS;
            for (int l = 0; l < S3D_N_LAYERS; ++l) {
                const LayerPtrs lp = layer_ptrs(b, H, l);
                const bool last = l == S3D_N_LAYERS - 1;
                {
                    ProfScope prof_(S3D_PROF_ATTN, s);
                    if (last)          // only token 0 of the last layer is consumed (models.py:83): absorbed form, every mode
                        TRY(attn_last_layer(b, H, lp, Xs, X0s, gn, T, lasts, prec, s, fold_ln));
                    else if (prec != S3D_PREC_F32)
                        TRY(launch_attn_layer_q(Xs, gn, T, lp, s, prec == S3D_PREC_F16, bf16));
                    else
                        TRY(launch_attn_layer(Xs, nullptr, gn, T, lp, s));
                }
                if (l == 0 && after_first_attn && hipEventRecord(after_first_attn, s) != hipSuccess) {
                    s3d_set_error("decode: hipEventRecord failed");
                    return (int)hipErrorUnknown;
                }
                if (!last) {
                    for (long f0 = 0; f0 < gn; f0 += S3D_FFN_LAUNCH_GROUPS) {   // (rows are independent: any split is the same result)
                        const long fc = gn - f0 < S3D_FFN_LAUNCH_GROUPS ? gn - f0 : S3D_FFN_LAUNCH_GROUPS;
                        ProfScope prof_(S3D_PROF_FFN, s);
                        TRY(launch_ffn_layer(Xs + (size_t)f0 * T * S3D_GROUP * 128, fc * T * S3D_GROUP, lp, nullptr, nullptr, nullptr,
                                             1.f, gpb, n_qry, g0 + gs + f0, ffn_prec, nullptr, s));
                    }
                    if (stages) TRY(launch_tok0_copy(Xs, stages + rows_all + (size_t)l * rows0_all, gn, T, 0, 128, s));
                } else {
                    ProfScope prof_(S3D_PROF_FFN_FINAL, s);
                    if (stages) {   // the final kernel keeps the layer's output rows in registers (LayerNorm -> fc_out): the capture
                                    // runs the full-row form of the same kernel on a copy of the token-0 rows
                        float* d = stages + rows_all + (size_t)l * rows0_all;
                        if (fold_ln) {   // X0 holds the pre-LayerNorm sums (the final kernel normalises them itself): the capture's copy is normalised here
                            TRY(launch_ln_fwd(X0s, lp.ln1g, lp.ln1b, d, gn * S3D_GROUP, s));
                        } else if (hipMemcpyAsync(d, X0s, rows0_all * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
                            s3d_set_error("decode stages: memcpy failed");
                            return (int)hipErrorUnknown;
                        }
                        TRY(launch_ffn_layer(d, gn * S3D_GROUP, lp, nullptr, nullptr, nullptr, 1.f, gpb, n_qry, g0 + gs, ffn_prec,
                                             nullptr, s));
                    }
                    TRY(launch_ffn_layer(X0s, gn * S3D_GROUP, lp, b + H.fco_w, b + H.fco_b, out, sign, gpb, n_qry, g0 + gs,
                                         ffn_prec, perm, s, fold_ln));
                }
            }
            return 0;
        };
        DecodeLanes* lanes = (!stages && prec != S3D_PREC_F32 && gc >= S3D_LANES_MIN_GROUPS && decode_lanes() == 2)
                                 ? decode_lanes_for_device() : nullptr;
        std::unique_lock<std::mutex> lanes_lock;
        if (lanes) {
            lanes_lock = std::unique_lock<std::mutex>(lanes->mu);
            if (!decode_lanes_ready(*lanes)) {   // no side stream on this device: the one-stream form, same results
                lanes_lock.unlock();
                lanes = nullptr;
            }
        }
        if (!lanes) {
            TRY(run_layers(st, 0, gc, nullptr));
        } else {
            // lane 0 = the first half on the caller's stream; lane 1 = the second half on the side stream, released when lane
            // 0's first attention launch is done (and with it the token builder): from then on one lane's attention phases
            // meet the other lane's FFN phases.  The device's mutex is held from the fork to the join (see DecodeLanes).
            const long h0 = (gc / 2 + 7) / 8 * 8;
            TRY(run_layers(st, 0, h0, lanes->stagger));
            if (hipStreamWaitEvent(lanes->aux, lanes->stagger, 0) != hipSuccess) {
                s3d_set_error("decode: hipStreamWaitEvent failed");
                return (int)hipErrorUnknown;
            }
            const int rc = run_layers(lanes->aux, h0, gc - h0, nullptr);
            // join even on failure: the caller's stream must not run ahead of work already enqueued on the side stream
            if (hipEventRecord(lanes->join, lanes->aux) != hipSuccess || hipStreamWaitEvent(st, lanes->join, 0) != hipSuccess) {
                s3d_set_error("decode: lane join failed");
                return (int)hipErrorUnknown;
            }
            if (rc) return rc;
        }
    }
    return 0;
}

extern "C" int s3d_decode_points_fwd(const void* head_packed, const S3dLatent* latent, const float* qry,
                                     const float* rot, const float* trans, int flip_yz, float* sdf_out,
                                     int batch, long n_qry, int n_slices, int prec, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    S3D_CHECK_ARG(qry, "decode_points: qry is NULL");
    return decode_impl(head_packed, latent, qry, rot, trans, flip_yz, 0, 1.f, 1.f, sdf_out, batch, n_qry,
                       n_slices, prec, workspace, workspace_bytes, (hipStream_t)stream);
}

// s3d_decode_points_fwd that also hands out the intermediate rows the reference's stage goldens probe (models.py:79-83):
// stages = [G][T][16][128] token tensor after fc_p / fc_s (token 0 = fc_p(qry_rot), token 1 + s = fc_s of slice s; row of
// query q of object b: group b * ceil(Q/16) + q / 16, lane q % 16), then [3][G*16][128]: token 0 after each encoder layer.
// Debug / test entry point: one decode pass, queries in caller order (Q < 4096 per object).
extern "C" size_t s3d_decode_stages_floats(int batch, long n_qry, int n_slices) {
    const size_t G = (size_t)((n_qry + S3D_GROUP - 1) / S3D_GROUP) * batch;
    return G * S3D_GROUP * 128 * (size_t)(n_slices + 1 + S3D_N_LAYERS);
}
extern "C" int s3d_decode_points_stages_fwd(const void* head_packed, const S3dLatent* latent, const float* qry,
                                            const float* rot, const float* trans, int flip_yz, float* sdf_out,
                                            float* stages, int batch, long n_qry, int n_slices, int prec, void* workspace,
                                            size_t workspace_bytes, void* stream) {
    S3D_CHECK_ARG(qry && stages, "decode_points_stages: null argument");
    return decode_impl(head_packed, latent, qry, rot, trans, flip_yz, 0, 1.f, 1.f, sdf_out, batch, n_qry, n_slices, prec,
                       workspace, workspace_bytes, (hipStream_t)stream, 0, stages);
}

extern "C" int s3d_decode_grid_fwd(const void* head_packed, const S3dLatent* latent, const float* trans, int nx,
                                   float box, float* logits_out, int n_slices, int prec, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    S3D_CHECK_ARG(nx >= 2 && nx <= 1024, "decode_grid: nx %d", nx);
    const long n = (long)nx * nx * nx;
    // mode='test' prologue (reconstruct.py:336 builds the model with mode='test'); logits = -sdf
    return decode_impl(head_packed, latent, nullptr, nullptr, trans, 1, nx, box, -1.f, logits_out, 1, n, n_slices,
                       prec, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int s3d_decode_grid_slab_fwd(const void* head_packed, const S3dLatent* latent, const float* trans, int nx,
                                        float box, long q_begin, long q_count, float* logits_out, int n_slices,
                                        int prec, void* workspace, size_t workspace_bytes, void* stream) {
    S3D_CHECK_ARG(nx >= 2 && nx <= 1024, "decode_grid_slab: nx %d", nx);
    const long n = (long)nx * nx * nx;
    S3D_CHECK_ARG(q_begin >= 0 && q_count >= 1 && q_begin + q_count <= n,
                  "decode_grid_slab: [%ld, %ld) outside the %ld-point grid", q_begin, q_begin + q_count, n);
    return decode_impl(head_packed, latent, nullptr, nullptr, trans, 1, nx, box, -1.f, logits_out, 1, q_count,
                       n_slices, prec, workspace, workspace_bytes, (hipStream_t)stream, q_begin);
}

// =============================================================================================
// stand-alone ops
// =============================================================================================
extern "C" int s3d_project_coord_fwd(const float* coords, const float* trans, float* out, int batch, long n_qry,
                                     void* stream) {
    S3D_CHECK_ARG(coords && trans && out, "project_coord: null argument");
    return launch_project_coord(coords, trans, out, batch, n_qry, (hipStream_t)stream);
}

extern "C" size_t s3d_query_sort_workspace_bytes(int batch, long n_qry) {
    return query_sort_ws_ints(batch, n_qry) * sizeof(int);
}

extern "C" int s3d_query_sort(const float* qry, const float* rot, const float* trans, int flip_yz, int batch,
                              long n_qry, int* perm_out, void* workspace, size_t workspace_bytes, void* stream) {
    S3D_CHECK_ARG(qry && trans && perm_out && workspace, "query_sort: null argument");
    S3D_CHECK_ARG(batch >= 1 && n_qry >= 1, "query_sort: bad dims");
    if (workspace_bytes < s3d_query_sort_workspace_bytes(batch, n_qry)) {
        s3d_set_error("query_sort: workspace %zu < %zu bytes", workspace_bytes,
                      s3d_query_sort_workspace_bytes(batch, n_qry));
        return S3D_E_WORKSPACE;
    }
    return launch_query_sort(qry, rot, trans, flip_yz, batch, n_qry, perm_out, (int*)workspace, (hipStream_t)stream);
}

extern "C" size_t s3d_sample_pyramid_workspace_bytes(int batch, long n_qry) {
    // point records (float4) | permutation | sort scratch
    return (size_t)batch * n_qry * 16 + ((size_t)batch * n_qry + 4 + query_sort_ws_ints(batch, n_qry)) * sizeof(int);
}

extern "C" int s3d_sample_pyramid_fwd(const S3dPyramid* pyr, const float* grid, float* out, int batch, int n_slices,
                                      long n_qry, void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    S3D_CHECK_ARG(pyr && grid && out && workspace, "sample_pyramid: null argument");
    S3D_CHECK_ARG(batch >= 1 && n_slices >= 1 && n_qry >= 1 && pyr->n_img == batch * n_slices,
                  "sample_pyramid: pyramid holds %d images, expected %d x %d", pyr ? pyr->n_img : 0, batch, n_slices);
    S3D_CHECK_ARG(pyr->size % 16 == 0 && pyr->size >= 16, "sample_pyramid: size %d", pyr->size);
    if (workspace_bytes < s3d_sample_pyramid_workspace_bytes(batch, n_qry)) {
        s3d_set_error("sample_pyramid: workspace %zu < %zu bytes", workspace_bytes,
                      s3d_sample_pyramid_workspace_bytes(batch, n_qry));
        return S3D_E_WORKSPACE;
    }
    float* pts = (float*)workspace;
    const int* perm = nullptr;
    if (n_qry >= S3D_SORT_MIN_QUERIES) {   // visit the points in image-space locality order
        int* pm = (int*)(pts + (size_t)batch * n_qry * 4);
        TRY(launch_query_sort(grid, nullptr, nullptr, 0, batch, n_qry, pm, pm + (size_t)batch * n_qry + 4, st));
        perm = pm;
    }
    TRY(launch_sample_pyramid_points(grid, perm, pts, batch, n_qry, st));
    ProfScope prof_(S3D_PROF_SAMPLE_PYR, st);
    return launch_sample_pyramid(pyr->level, grid, perm, pts, out, batch, n_slices, pyr->size, n_qry, st);
}

extern "C" int s3d_sample_planes_fwd(const float* plane, const float* grid, float* out, int n, int h, int w,
                                     int c, long m, void* stream) {
    S3D_CHECK_ARG(plane && grid && out, "sample_planes: null argument");
    S3D_CHECK_ARG(n >= 1 && h >= 1 && w >= 1 && m >= 0, "sample_planes: bad dims");
    return launch_sample_planes(plane, grid, out, n, h, w, c, m, (hipStream_t)stream);
}

extern "C" int s3d_nchw_to_nhwc(const float* in, float* out, int n, int c, int h, int w, void* stream) {
    S3D_CHECK_ARG(in && out, "nchw_to_nhwc: null argument");
    return launch_nchw_to_nhwc(in, out, n, c, h, w, c, (hipStream_t)stream);
}

extern "C" int s3d_nhwc_to_nchw(const float* in, float* out, int n, int c, int h, int w, void* stream) {
    S3D_CHECK_ARG(in && out, "nhwc_to_nchw: null argument");
    return launch_nhwc_to_nchw(in, out, n, c, h, w, (hipStream_t)stream);
}

#include "api_gt.inc"
#include "api_ldm.inc"
#include "api_train.inc"
#include "api_train_gt.inc"
