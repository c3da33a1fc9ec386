"""Latent-diffusion denoising U-Net (BASELINE configs[4]): host-module contract (CPU) and parity of
the HIP path against goldens captured from the REAL reference UNetModel (tests/golden/make_golden_ldm.py)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, ldm_inputs

LDM_FULL = dict(image_size=64, in_channels=8, out_channels=4, model_channels=192, attention_resolutions=[1, 2, 4, 8],
                num_res_blocks=2, channel_mult=[1, 2, 2, 4, 4], num_heads=8, use_scale_shift_norm=True,
                resblock_updown=True)
LDM_SMALL = dict(image_size=32, in_channels=8, out_channels=4, model_channels=32, attention_resolutions=[1, 2, 4],
                 num_res_blocks=1, channel_mult=[1, 2, 2], num_heads=4, use_scale_shift_norm=True,
                 resblock_updown=True)


@pytest.mark.parametrize("name,cfg", [("small", LDM_SMALL), ("full", LDM_FULL)])
def test_ldm_state_dict_contract(name, cfg):
    from slice3d_amd.ldm_unet import UNetModel
    want = {k: tuple(v) for k, v in json.load(open(os.path.join(GOLDEN, "state_dict_keys_ldm_%s.json" % name))).items()}
    with torch.device("meta"):
        m = UNetModel(backend="none", **cfg)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want


def _golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return z["y"], int(z["meta"][0]), int(z["meta"][1])


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("f32", 2e-4), ("f16x3", 2e-4)])
def test_ldm_small_matches_reference(prec, tol):
    from slice3d_amd.ldm_unet import UNetModel
    from slice3d_amd.weights import load_seeded
    y, batch, seed = _golden("ldm_small_b2")
    m = load_seeded(UNetModel(prec=prec, **LDM_SMALL), 0).cuda().eval()
    x, t, cf = ldm_inputs(LDM_SMALL, batch, seed)
    out = m(x.cuda(), t.cuda(), c_fmaps={k: v.cuda() for k, v in cf.items()}).cpu().numpy()
    assert out.shape == y.shape
    assert np.abs(out - y).max() < tol * max(1.0, float(np.abs(y).max()))


@pytest.mark.gpu
def test_ldm_full_config_matches_reference():
    """The Slice3D configuration (295 M parameters, 64x64x4 latent mosaic, 21 attention blocks up to 4 096 tokens)."""
    from slice3d_amd.ldm_unet import UNetModel
    from slice3d_amd.weights import load_seeded
    y, batch, seed = _golden("ldm_full_b1")
    m = load_seeded(UNetModel(**LDM_FULL), 0).cuda().eval()
    x, t, cf = ldm_inputs(LDM_FULL, batch, seed)
    out = m(x.cuda(), t.cuda(), c_fmaps={k: v.cuda() for k, v in cf.items()}).cpu().numpy()
    err = np.abs(out - y).max()
    print("LDM full configuration: max |out - reference| = %.3e (max |y| %.2f)" % (err, np.abs(y).max()))
    assert err < 2e-4 * max(1.0, float(np.abs(y).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg", [("ldm_small_b2", LDM_SMALL), ("ldm_full_b1", LDM_FULL)])
def test_ldm_fused_group_norm_convolution_equals_the_two_operator_form(name, cfg):
    """GroupNorm -> [FiLM] -> SiLU -> conv3x3 as one operator (s3d_group_norm_table_fwd + s3d_conv_gn_fwd: the convolution
    normalises while it stages its input tile, openaimodel.py:188-194, :229-236) against the same network with the
    GroupNorm applied by its own kernel (fuse_gn=False): the same statistics and the same affine map in a different
    association, so fp32 rounding apart (1e-5 of max|y|); both meet the reference golden.  Small config: the 32-channel
    tile and two-source (skip concat) inputs; full config: the 64-channel tile, split-K layers, 1536-channel concats."""
    from slice3d_amd.ldm_unet import UNetModel
    from slice3d_amd.weights import load_seeded
    y, batch, seed = _golden(name)
    x, t, cf = ldm_inputs(cfg, batch, seed)
    cf = {k: v.cuda() for k, v in cf.items()}
    outs = {}
    for fuse in (True, False):
        m = load_seeded(UNetModel(fuse_gn=fuse, **cfg), 0).cuda().eval()
        outs[fuse] = m(x.cuda(), t.cuda(), c_fmaps=cf).cpu().numpy()
        assert np.abs(outs[fuse] - y).max() < 2e-4 * max(1.0, float(np.abs(y).max())), fuse
    scale = max(1.0, float(np.abs(y).max()))
    d = np.abs(outs[True] - outs[False]).max() / scale
    print("LDM %s: fused vs two-operator GroupNorm: %.2e of max|y|" % (name, d))
    assert 0 < d < 2e-5      # (not bit-identical: the fused path really ran)
    # the split-K finish passes deferred to the next GroupNorm's statistics kernel (opt-in, measured neutral) against each
    # convolution finishing itself: the splits are added in the same order, bias and residual joined in the same order — the same bits
    m = load_seeded(UNetModel(defer_finish=True, **cfg), 0).cuda().eval()
    assert np.array_equal(m(x.cuda(), t.cuda(), c_fmaps=cf).cpu().numpy(), outs[True])
    # the ResBlocks' skip convolutions on a side stream (opt-in: a parallel branch of the sampler's HIP graph, measured slower):
    # the same kernels on the same data in another order of launch — the same bits
    m = load_seeded(UNetModel(branch_streams=True, **cfg), 0).cuda().eval()
    assert np.array_equal(m(x.cuda(), t.cuda(), c_fmaps=cf).cpu().numpy(), outs[True])


@pytest.mark.gpu
def test_small_map_convolution_kernel_is_bit_identical_to_the_tiled_kernel(tmp_path):
    """conv3x3_small_f16x3_kernel (conv.hip; the 4 x 4 maps) keeps the tiled kernel's chunk ranges per split and its order
    of products per accumulator: at batch 1 (same split counts) the step's output must not change by a bit; at batch 2
    (two pixel tiles, both images in one workgroup) the maps get their own split counts, so the outputs agree to rounding.
    Two processes (the switch is read once per process), each loading the full configuration once for both batch sizes."""
    import subprocess, sys
    outs = []
    for sw in ("0", "1"):
        f = str(tmp_path / ("y%s.pt" % sw))
        env = dict(os.environ, S3D_CONV_SMALL=sw)
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools",
                                                         "ldm_out.py"), f, "1", "2"], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f))
    assert torch.equal(outs[0][1], outs[1][1])
    assert (outs[0][2] - outs[1][2]).abs().max() < 2e-5 * outs[0][2].abs().max()


@pytest.mark.gpu
def test_ldm_single_pass_f16_mode_runs_and_reports_its_error():
    """prec='f16' (S3D_PREC_F16 through s3d_conv_fwd: one f16 MFMA per product in every convolution with 32-aligned channel
    counts; the attention operators keep the split form) — the precision BASELINE configs[4] names ("bf16").  Not fp32-class:
    it must be a sane approximation of the reference's output (5e-2 of max|y|; measured ~3e-3) and measurably different from the
    split-precision mode, which meets 2e-4."""
    from slice3d_amd.ldm_unet import UNetModel
    from slice3d_amd.weights import load_seeded
    y, batch, seed = _golden("ldm_full_b1")
    x, t, cf = ldm_inputs(LDM_FULL, batch, seed)
    cf = {k: v.cuda() for k, v in cf.items()}
    m16 = load_seeded(UNetModel(prec="f16", **LDM_FULL), 0).cuda().eval()
    out16 = m16(x.cuda(), t.cuda(), c_fmaps=cf).cpu().numpy()
    scale = max(1.0, float(np.abs(y).max()))
    e16 = np.abs(out16 - y).max() / scale
    print("LDM full configuration, single-pass f16: max |out - reference| / max|y| = %.3e" % e16)
    assert np.isfinite(out16).all() and 2e-4 < e16 < 5e-2


@pytest.mark.gpu
def test_ldm_full_config_at_128_latent_matches_reference():
    """BASELINE configs[4] names 256^2 slice generation: a 128x128x4 latent mosaic, 16 384-token attention at full
    resolution (openaimodel.py:278-377 materialises 16 384^2 x 8 attention weights there; the HIP kernel streams keys).
    Golden from the REAL reference UNetModel (tests/golden/make_golden_ldm128.py)."""
    from slice3d_amd.ldm_unet import UNetModel
    from slice3d_amd.weights import load_seeded
    cfg = dict(LDM_FULL, image_size=128)
    y, batch, seed = _golden("ldm_full128_b1")
    m = load_seeded(UNetModel(**cfg), 0).cuda().eval()
    x, t, cf = ldm_inputs(cfg, batch, seed)
    out = m(x.cuda(), t.cuda(), c_fmaps={k: v.cuda() for k, v in cf.items()}).cpu().numpy()
    assert out.shape == y.shape == (1, 4, 128, 128)
    err = np.abs(out - y).max()
    print("LDM 128x128 latent: max |out - reference| = %.3e (max |y| %.2f)" % (err, np.abs(y).max()))
    assert err < 2e-4 * max(1.0, float(np.abs(y).max()))


def test_ddim_schedule_matches_the_reference_sampler():
    """ldm_sampler's schedule (beta schedule of the yaml -> alphas_cumprod -> 200 uniform DDIM steps, eta = 1) against the
    arrays the REAL reference DDIMSampler.make_schedule produced (tests/golden/make_golden_ldm_ddim.py)."""
    from slice3d_amd.ldm_sampler import DDIMSampler
    z = np.load(os.path.join(GOLDEN, "ldm_ddim_small_b2.npz"))
    s = DDIMSampler(unet=None)
    s.make_schedule(200, 1.0)
    assert np.array_equal(s.ddim_timesteps, z["timesteps200"])
    for got, key in ((s.ddim_sigmas, "sigmas200"), (s.ddim_alphas, "alphas200"), (s.ddim_alphas_prev, "alphas_prev200")):
        assert np.abs(np.asarray(got, np.float64) - z[key]).max() <= 1e-7 * np.abs(z[key]).max(), key
    s.make_schedule(4, 1.0)
    assert np.array_equal(s.ddim_timesteps, z["timesteps4"])


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_ddim_sampler_reproduces_the_reference_run(prec, graph):
    """A complete 4-step DDIM run (eta = 1; ddim.py:56-203 — the reference's own loop, schedule and x_{t-1} update around
    the real UNetModel, golden from the imported reference) against slice3d_amd.ldm_sampler around the HIP UNet, fed the
    noise the reference drew: every intermediate latent and x0 prediction, replayed as a HIP graph and launched eagerly."""
    from slice3d_amd.ldm_sampler import DDIMSampler
    from slice3d_amd.ldm_unet import UNetModel
    from slice3d_amd.weights import load_seeded
    z = np.load(os.path.join(GOLDEN, "ldm_ddim_small_b2.npz"))
    batch, seed = int(z["meta"][0]), int(z["meta"][1])
    x8, _, cf = ldm_inputs(LDM_SMALL, batch, seed)
    m = load_seeded(UNetModel(prec=prec, **LDM_SMALL), 0).cuda().eval()
    smp = DDIMSampler(m, use_graph=graph)
    x_T, c_concat = x8[:, :4].contiguous().cuda(), x8[:, 4:].contiguous().cuda()
    samples, inter = smp.sample(4, x_T, c_concat, {k: v.cuda() for k, v in cf.items()}, eta=1.0,
                                noises=[torch.from_numpy(n) for n in z["noises"]])
    assert bool(smp._graph) == graph
    worst = 0.0
    for i in range(1, 5):
        for key in ("x_inter", "pred_x0"):
            ref = z[key][i]
            err = float(np.abs(inter[key][i].cpu().numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
            worst = max(worst, err)
            assert err < 5e-4, (key, i, err)
    assert float(np.abs(samples.cpu().numpy() - z["samples"]).max()) < 5e-4 * max(1.0, float(np.abs(z["samples"]).max()))
    print("ddim 4-step run (%s, graph=%s): worst relative deviation %.2e" % (prec, graph, worst))


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [True, False])
def test_ddim_sampler_sees_a_conditioning_latent_refilled_in_place(graph):
    """The captured graph reads a STATIC copy of c_concat: a caller that refills its conditioning tensor in place between
    two sample() calls (same address, same shapes — no re-capture) must get the second condition's result, i.e. the same
    latents a fresh sampler produces for it."""
    from slice3d_amd.ldm_sampler import DDIMSampler
    from slice3d_amd.ldm_unet import UNetModel
    from slice3d_amd.weights import load_seeded
    x8, _, cf = ldm_inputs(LDM_SMALL, 1, 11)
    m = load_seeded(UNetModel(prec="f16x3", **LDM_SMALL), 0).cuda().eval()
    cfd = {k: v.cuda() for k, v in cf.items()}
    x_T, c_a = x8[:, :4].contiguous().cuda(), x8[:, 4:].contiguous().cuda()
    c_b = torch.flip(c_a, dims=(-1,)).contiguous() * 0.5
    noises = [torch.randn(x_T.shape, generator=torch.Generator().manual_seed(5 + i)) for i in range(2)]
    smp = DDIMSampler(m, use_graph=graph)
    cond = c_a.clone()
    out_a, _ = smp.sample(2, x_T, cond, cfd, noises=noises)
    out_a = out_a.clone()
    cond.copy_(c_b)                                # in place: same data_ptr, no new capture
    out_b, _ = smp.sample(2, x_T, cond, cfd, noises=noises)
    fresh, _ = DDIMSampler(m, use_graph=False).sample(2, x_T, c_b, cfd, noises=noises)
    assert float((out_b - fresh).abs().max()) < 1e-5
    assert float((out_b - out_a).abs().max()) > 1e-3      # and the condition really matters


@pytest.mark.gpu
def test_group_norm_of_two_sources_equals_group_norm_of_the_concatenation():
    """s3d_group_norm2_fwd (the th.cat([h, hs.pop()]) of openaimodel.py:750, never materialised) == s3d_group_norm_fwd on
    torch.cat, bit for bit, on the fused one-launch path and on the sliced two-pass path (large map)."""
    import ctypes as C
    from slice3d_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    for n, hw, c0, c1 in ((2, 80, 96, 32), (1, 128 * 128, 192, 192)):
        x0 = torch.randn(n, hw, c0, generator=g).cuda()
        x1 = torch.randn(n, hw, c1, generator=g).cuda()
        c = c0 + c1
        gw, gb = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.1).cuda()
        film = (torch.randn(n, 2 * c, generator=g) * 0.3).cuda()
        cat = torch.cat([x0, x1], -1).contiguous()
        ya, yb = torch.empty_like(cat), torch.empty_like(cat)
        stats = torch.empty(n, 32, 200, device="cuda")
        _lib.check(lib.s3d_group_norm_fwd(cat.data_ptr(), gw.data_ptr(), gb.data_ptr(), film.data_ptr(), ya.data_ptr(),
                                          stats.data_ptr(), n, hw, c, 32, C.c_float(1e-5), 1, None), "gn")
        _lib.check(lib.s3d_group_norm2_fwd(x0.data_ptr(), c0, x1.data_ptr(), c1, gw.data_ptr(), gb.data_ptr(),
                                           film.data_ptr(), yb.data_ptr(), stats.data_ptr(), n, hw, 32, C.c_float(1e-5), 1,
                                           None), "gn2")
        torch.cuda.synchronize()
        assert torch.equal(ya, yb), (n, hw)


@pytest.mark.gpu
def test_ldm_primitives_match_torch():
    """GroupNorm(+FiLM+SiLU), QKVAttentionLegacy, resampling through the C ABI vs the torch ops the reference calls."""
    import ctypes as C
    import math
    import torch.nn.functional as F
    from slice3d_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    n, h, w, c = 2, 12, 10, 96
    x = torch.randn(n, c, h, w, generator=g)
    gn = torch.nn.GroupNorm(32, c)
    gn.weight.data = torch.rand(c, generator=g) + 0.5
    gn.bias.data = torch.randn(c, generator=g) * 0.1
    film = torch.randn(n, 2 * c, generator=g) * 0.3
    want = gn(x) * (1 + film[:, :c, None, None]) + film[:, c:, None, None]
    want = F.silu(want).permute(0, 2, 3, 1).contiguous()
    xc = x.permute(0, 2, 3, 1).contiguous().cuda()
    y = torch.empty_like(xc)
    stats = torch.empty(n, 32, 200, device="cuda")
    gw, gb, fc = gn.weight.detach().cuda(), gn.bias.detach().cuda(), film.cuda()   # keep the device copies alive
    _lib.check(lib.s3d_group_norm_fwd(xc.data_ptr(), gw.data_ptr(), gb.data_ptr(), fc.data_ptr(), y.data_ptr(),
                                      stats.data_ptr(), n, h * w, c, 32, C.c_float(1e-5), 1, None), "gn")
    torch.cuda.synchronize()
    assert (y.cpu() - want.detach()).abs().max() < 2e-5
    # attention: T not a multiple of 64, head width 24 (not a multiple of 16)
    heads, ch, T = 4, 24, 150
    qkv = torch.randn(n, heads * 3 * ch, T, generator=g)
    q, k, v = qkv.reshape(n * heads, ch * 3, T).split(ch, dim=1)
    sc = 1 / math.sqrt(math.sqrt(ch))
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", q * sc, k * sc), dim=-1)
    want = torch.einsum("bts,bcs->bct", wgt, v).reshape(n, -1, T).permute(0, 2, 1).contiguous()
    qc = qkv.permute(0, 2, 1).contiguous().cuda()
    out = torch.empty(n, T, heads * ch, device="cuda")
    for prec in (_lib.PREC_F32, _lib.PREC_F16X3):   # (the split-precision attention kernel is opt-in: same kernel here)
        out.zero_()
        _lib.check(lib.s3d_qkv_attention_fwd(qc.data_ptr(), out.data_ptr(), n, T, heads, ch, prec, None), "attn")
        assert (out.cpu() - want).abs().max() < 2e-5, (prec, float((out.cpu() - want).abs().max()))
    for ch2, T2 in ((48, 100), (96, 40), (8, 300)):   # the other head widths of the full configuration
        qkv2 = torch.randn(1, 2 * 3 * ch2, T2, generator=g)
        q2, k2, v2 = qkv2.reshape(2, ch2 * 3, T2).split(ch2, dim=1)
        sc2 = 1 / math.sqrt(math.sqrt(ch2))
        w2 = torch.softmax(torch.einsum("bct,bcs->bts", q2 * sc2, k2 * sc2), dim=-1)
        want2 = torch.einsum("bts,bcs->bct", w2, v2).reshape(1, -1, T2).permute(0, 2, 1).contiguous()
        qc2 = qkv2.permute(0, 2, 1).contiguous().cuda()
        out2 = torch.empty(1, T2, 2 * ch2, device="cuda")
        _lib.check(lib.s3d_qkv_attention_fwd(qc2.data_ptr(), out2.data_ptr(), 1, T2, 2, ch2, _lib.PREC_F16X3, None), "attn")
        assert (out2.cpu() - want2).abs().max() < 2e-5, (ch2, float((out2.cpu() - want2).abs().max()))
    # the f16-MFMA attention with three-way split logits (ldm_attn.hip) holds the SAME 2e-5 bound on random inputs; ragged T
    # (tails of the 64-key blocks and of the 128-query workgroups), all four head widths, batch 2
    # 48-wide heads (round 6: two k-steps; key split over 4 workgroups at 1 024 / 1 100 tokens, none at 150 / 257, two query
    # tiles per wave and no split at 2 x 4 200)
    for heads3, ch3, T3, n3 in ((4, 24, 150, 2), (8, 24, 1100, 1), (2, 32, 64, 1), (3, 16, 257, 1), (2, 8, 129, 2),
                                 (8, 24, 4200, 2),    # >= 512 workgroups, two query tiles per wave
                                 (8, 48, 1024, 1), (8, 48, 1100, 1), (2, 48, 150, 2), (3, 48, 257, 1), (8, 48, 4200, 2)):
        qkv3 = torch.randn(n3, heads3 * 3 * ch3, T3, generator=g)
        q3, k3, v3 = qkv3.double().reshape(n3 * heads3, ch3 * 3, T3).split(ch3, dim=1)
        sc3 = 1 / math.sqrt(math.sqrt(ch3))
        w3 = torch.softmax(torch.einsum("bct,bcs->bts", q3 * sc3, k3 * sc3), dim=-1)
        want3 = torch.einsum("bts,bcs->bct", w3, v3).reshape(n3, -1, T3).permute(0, 2, 1).contiguous()
        qc3 = qkv3.permute(0, 2, 1).contiguous().cuda()
        out3 = torch.zeros(n3, T3, heads3 * ch3, device="cuda")
        nb = lib.s3d_qkv_attention_ws_bytes(n3, T3, heads3, ch3)
        assert nb > 0
        ws3 = torch.empty(nb, dtype=torch.uint8, device="cuda")
        _lib.check(lib.s3d_qkv_attention_ws_fwd(qc3.data_ptr(), out3.data_ptr(), n3, T3, heads3, ch3, ws3.data_ptr(), nb, None),
                   "attn_ws")
        err3 = float((out3.cpu().double() - want3).abs().max())
        assert err3 < 2e-5, (heads3, ch3, T3, err3)
    assert lib.s3d_qkv_attention_ws_bytes(1, 64, 8, 96) == 0      # wide heads stay on s3d_qkv_attention_fwd
    # the timestep-embedding layers (openaimodel.py:718-719 time_embed, the ResBlocks' emb_layers): the streaming form
    # (K % 4 == 0: four rows per wave, up to four images per pass) and the scalar form, ragged M, more than four images
    for nl, kl, ml, silu in ((1, 768, 36096 // 8 + 3, 1), (5, 256, 4099, 1), (5, 192, 70, 0), (3, 770, 33, 1)):
        xl = torch.randn(nl, kl, generator=g)
        lin = torch.nn.Linear(kl, ml)
        wantl = lin(F.silu(xl) if silu else xl).detach()
        xg, wg, bg = xl.cuda(), lin.weight.detach().cuda(), lin.bias.detach().cuda()
        outl = torch.zeros(nl, ml, device="cuda")
        _lib.check(lib.s3d_small_linear_fwd(xg.data_ptr(), wg.data_ptr(), bg.data_ptr(), outl.data_ptr(), nl, kl, ml, silu, None), "lin")
        assert (outl.cpu() - wantl).abs().max() < 2e-5, (nl, kl, ml, float((outl.cpu() - wantl).abs().max()))
    up = torch.empty(n, 2 * h, 2 * w, c, device="cuda")
    _lib.check(lib.s3d_resample2x_fwd(xc.data_ptr(), up.data_ptr(), n, h, w, c, 1, None), "up")
    assert torch.equal(up.cpu(), F.interpolate(x, scale_factor=2, mode="nearest").permute(0, 2, 3, 1))
    # c_fmaps injection with the feature map in the reference's NCHW layout (openaimodel.py:735-746): one launch
    fm = torch.randn(n, c, h, w, generator=g)
    inj = torch.empty_like(xc)
    fmc = fm.cuda()
    _lib.check(lib.s3d_add_nchw_fwd(xc.data_ptr(), fmc.data_ptr(), inj.data_ptr(), n, c, h, w, None), "add_nchw")
    assert torch.equal(inj.cpu(), (x + fm).permute(0, 2, 3, 1).contiguous())
    dn = torch.empty(n, h // 2, w // 2, c, device="cuda")
    _lib.check(lib.s3d_resample2x_fwd(xc.data_ptr(), dn.data_ptr(), n, h, w, c, 0, None), "down")
    assert (dn.cpu() - F.avg_pool2d(x, 2).permute(0, 2, 3, 1)).abs().max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("k,n,rows", [(128, 384, 70001), (96, 120, 65600), (32, 64, 131072), (64, 128, 66000)])
def test_row_linear_path_matches_torch(k, n, rows):
    """1x1 layers with K <= 128 on >= 65 536 rows take lin_rows_f16x3_kernel (rows kept in registers, all output tiles
    walked by one wave) in split precision: ragged row tails, an output width below its padded width, bias and residual;
    the f32 mode (tile menu) is the second opinion."""
    import torch
    from slice3d_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(k + n)
    w = (torch.randn(n, k, 1, 1, generator=g) * 0.1).cuda()
    b = torch.randn(n, generator=g).cuda()
    x = torch.randn(1, 1, rows, k, generator=g).cuda()
    res = torch.randn(1, 1, rows, n, generator=g).cuda()
    nb = lib.s3d_conv_packed_bytes(n, k, 0, 1)
    buf = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.s3d_conv_pack(w.data_ptr(), b.data_ptr(), n, k, 0, 1, buf.data_ptr(), nb, None), "pack")
    ws = torch.empty(1 << 20, dtype=torch.float32, device="cuda")
    sel = torch.cat([torch.arange(0, 300), torch.arange(rows - 300, rows)])          # head and ragged tail
    want = (torch.nn.functional.linear(x[0, 0, sel].double().cpu(), w[:, :, 0, 0].double().cpu(), b.double().cpu())
            + res[0, 0, sel].double().cpu())
    outs = []
    for prec in (1, 0):
        out = torch.full((1, 1, rows, n), float("nan"), device="cuda")
        _lib.check(lib.s3d_conv_fwd(buf.data_ptr(), x.data_ptr(), None, res.data_ptr(), out.data_ptr(), 1, 1, rows, n, k,
                                    0, 1, prec, ws.data_ptr(), ws.numel() * 4, None), "conv")
        assert torch.isfinite(out).all()
        err = float((out[0, 0, sel].double().cpu() - want).abs().max())
        assert err < 2e-5, (prec, err)
        outs.append(out)
    assert float((outs[0] - outs[1]).abs().max()) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,cin0,cin1,cout,ks", [(1, 4, 4, 768, 0, 768, 3), (1, 4, 4, 384, 384, 768, 3), (2, 4, 4, 256, 0, 128, 3),
                                                      (4, 4, 4, 256, 0, 128, 3), (1, 4, 4, 768, 0, 2304, 1), (1, 8, 8, 384, 384, 768, 1), (4, 4, 4, 256, 0, 192, 1),
                                                      (3, 3, 5, 64, 0, 64, 3), (2, 3, 5, 64, 32, 64, 3), (1, 2, 7, 96, 32, 128, 1),
                                                      (1, 16, 16, 384, 0, 384, 1), (1, 9, 9, 64, 32, 128, 1)])   # 1x1 beyond 64 pixels: the implicit-GEMM kernel
def test_small_map_convolutions_match_torch(n, h, w, cin0, cin1, cout, ks):
    """conv3x3_small_f16x3_kernel (conv.hip): the split-K kernel of maps of a few pixels — 3x3 (zero border, all images of the
    batch in one workgroup, odd map shapes) and 1x1, one and two sources, bias and residual — against a float64 convolution."""
    import torch.nn.functional as F
    from slice3d_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n * 1000 + h * 100 + cin0 + cout + ks)
    cin = cin0 + cin1
    wt = torch.randn(cout, cin, ks, ks, generator=g) * (1.0 / (cin * ks * ks) ** 0.5)
    b = torch.randn(cout, generator=g)
    x = torch.randn(n, cin, h, w, generator=g)
    res = torch.randn(n, cout, h, w, generator=g)
    want = F.conv2d(x.double(), wt.double(), b.double(), padding=ks // 2) + res.double()
    xc = x.permute(0, 2, 3, 1).contiguous().cuda()
    x0 = xc[..., :cin0].contiguous()
    x1 = xc[..., cin0:].contiguous() if cin1 else None
    rc = res.permute(0, 2, 3, 1).contiguous().cuda()
    wg, bg = wt.cuda(), b.cuda()
    nb = lib.s3d_conv_packed_bytes(cout, cin0, cin1, ks)
    buf = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.s3d_conv_pack(wg.data_ptr(), bg.data_ptr(), cout, cin0, cin1, ks, buf.data_ptr(), nb, None), "pack")
    ws = torch.empty(1 << 22, dtype=torch.float32, device="cuda")
    out = torch.full((n, h, w, cout), float("nan"), device="cuda")
    _lib.check(lib.s3d_conv_fwd(buf.data_ptr(), x0.data_ptr(), x1.data_ptr() if x1 is not None else None, rc.data_ptr(), out.data_ptr(),
                                n, h, w, cout, cin0, cin1, ks, _lib.PREC_F16X3, ws.data_ptr(), ws.numel() * 4, None), "conv")
    torch.cuda.synchronize()
    err = float((out.cpu().double().permute(0, 3, 1, 2) - want).abs().max())
    assert err < 2e-5, err
