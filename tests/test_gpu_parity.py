"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
 (1) the committed golden vectors captured from the real reference,
 (2) the CPU oracle on fresh seeded inputs (incl. BASELINE's full 256^2 x 12-slice size),
 (3) size-independent properties (chunk invariance, permutation equivariance, batch-major order,
     dense-grid == explicit make_3d_grid queries).
Tolerance: 1e-4 absolute on sdf / occupancy logits (BASELINE.json north_star); images 1e-4."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, GOLDEN_CASES, golden_feed, load_golden, seeded_sd_from_shapes

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _shapes(n_slices):
    shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(GOLDEN, "state_dict_keys.json"))).items()}
    shapes["slices_generator.emds.weight"] = (n_slices, 128)
    return shapes


_models = {}
_oracle_cache = {}     # oracle results shared by the precision-parametrised tests (same inputs, same weights)


PRECS = ("f32", "f16x3")     # f16x3 is what bench.py times: every oracle / golden comparison runs in both modes


def get_model(n_slices, mode, prec="f32"):
    from slice3d_amd.models import Slices3DRegModel
    from slice3d_amd.weights import load_seeded
    key = (n_slices, mode, prec)
    if key not in _models:
        m = Slices3DRegModel(n_slices=n_slices, mode=mode, prec=prec)
        load_seeded(m, 0)
        _models[key] = m.cuda().eval()
    return _models[key]


def to_gpu(fd):
    return {k: v.cuda() for k, v in fd.items()}


def _oracle_case(name, compute):
    """Oracle outputs of a fixed-seed case as a dict of numpy arrays: the committed tests/golden/oracle_<name>.npz (written
    by tests/golden/make_oracle_golden.py from `compute` itself) or, without the file / with S3D_LIVE_ORACLE=1, computed
    here.  Shared by the precision-parametrised tests."""
    from helpers import load_oracle_golden
    if name not in _oracle_cache:
        z = load_oracle_golden(name)
        _oracle_cache[name] = compute() if z is None else z
    return _oracle_cache[name]


def _probe(x, phase=0, step=4):
    """The strided pixel probe the oracle goldens keep of an image tensor (..., H, W)."""
    return x[..., (1 + phase) % step::step, (2 + phase) % step::step]


def _c4_indices(nx=256):
    """Grid indices of the dense-grid test: 4096 random ones, the 8 corners, both sides of every 262 144-query pass boundary,
    512 of the last pass, the last index."""
    n = nx ** 3
    rng = np.random.default_rng(5)
    corners = [(ix * nx + iy) * nx + iz for ix in (0, nx - 1) for iy in (0, nx - 1) for iz in (0, nx - 1)]
    edges = [k * 262144 + d for k in range(1, 64) for d in (-1, 0)]
    return np.unique(np.concatenate([rng.choice(n, 4096, replace=False), corners, edges,
                                     rng.integers(n - 262144, n, 512), [n - 1]])).astype(np.int64)


def _c4_points(idx, nx=256):
    idx_t = torch.from_numpy(idx)
    lin = torch.linspace(-0.5, 0.5, nx)
    return torch.stack([lin[idx_t // (nx * nx)], lin[(idx_t // nx) % nx], lin[idx_t % nx]], -1).unsqueeze(0)


def _oc_full256():
    """Oracle at BASELINE configs[1]'s inference shape (256^2 x 12 slices, seed 2024): sdf on the 4096-query subset of the
    full-size test, slices_rec on the pixel probe, and sdf on the dense-grid test's indices (same image, same pyramid)."""
    from oracle import ref_cpu
    from slice3d_amd.synth import make_feed_dict
    sd = seeded_sd_from_shapes(_shapes(12))
    fd = make_feed_dict(1, 256, 100000, 12, seed=2024, with_slices=False)
    idx = torch.from_numpy(np.random.default_rng(0).choice(100000, 4096, replace=False))
    with torch.no_grad():
        feats, rec = ref_cpu.unet_forward(sd, fd["img_input"], 12)
        qr = ref_cpu.rotate_queries({"qry_norot": fd["qry_norot"][:, idx]}, "test")
        sub = ref_cpu.decode_points(sd, feats, qr, fd["trans_mat_wo_rot_tp"], 12)
        qc = ref_cpu.rotate_queries({"qry_norot": _c4_points(_c4_indices())}, "test")
        c4 = ref_cpu.decode_points(sd, feats, qc, fd["trans_mat_wo_rot_tp"], 12)
    return {"sdf_sub": sub.numpy(), "rec_probe": _probe(rec).contiguous().numpy(), "c4_sdf": c4.reshape(-1).numpy()}


def _oc_white():
    """fp32 and fp64 oracle on SURVEY 8(d)'s white-noise inputs (seed 1234, 256^2, 6000 queries)."""
    from oracle import ref_cpu
    from slice3d_amd.synth import make_feed_dict
    fd = make_feed_dict(1, 256, 6000, 12, seed=1234, smooth=False, with_slices=False)
    sd = seeded_sd_from_shapes(_shapes(12))
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    r32 = ref_cpu.forward(sd, fd, mode="test", n_slices=12, with_vgg=False)
    r64 = ref_cpu.forward(sd64, {k: v.double() for k, v in fd.items()}, mode="test", n_slices=12, with_vgg=False)
    return {"sdf32": r32["sdf_pred"].numpy(), "sdf64": r64["sdf_pred"].numpy(),
            "rec64_probe": _probe(r64["slices_rec"]).contiguous().numpy(),      # float64
            "e_img_ref": np.array([float((r32["slices_rec"].double() - r64["slices_rec"]).abs().max())])}


SWEEP_SEED, SWEEP_N = 20260928, 24


def _oc_sweep():
    """Oracle outputs of the 24 sweep shapes: sdf whole, slices_rec on the pixel probe (phase = case index)."""
    from oracle import ref_cpu
    from slice3d_amd.synth import make_feed_dict
    z = {}
    for i, (b, s, q, ns, mode) in enumerate(_sweep_cases(SWEEP_SEED, SWEEP_N)):
        fd = make_feed_dict(b, s, q, ns, seed=7000 + i, with_slices=False)
        ref = ref_cpu.forward(seeded_sd_from_shapes(_shapes(ns)), fd, mode=mode, n_slices=ns, with_vgg=False)
        z["sdf:%d" % i] = ref["sdf_pred"].numpy()
        z["rec:%d" % i] = _probe(ref["slices_rec"], i).contiguous().numpy()
    return z


def test_native_library_is_loaded():
    from slice3d_amd import _lib
    lib = _lib.load()
    assert lib.s3d_version() >= 100
    maps = open("/proc/self/maps").read()
    assert "libslice3d_hip.so" in maps


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_forward_matches_reference_golden(name, prec):
    g = load_golden(name)
    model = get_model(g["n_slices"], g["mode"], prec)
    fd = golden_feed(g)
    qry_before = fd["qry_norot"].clone()
    out = model(to_gpu(fd))
    torch.cuda.synchronize()
    err = np.abs(out["sdf_pred"].cpu().numpy() - g["sdf_pred"]).max()
    assert err < TOL, err
    rec = out["slices_rec"][:, :, ::4, ::4].cpu().numpy()
    assert np.abs(rec - g["slices_rec_strided"]).max() < TOL
    assert out["slices_rec"].shape == (g["batch"], 3 * g["n_slices"], g["size"], g["size"])
    assert torch.equal(fd["qry_norot"], qry_before)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_pyramid_matches_reference_golden(name, prec):
    g = load_golden(name)
    model = get_model(g["n_slices"], g["mode"], prec)
    feats, rec = model.slices_generator(torch.from_numpy(g["img_input"]).cuda())
    for l, f in enumerate(feats):
        assert tuple(f.shape) == tuple(g["pyr%d_shape" % l])
        got = f.reshape(-1)[torch.from_numpy(g["pyr%d_idx" % l]).cuda()].cpu().numpy()
        ref = g["pyr%d_val" % l]
        assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", ["g2_s64_n12_q2048_test", "g3_s32_n12_q512_b2_train"])
def test_decoder_stage_rows_match_reference_golden(name, prec):
    """SURVEY 8(a) a-10 / a-11 stage by stage on the HIP path: fc_p / fc_s rows (models.py:79-82) and token 0 after each of
    the three encoder layers (models.py:83) of the first 64 queries, captured from the REAL reference by forward hooks
    (tests/golden/make_golden.py), against s3d_decode_points_stages_fwd (the product kernels; the capture copies rows
    between them).  The oracle is checked against the same arrays in tests/test_oracle.py."""
    g = load_golden(name)
    model = get_model(g["n_slices"], g["mode"], prec)
    fd = to_gpu(golden_feed(g))
    code = model.encode(fd)
    st = model.decode_stages(fd["qry_norot"], code)
    torch.cuda.synchronize()
    assert np.abs(st["sdf"].cpu().numpy() - g["sdf_pred"]).max() < TOL
    assert torch.equal(st["sdf"], model.decode_sdf(fd["qry_norot"], code))        # the capture does not disturb the path
    n = g["fc_s_rows"].shape[0]
    tol = 5e-5     # the oracle's own bound (tests/test_oracle.py); measured 1-2e-5 in both modes
    assert np.abs(st["fc_p"][0, :n].cpu().numpy() - g["fc_p_rows"]).max() < 1e-5
    assert np.abs(st["fc_s"][0, :n].cpu().numpy() - g["fc_s_rows"]).max() < tol
    worst = 0.0
    for i in range(3):
        e = float(np.abs(st["layer%d" % i][0, :n].cpu().numpy() - g["layer%d_tok0" % i]).max())
        worst = max(worst, e)
        assert e < tol, (i, e)
    print("stage rows %s (%s): worst layer deviation %.2e" % (name, prec, worst))


@pytest.mark.parametrize("prec", PRECS)
def test_helper_ops_match_golden(prec):
    g = load_golden("g3_s32_n12_q512_b2_train")
    model = get_model(12, "train", prec)
    from oracle import ref_cpu
    fd = golden_feed(g)
    qr = ref_cpu.rotate_queries(fd, "train")
    pts = model.project_coord(qr.cuda(), fd["trans_mat_wo_rot_tp"].cuda())
    assert np.abs(pts.cpu().numpy() - g["img_pts"]).max() < 1e-6
    feats, _ = model.slices_generator(fd["img_input"].cuda())
    b, q, ns = g["batch"], g["n_qry"], 12
    pts_t = torch.from_numpy(g["img_pts"]).view(b, 1, q, 2).expand(-1, ns, -1, -1).reshape(b * ns, q, 2)
    s2 = model.sample_from_planes(feats[2], pts_t[:, :16].contiguous().cuda())
    assert s2.shape == (b * ns, 1, 16, 128)
    assert np.abs(s2.squeeze(1).cpu().numpy() - g["sample_l2"]).max() < 5e-5


@pytest.mark.parametrize("b,s,q,ns,mode", [(1, 64, 1500, 12, "test"), (3, 48, 257, 12, "train"),
                                           (1, 96, 33, 7, "train"), (2, 16, 1, 12, "test"),
                                           (1, 32, 16, 1, "train"), (1, 32, 1000, 11, "test"),
                                           (2, 32, 700, 2, "train")])
@pytest.mark.parametrize("prec", PRECS)
def test_forward_matches_oracle(b, s, q, ns, mode, prec):
    """Ragged / edge shapes: Q not a multiple of 16, Q=1, 1 / 2 / 7 / 11 slices (2: three valid rows of the sixteen-row
    attention tile; the C ABI stops at the reference's 12), 16^2 images, B=3, several attention items per workgroup."""
    from oracle import ref_cpu
    from slice3d_amd.synth import make_feed_dict
    model = get_model(ns, mode, prec)
    sd = seeded_sd_from_shapes(_shapes(ns))
    fd = make_feed_dict(b, s, q, ns, seed=100 + q, with_slices=False)
    out = model(to_gpu(fd))
    ref = ref_cpu.forward(sd, fd, mode=mode, n_slices=ns, with_vgg=False)
    assert (out["sdf_pred"].cpu() - ref["sdf_pred"]).abs().max() < TOL
    assert (out["slices_rec"].cpu() - ref["slices_rec"]).abs().max() < TOL


@pytest.mark.parametrize("prec", PRECS)
def test_full_size_256_matches_oracle(prec):
    """BASELINE configs[1] shape: 256^2 x 12 slices; 100k queries decoded on the GPU, a 4096-query subset and every fourth
    pixel of slices_rec checked against the oracle (committed outputs: _oc_full256)."""
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "test", prec)
    fd = make_feed_dict(1, 256, 100000, 12, seed=2024, with_slices=False)
    out = model(to_gpu(fd))
    sdf = out["sdf_pred"].cpu()
    assert sdf.shape == (1, 100000) and torch.isfinite(sdf).all()
    idx = torch.from_numpy(np.random.default_rng(0).choice(100000, 4096, replace=False))
    z = _oracle_case("full256", _oc_full256)
    assert (sdf[:, idx] - torch.from_numpy(z["sdf_sub"])).abs().max() < TOL
    rec = out["slices_rec"].cpu().view(12, 3, 256, 256)
    assert torch.isfinite(rec).all()
    assert (_probe(rec) - torch.from_numpy(z["rec_probe"])).abs().max() < TOL


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", ["g1_c1_s64_n4_q1000_train", "g3_s32_n12_q512_b2_train"])
def test_vgg_loss_matches_reference_golden(name, prec):
    """Full reference forward contract incl. vgg_loss (models.py:86-94)."""
    g = load_golden(name)
    model = get_model(g["n_slices"], g["mode"], prec)
    out = model(to_gpu(golden_feed(g)))
    want = float(g["vgg_loss"])
    assert abs(float(out["vgg_loss"]) - want) < 2e-5 * abs(want) + 1e-8, (float(out["vgg_loss"]), want)


@pytest.mark.parametrize("prec", PRECS)
def test_vgg_loss_matches_oracle_128(prec):
    from oracle import ref_cpu
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "train", prec)
    sd = seeded_sd_from_shapes(_shapes(12))
    fd = make_feed_dict(1, 128, 64, 12, seed=31)
    out = model(to_gpu(fd))
    ref = ref_cpu.forward(sd, fd, mode="train", n_slices=12)
    assert abs(float(out["vgg_loss"]) - float(ref["vgg_loss"])) < 2e-5 * float(ref["vgg_loss"])
    assert float(model.vgg_loss(fd["img_slices"].view(12, 3, 128, 128).cuda(), fd["img_slices"].cuda())) == 0.0


@pytest.mark.parametrize("prec", PRECS)
def test_chunk_invariance_and_permutation(prec):
    """Queries are independent given the pyramid: decoding in one call, in chunks (Generator3D's
    eval_points pattern) or in a permuted order gives the same value per query, bit for bit."""
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "test", prec)
    fd = to_gpu(make_feed_dict(1, 64, 5000, 12, seed=9, with_slices=False))
    code = model.encode(fd)
    full = model.decode_sdf(fd["qry_norot"], code)
    parts = [model.decode_sdf(fd["qry_norot"][:, s:s + 1234].contiguous(), code) for s in range(0, 5000, 1234)]
    assert torch.equal(full, torch.cat(parts, 1))
    perm = torch.randperm(5000, device="cuda")
    permuted = model.decode_sdf(fd["qry_norot"][:, perm].contiguous(), code)
    assert torch.equal(full[:, perm], permuted)


@pytest.mark.parametrize("prec", PRECS)
def test_sorted_and_unsorted_decodes_agree_across_chunks(prec):
    """>= 4096 queries per object are decoded in image-space locality order, fewer in caller order.  Three objects x
    100k queries span two decode chunks (the boundary falls inside the last object); decoding the same queries 3000
    at a time must give the same bits (train-mode rotation path, 128^2)."""
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "train", prec)
    fd = to_gpu(make_feed_dict(3, 128, 100000, 12, seed=19, with_slices=False))
    code = model.encode(fd)
    kw = dict(obj_rot_mat=fd["obj_rot_mat"], trans_mat_wo_rot_tp=fd["trans_mat_wo_rot_tp"])
    full = model.decode_sdf(fd["qry_norot"], code, **kw)
    assert torch.isfinite(full).all()
    for s in (0, 33000, 97000):
        part = model.decode_sdf(fd["qry_norot"][:, s:s + 3000].contiguous(), code, **kw)
        assert torch.equal(full[:, s:s + 3000], part), s


@pytest.mark.parametrize("prec", PRECS)
def test_decode_is_bit_reproducible_run_to_run(prec):
    """The decode has no atomics and no order-dependent reductions: repeated runs must agree bit for bit (this also
    guards the LDS-DMA weight rings, whose publishing barrier once let a wave read a chunk before it had landed)."""
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "test", prec)
    for b, s, q in ((1, 256, 100000), (2, 64, 4500)):
        fd = to_gpu(make_feed_dict(b, s, q, 12, seed=77, with_slices=False))
        code = model.encode(fd)
        first = model.decode_sdf(fd["qry_norot"], code).clone()
        for _ in range(8):
            assert torch.equal(model.decode_sdf(fd["qry_norot"], code), first)


@pytest.mark.parametrize("prec", ["f16x3", "f16"])
def test_fused_last_layer_is_bit_identical(prec):
    """Round 6: the token-0 attention block of the last layer is ONE kernel (csrc/decode_last.hip: absorbed q GEMM, 13-row
    mixing step, absorbed output GEMM + residual) instead of a row copy, two row GEMMs and the mixing kernel.  Every
    accumulator adds the same products in the same order, so the two forms must agree bit for bit — on odd group counts
    (the kernel walks two groups at a time), on fewer than 12 slices (T < 13 tokens) and on ragged query counts."""
    from slice3d_amd import _lib
    from slice3d_amd.synth import make_feed_dict
    lib = _lib.load()
    try:
        for ns, b, s, q in ((12, 1, 64, 16), (12, 2, 64, 4500), (12, 1, 64, 16 * 37 + 5), (5, 2, 64, 3001), (3, 1, 64, 1),
                            (12, 2, 128, 150000)):
            model = get_model(ns, "test", prec)
            fd = to_gpu(make_feed_dict(b, s, q, ns, seed=91 + ns, with_slices=False))
            code = model.encode(fd)
            assert lib.s3d_decode_set_last_fused(0) == 0
            four = model.decode_sdf(fd["qry_norot"], code).clone()
            assert lib.s3d_decode_set_last_fused(1) == 0
            for _ in range(2):
                assert torch.equal(model.decode_sdf(fd["qry_norot"], code), four), (ns, b, s, q)
        assert lib.s3d_decode_set_last_fused(2) != 0
    finally:
        lib.s3d_decode_set_last_fused(1)


@pytest.mark.parametrize("prec", PRECS)
def test_shared_footprint_sampler_is_bit_identical(prec):
    """Round 6: the token builder evaluates the three folded pyramid levels of a group of 16 queries through the group's
    shared 4 x 4 pixel window on the fp32 MFMA when the group's footprints fit one, per lane otherwise (csrc/decode.hip).
    The MFMA is a k-ordered fmaf chain, exact zeros in it change nothing and the window visits a query's taps in the
    per-lane order: both forms must give the same bits — on sorted queries (windows nearly always), on unsorted ones
    (< 4096 per object: mixed), on the dense grid, on ragged counts and few slices."""
    from slice3d_amd import _lib
    from slice3d_amd.synth import make_feed_dict
    lib = _lib.load()
    try:
        for ns, b, s, q in ((12, 1, 64, 16), (12, 2, 64, 3000), (12, 1, 256, 100000), (5, 2, 128, 4097), (3, 1, 64, 1),
                            (12, 2, 128, 20000)):
            model = get_model(ns, "test", prec)
            fd = to_gpu(make_feed_dict(b, s, q, ns, seed=57 + ns, with_slices=False))
            code = model.encode(fd)
            assert lib.s3d_decode_set_shared_footprint(0) == 0
            lanes = model.decode_sdf(fd["qry_norot"], code).clone()
            assert lib.s3d_decode_set_shared_footprint(1) == 0
            for _ in range(2):
                assert torch.equal(model.decode_sdf(fd["qry_norot"], code), lanes), (ns, b, s, q)
        model = get_model(12, "test", prec)
        fd = to_gpu(make_feed_dict(1, 64, 16, 12, seed=5, with_slices=False))
        code = model.encode(fd)
        assert lib.s3d_decode_set_shared_footprint(0) == 0
        lanes = model.decode_grid(code, 40).clone()
        assert lib.s3d_decode_set_shared_footprint(1) == 0
        assert torch.equal(model.decode_grid(code, 40), lanes)
        assert lib.s3d_decode_set_shared_footprint(2) != 0
    finally:
        lib.s3d_decode_set_shared_footprint(1)


def test_two_decode_lanes_and_smaller_passes_give_the_same_bits():
    """The pass of >= 131 072 queries is decoded as two halves whose layer chains run on the caller's stream and on the
    library's side stream (s3d_decode_set_lanes, api.hip): same bits as everything on one stream, and as a decode whose
    workspace only allows smaller passes (s3d_decode_workspace_bytes_min: the pass is halved until it fits); a workspace
    below the minimum is refused loudly."""
    import ctypes as C
    from slice3d_amd import _lib
    from slice3d_amd.synth import make_feed_dict
    lib = _lib.load()
    model = get_model(12, "test", "f16x3")
    fd = to_gpu(make_feed_dict(2, 64, 150000, 12, seed=31, with_slices=False))
    code = model.encode(fd)
    try:
        assert lib.s3d_decode_set_lanes(1) == 0
        one = model.decode_sdf(fd["qry_norot"], code).clone()
        assert lib.s3d_decode_set_lanes(2) == 0
        for _ in range(3):
            assert torch.equal(model.decode_sdf(fd["qry_norot"], code), one)
        assert lib.s3d_decode_set_lanes(3) != 0
    finally:
        lib.s3d_decode_set_lanes(1)
    b, q = 2, 150000
    full, small = lib.s3d_decode_workspace_bytes(b, q, 12), lib.s3d_decode_workspace_bytes_min(b, q, 12)
    assert small < full
    out = torch.empty((b, q), dtype=torch.float32, device="cuda")
    tm = fd["trans_mat_wo_rot_tp"]
    for nbytes, ok in ((small, True), ((small + full) // 2, True), (small - 4096, False)):
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
        out.zero_()
        rc = lib.s3d_decode_points_fwd(model._head_packed.data_ptr(), C.byref(code._latent_struct), fd["qry_norot"].data_ptr(),
                                       None, tm.data_ptr(), 1, out.data_ptr(), b, q, 12, model.prec, ws.data_ptr(), nbytes,
                                       None)
        torch.cuda.synchronize()
        if ok:
            assert rc == 0 and torch.equal(out, one), nbytes
        else:
            assert rc != 0 and b"workspace" in lib.s3d_last_error()


@pytest.mark.parametrize("prec", PRECS)
def test_batch_items_are_independent(prec):
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "train", prec)
    fd = to_gpu(make_feed_dict(2, 32, 100, 12, seed=4, with_slices=False))
    both = model(fd)["sdf_pred"]
    for b in range(2):
        one = model({k: v[b:b + 1].contiguous() for k, v in fd.items()})["sdf_pred"]
        assert torch.equal(both[b:b + 1], one)


@pytest.mark.parametrize("prec", PRECS)
def test_dense_grid_equals_explicit_grid_queries(prec):
    """s3d_decode_grid_fwd (in-kernel coordinates) == decode of make_3d_grid points, negated."""
    from oracle import ref_cpu
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "test", prec)
    fd = to_gpu(make_feed_dict(1, 64, 16, 12, seed=21, with_slices=False))
    code = model.encode(fd)
    nx = 20
    grid = model.decode_grid(code, nx)
    pts = ref_cpu.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3).unsqueeze(0).cuda()
    explicit = -model.decode_sdf(pts, code)
    assert grid.shape == (nx, nx, nx)
    assert (grid.reshape(1, -1) - explicit).abs().max() < 1e-5


@pytest.mark.parametrize("prec", PRECS)
def test_grid_slabs_tile_the_dense_grid_bit_exactly(prec):
    """s3d_decode_grid_slab_fwd: the query-parallel split of one object's grid — ragged slabs
    (shard_range of 3 and 7 ranks) concatenated equal the single-call grid bit for bit."""
    from slice3d_amd.parallel import shard_range
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "test", prec)
    fd = to_gpu(make_feed_dict(1, 64, 16, 12, seed=21, with_slices=False))
    code = model.encode(fd)
    nx = 37
    full = model.decode_grid(code, nx, box=1.1).reshape(-1)
    for world in (3, 7):
        parts = [model.decode_grid(code, nx, box=1.1, q_range=shard_range(nx ** 3, r, world)).clone()
                 for r in range(world)]
        assert torch.equal(torch.cat(parts), full), world
    with pytest.raises(ValueError):
        model.decode_grid(code, nx, q_range=(5, nx ** 3 + 1))
    with pytest.raises(ValueError):
        get_model(12, "train", prec).decode_grid(code, nx)       # the grid kernel is mode='test' only


def test_encode_decode_api_and_logits_sign():
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "test")
    fd = to_gpu(make_feed_dict(1, 32, 64, 12, seed=2, with_slices=False))
    c = model.encode(fd)
    d = model.decode(fd["qry_norot"], c)
    assert torch.equal(d.logits, -d.sdf)
    assert torch.equal(d.sdf, model(fd)["sdf_pred"])


def test_repack_after_weight_update():
    from slice3d_amd.synth import make_feed_dict
    from slice3d_amd.models import Slices3DRegModel
    from slice3d_amd.weights import load_seeded
    m = load_seeded(Slices3DRegModel(n_slices=12, mode="test"), 0).cuda().eval()
    fd = to_gpu(make_feed_dict(1, 32, 64, 12, seed=2, with_slices=False))
    a = m(fd)["sdf_pred"].clone()
    with torch.no_grad():
        m.fc_out[0].bias.add_(0.5)
    b = m(fd)["sdf_pred"]
    assert torch.allclose(b, a + 0.5, atol=1e-6)


def test_error_codes_are_loud():
    from slice3d_amd import _lib
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "test")
    bad = to_gpu(make_feed_dict(1, 40, 8, 12, with_slices=False))   # 40 is not a multiple of 16
    with pytest.raises(ValueError):
        model(bad)
    lib = _lib.load()
    rc = lib.s3d_unet_encode_fwd(None, None, None, None, 1, 64, 12, 0, None, 0, None)
    assert rc == -1 and b"null" in lib.s3d_last_error()


def test_generator3d_mise_and_dense_paths():
    """Generator3D (reconstruct.py:24-243): eval_points == -sdf; the MISE grid (resolution0=8, 2 upsampling
    steps -> 33^3) agrees with the dense grid wherever MISE actually evaluated a point; marching cubes of
    both give a closed, non-empty mesh in the unit cube."""
    from slice3d_amd.generator import Generator3D
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "test")
    fd = to_gpu(make_feed_dict(1, 64, 500, 12, seed=77, with_slices=False))
    gen = Generator3D(model, threshold=0.5, resolution0=8, upsampling_steps=2, chunk_size=3000, pred_type="sdf")
    vals = gen.eval_points(fd)
    assert torch.equal(vals, -model(fd)["sdf_pred"].squeeze(0))
    grid_mise = gen.generate_value_grid(fd)
    dense = Generator3D(model, resolution0=33, upsampling_steps=0, pred_type="sdf").generate_value_grid(fd)
    assert grid_mise.shape == dense.shape == (33, 33, 33)
    coarse = np.abs(grid_mise[::4, ::4, ::4] - dense[::4, ::4, ::4]).max()    # level-0 points are always evaluated
    assert coarse < 1e-4, coarse
    mesh, stats = gen.generate_mesh(fd)
    if len(mesh.faces):
        assert mesh.vertices.min() >= -0.5 - 1e-6 and mesh.vertices.max() <= 0.5 + 1e-6
        assert mesh.faces.max() < len(mesh.vertices)
    assert "time (eval points)" in stats and "time (marching cubes)" in stats


def test_f16x3_mode_is_fp32_class():
    """prec='f16x3' (convs, attention projections and FFN on 3 f16 MFMAs per product, operands split
    hi+lo) must meet the same 1e-4 gate against the oracle and sit within fp32 rounding of the f32 mode."""
    from oracle import ref_cpu
    from slice3d_amd.models import Slices3DRegModel
    from slice3d_amd.synth import make_feed_dict
    from slice3d_amd.weights import load_seeded
    m16 = load_seeded(Slices3DRegModel(n_slices=12, mode="test", prec="f16x3"), 0).cuda().eval()
    m32 = get_model(12, "test")
    sd = seeded_sd_from_shapes(_shapes(12))
    fd = make_feed_dict(1, 64, 3000, 12, seed=55, with_slices=False)
    a = m16(to_gpu(fd))["sdf_pred"].cpu()
    b = m32(to_gpu(fd))["sdf_pred"].cpu()
    ref = ref_cpu.forward(sd, fd, mode="test", n_slices=12, with_vgg=False)["sdf_pred"]
    print("f16x3 vs f32: %.3e   f16x3 vs oracle: %.3e   f32 vs oracle: %.3e" %
          (float((a - b).abs().max()), float((a - ref).abs().max()), float((b - ref).abs().max())))
    assert (a - ref).abs().max() < TOL
    assert (a - b).abs().max() < 5e-5      # two fp32-class evaluations (cf. oracle-vs-reference noise, synth.py)


def test_query_sort_is_a_deterministic_locality_permutation():
    """s3d_query_sort: a permutation of range(Q) per batch item, ordered by the Morton code of the projected
    256^2 bin with ascending query index inside a bin, identical on every call (the train step's dropout
    streams and row-wise comparisons rely on that)."""
    import ctypes as C
    from slice3d_amd import _lib
    from slice3d_amd.synth import make_feed_dict
    lib = _lib.load()
    b, q = 2, 30000
    fd = make_feed_dict(b, 32, q, 4, seed=3, device="cuda")
    qry = fd["qry_norot"].contiguous()
    # pile many queries onto a few pixels (large bins -> the workgroup rank-sort path) and beyond the clamp
    qry[0, :5000] = qry[0, 0]
    qry[1, 100:3000, :2] *= 40.0
    rot, trans = fd["obj_rot_mat"].contiguous(), fd["trans_mat_wo_rot_tp"].contiguous()
    nbytes = lib.s3d_query_sort_workspace_bytes(b, q)
    ws = torch.empty(nbytes // 4 + 1, dtype=torch.int32, device="cuda")
    perms = []
    for _ in range(3):
        perm = torch.empty(b, q, dtype=torch.int32, device="cuda")
        _lib.check(lib.s3d_query_sort(qry.data_ptr(), rot.data_ptr(), trans.data_ptr(), 0, b, q, perm.data_ptr(),
                                      ws.data_ptr(), nbytes, None), "s3d_query_sort")
        torch.cuda.synchronize()
        perms.append(perm.cpu().long())
    assert torch.equal(perms[0], perms[1]) and torch.equal(perms[0], perms[2])
    # reference keys computed with the module's own projection op
    pts = torch.bmm(qry, rot)
    g = torch.empty(b, q, 2, device="cuda")
    _lib.check(lib.s3d_project_coord_fwd(pts.contiguous().data_ptr(), trans.data_ptr(), g.data_ptr(), b, q, None), "proj")
    g = g.clamp(-1, 1).cpu()
    px = ((g[..., 0] + 1) * 127.5).clamp(0, 255).long()
    py = ((g[..., 1] + 1) * 127.5).clamp(0, 255).long()

    def spread(v):
        v = (v | (v << 4)) & 0x0F0F
        v = (v | (v << 2)) & 0x3333
        v = (v | (v << 1)) & 0x5555
        return v
    key = spread(px) | (spread(py) << 1)
    for bb in range(b):
        pm = perms[0][bb]
        assert torch.equal(torch.sort(pm).values, torch.arange(q))
        k = key[bb][pm]
        comp = k * q + pm                      # (bin, query index) must be strictly increasing
        # fp rounding of the projection can move a query across a bin edge; allow a handful of such ties
        bad = int((comp[1:] <= comp[:-1]).sum())
        assert bad <= 8, bad


@pytest.mark.parametrize("q", [700, 5000])
def test_sample_pyramid_matches_grid_sample(q):
    """s3d_sample_pyramid_fwd == the reference's five sample_from_planes calls + cat (models.py:63-73), with and
    without the internal locality order (q >= 4096)."""
    import torch.nn.functional as F
    from slice3d_amd.models import Slices3DRegModel, LEVEL_CHANNELS
    b, ns, s = 2, 3, 32
    g = torch.Generator().manual_seed(11)
    pyr = [torch.randn(b * ns, (s // 16) << l, (s // 16) << l, LEVEL_CHANNELS[l], generator=g) for l in range(5)]
    grid = torch.rand(b, q, 2, generator=g) * 2.4 - 1.2
    grid = grid.clamp(-1, 1)
    m = Slices3DRegModel(img_size=s, n_slices=ns, mode="test").cuda().eval()
    got = m.sample_pyramid([p.cuda() for p in pyr], grid.cuda()).cpu()
    gg = grid.view(b, 1, q, 2).expand(-1, ns, -1, -1).reshape(b * ns, 1, q, 2)
    want = torch.cat([F.grid_sample(p.permute(0, 3, 1, 2), gg, mode="bilinear", padding_mode="zeros",
                                    align_corners=True).permute(0, 3, 2, 1).reshape(b * ns, q, -1) for p in pyr], 2)
    assert got.shape == want.shape == (b * ns, q, 992)
    assert (got - want).abs().max() < 2e-5


@pytest.mark.parametrize("prec", PRECS)
def test_dense_256_cubed_grid_matches_oracle(prec):
    """BASELINE configs[3]: `reconstruct.py --mc_res0 256 --mc_up_steps 0` (reconstruct.py:135-146 with
    make_3d_grid, common.py:145-164) — the dense 256^3 grid (16 777 216 queries) of a 256^2 x 12-slice object
    through s3d_decode_grid_fwd (64 passes of 262 144 in-kernel coordinates), copied to the host like
    Generator3D does, and checked against the oracle on 4096 random grid indices plus the 8 corners, both
    sides of every pass boundary and 512 indices of the last pass."""
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "test", prec)
    fd = make_feed_dict(1, 256, 16, 12, seed=2024, with_slices=False)   # same image as the full-size test
    nx = 256
    code = model.encode(to_gpu(fd))
    grid = model.decode_grid(code, nx)
    torch.cuda.synchronize()
    g = grid.cpu()
    assert g.shape == (nx, nx, nx) and torch.isfinite(g).all()
    idx = _c4_indices(nx)
    idx_t = torch.from_numpy(idx)
    pts = _c4_points(idx, nx)
    ref = torch.from_numpy(_oracle_case("full256", _oc_full256)["c4_sdf"])
    err = float((g.reshape(-1)[idx_t] + ref.reshape(-1)).abs().max())      # grid holds logits = -sdf
    print("dense 256^3 (%s): max |grid + oracle sdf| over %d indices = %.3e" % (prec, len(idx), err))
    assert err < TOL, err
    # size-independent property at full size: the grid equals explicit decode_sdf calls on the same coordinates
    explicit = -model.decode_sdf(pts.cuda(), code)
    assert (g.reshape(-1)[idx_t] - explicit.cpu().reshape(-1)).abs().max() < 1e-5


@pytest.mark.parametrize("prec", PRECS)
def test_white_noise_images_within_the_reference_rounding_floor(prec):
    """SURVEY.md 8(d) specifies white-noise inputs, rng.uniform(-1,1).  With them the feature pyramid changes by
    O(its own size) between neighbouring pixels, so fp32 rounding of the projected coordinates alone moves sdf_pred
    by ~1e-4 in ANY fp32 evaluation, the reference included.  The gate is therefore stated against an fp64
    evaluation of the oracle:  max|hip - ref_fp64| <= max|ref_fp32 - ref_fp64| + eps  (eps = 5e-5, half the
    smooth-image tolerance), and the median error must be fp32-class (< 1e-5)."""
    from slice3d_amd.synth import make_feed_dict
    model = get_model(12, "test", prec)
    fd = make_feed_dict(1, 256, 6000, 12, seed=1234, smooth=False, with_slices=False)   # SURVEY 8(d) seed and size
    z = _oracle_case("white_s256_q6000", _oc_white)
    r32, r64 = torch.from_numpy(z["sdf32"]).double(), torch.from_numpy(z["sdf64"])
    out = model(to_gpu(fd))
    hip = out["sdf_pred"].cpu().double()
    e_hip = (hip - r64).abs()
    e_ref = (r32 - r64).abs()
    print("white noise (%s): max|hip-f64| %.3e  max|ref32-f64| %.3e  median %.3e / %.3e" %
          (prec, float(e_hip.max()), float(e_ref.max()), float(e_hip.median()), float(e_ref.median())))
    assert float(e_hip.max()) <= float(e_ref.max()) + 5e-5
    assert float(e_hip.median()) < 1e-5
    assert torch.isfinite(out["slices_rec"]).all()
    e_img = (_probe(out["slices_rec"].cpu().double()) - torch.from_numpy(z["rec64_probe"])).abs().max()
    assert float(e_img) <= float(z["e_img_ref"][0]) + 5e-5       # e_img_ref: the fp32 oracle's distance over ALL pixels


def test_single_pass_f16_throughput_mode_runs_and_reports_its_error():
    """prec='f16' (S3D_PREC_F16): operands rounded to f16, ONE MFMA per product — the "bf16" of BASELINE configs[1].
    Not fp32-class: it must NOT be expected to meet the 1e-4 gate; this test pins that it runs through every kernel
    (U-Net convs, latent build, attention, FFN, last-layer GEMMs), stays a sane approximation (5e-2 on sdf, 1e-2 on the
    slice images, for these weights; measured 1.1e-2 / 1.6e-3) and is measurably different from the split-precision mode."""
    from oracle import ref_cpu
    from slice3d_amd.synth import make_feed_dict
    m16 = get_model(12, "test", "f16")
    m3 = get_model(12, "test", "f16x3")
    sd = seeded_sd_from_shapes(_shapes(12))
    fd = make_feed_dict(1, 64, 6000, 12, seed=55, with_slices=False)
    a = m16(to_gpu(fd))
    b = m3(to_gpu(fd))
    ref = ref_cpu.forward(sd, fd, mode="test", n_slices=12, with_vgg=False)
    e_sdf = float((a["sdf_pred"].cpu() - ref["sdf_pred"]).abs().max())
    e_img = float((a["slices_rec"].cpu() - ref["slices_rec"]).abs().max())
    e3 = float((b["sdf_pred"].cpu() - ref["sdf_pred"]).abs().max())
    print("single-pass f16: max|sdf - oracle| %.3e (f16x3: %.3e), max|slices_rec - oracle| %.3e" % (e_sdf, e3, e_img))
    assert torch.isfinite(a["sdf_pred"]).all()
    assert e3 < TOL < e_sdf < 5e-2 and e_img < 1e-2


def test_bf16_decoder_mode_runs_and_reports_its_error():
    """prec='bf16' (S3D_PREC_BF16): the decoder's attention and FFN GEMMs on v_mfma_f32_16x16x32_bf16 with bf16-rounded
    operands (the precision BASELINE configs[1] literally names), everything else as the single-pass f16 mode.  A throughput
    mode further from fp32 than 'f16' (8 significand bits instead of 11): it must run through the kernels, stay a sane
    approximation (0.3 on sdf for these weights; the U-Net, which it does not touch, within the f16 mode's 1e-2) and sit
    further from the oracle than the f16 mode does."""
    from oracle import ref_cpu
    from slice3d_amd.synth import make_feed_dict
    mb = get_model(12, "test", "bf16")
    m16 = get_model(12, "test", "f16")
    sd = seeded_sd_from_shapes(_shapes(12))
    fd = make_feed_dict(1, 64, 6000, 12, seed=55, with_slices=False)
    a = mb(to_gpu(fd))
    b = m16(to_gpu(fd))
    ref = ref_cpu.forward(sd, fd, mode="test", n_slices=12, with_vgg=False)
    e_bf = float((a["sdf_pred"].cpu() - ref["sdf_pred"]).abs().max())
    e_16 = float((b["sdf_pred"].cpu() - ref["sdf_pred"]).abs().max())
    e_img = float((a["slices_rec"].cpu() - ref["slices_rec"]).abs().max())
    print("bf16 decoder mode: max|sdf - oracle| %.3e (single-pass f16: %.3e), max|slices_rec - oracle| %.3e" % (e_bf, e_16, e_img))
    assert torch.isfinite(a["sdf_pred"]).all()
    assert e_16 < e_bf < 0.3 and e_img < 1e-2
    # decode is deterministic and chunk-invariant in this mode too
    code = mb.encode(to_gpu(fd))
    full = mb.decode_sdf(fd["qry_norot"].cuda(), code)
    part = mb.decode_sdf(fd["qry_norot"][:, 1000:3000].contiguous().cuda(), code)
    assert torch.equal(full[:, 1000:3000], part)


def _sweep_cases(seed, n):
    """Deterministic pseudo-random shape sweep: image sizes that are multiples of 16, ragged query counts around the
    16-query group / 8-query attention item / chunk boundaries, every slice count the C ABI accepts, both prologues."""
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        ns = int(rng.integers(1, 13))
        s = int(rng.choice([16, 32, 48, 64, 80, 96]))
        b = int(rng.integers(1, 4))
        q = int(rng.choice([1, 7, 8, 9, 15, 16, 17, 31, 33, 127, 129, 255, 500, 1023, 2049, 4100]))
        mode = "test" if rng.random() < 0.5 else "train"
        cases.append((b, s, q, ns, mode))
    return cases


@pytest.mark.parametrize("prec", PRECS)
def test_shape_sweep_matches_oracle(prec):
    """24 shapes drawn from a fixed seed (new ones each round would make the suite a moving target): the C-ABI path against
    the oracle on every one of them, in both arithmetic modes."""
    from slice3d_amd.synth import make_feed_dict
    worst = 0.0
    n_cases = int(os.environ.get("S3D_SWEEP_N", str(SWEEP_N)))   # a longer one-off sweep: S3D_SWEEP_N=200 (profiles/r02_shape_sweep.md)
    z = _oracle_case("sweep24", _oc_sweep) if n_cases == SWEEP_N else None     # committed oracle outputs of the 24 shapes
    for i, (b, s, q, ns, mode) in enumerate(_sweep_cases(SWEEP_SEED, n_cases)):
        model = get_model(ns, mode, prec)
        fd = make_feed_dict(b, s, q, ns, seed=7000 + i, with_slices=False)
        if z is not None:
            ref_sdf, ref_rec = torch.from_numpy(z["sdf:%d" % i]), torch.from_numpy(z["rec:%d" % i])
        else:
            from oracle import ref_cpu
            key = ("sweep", i)
            if key not in _oracle_cache:
                _oracle_cache[key] = ref_cpu.forward(seeded_sd_from_shapes(_shapes(ns)), fd, mode=mode, n_slices=ns, with_vgg=False)
            ref_sdf, ref_rec = _oracle_cache[key]["sdf_pred"], _probe(_oracle_cache[key]["slices_rec"], i)
        out = model(to_gpu(fd))
        assert torch.isfinite(out["slices_rec"]).all()
        e_sdf = float((out["sdf_pred"].cpu() - ref_sdf).abs().max())
        e_img = float((_probe(out["slices_rec"].cpu(), i) - ref_rec).abs().max())
        assert e_sdf < TOL and e_img < TOL, ((b, s, q, ns, mode), e_sdf, e_img)
        worst = max(worst, e_sdf, e_img)
    print("shape sweep (%s): worst deviation %.2e over %d shapes" % (prec, worst, n_cases))
