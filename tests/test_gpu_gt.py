"""GPU parity of the Slices3DGTModel path (s3d_gt_* through the C ABI) against the reference goldens and the
CPU oracle.  Tolerance: 1e-4 absolute on sdf (the north_star gate), as for Slices3DRegModel."""
import numpy as np
import pytest
import torch

from helpers import load_golden
from test_gt_oracle import GT_CASES, gt_feed, gt_shapes

pytestmark = pytest.mark.gpu


def make_model(ns, prec, mode):
    from slice3d_amd.models_gt import Slices3DGTModel
    from slice3d_amd.weights import load_seeded
    return load_seeded(Slices3DGTModel(n_slices=ns, mode=mode, prec=prec), 0).cuda().eval()


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("case", GT_CASES)
def test_gt_matches_reference_golden(case, prec):
    g = load_golden(case)
    m = make_model(g["n_slices"], prec, g["mode"])
    fd = gt_feed(g, "cuda")
    code = m.encode(fd)
    for l in range(5):   # raw pyramid probes (reference tensors are NCHW)
        n, c, h, w = [int(v) for v in g["pyr%d_shape" % l]]
        nchw = code.pyramid[l].permute(0, 3, 1, 2).contiguous().reshape(-1).cpu().numpy()
        got = nchw[g["pyr%d_idx" % l]]
        scale = max(1.0, float(np.abs(g["pyr%d_val" % l]).max()))
        assert np.abs(got - g["pyr%d_val" % l]).max() < 1e-4 * scale, l
    out = m(fd)["sdf_pred"].cpu().numpy()
    assert np.abs(out - g["sdf_pred"]).max() < 1e-4


@pytest.mark.parametrize("b,s,q,ns,mode", [(1, 32, 70, 12, "train"), (2, 48, 4500, 5, "test")])
def test_gt_matches_oracle(b, s, q, ns, mode):
    """Other shapes (ragged query count, fewer slices, the locality-sorted path at q >= 4096) vs the oracle."""
    from oracle import ref_cpu
    from helpers import seeded_sd_from_shapes
    from slice3d_amd.synth import make_feed_dict
    fd = make_feed_dict(b, s, q, ns, seed=900 + q)
    sd = seeded_sd_from_shapes(gt_shapes())
    with torch.no_grad():
        want, _ = ref_cpu.gt_forward(sd, fd, mode, ns)
    m = make_model(ns, "f16x3", mode)
    got = m({k: v.cuda() for k, v in fd.items()})["sdf_pred"].cpu()
    assert (got - want).abs().max() < 1e-4


def test_gt_dense_grid_equals_point_decode():
    from slice3d_amd.synth import make_feed_dict
    ns, s, nx = 12, 32, 9
    fd = {k: v.cuda() for k, v in make_feed_dict(1, s, 8, ns, seed=5).items()}
    m = make_model(ns, "f16x3", "test")
    code = m.encode(fd)
    logits = m.decode_grid(code, nx, box=1.0, trans_mat_wo_rot_tp=fd["trans_mat_wo_rot_tp"])
    lin = torch.linspace(-0.5, 0.5, nx, device="cuda")
    pts = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(1, -1, 3)
    sdf = m.decode_sdf(pts, code, trans_mat_wo_rot_tp=fd["trans_mat_wo_rot_tp"], mode="test")
    assert logits.shape == (nx, nx, nx)
    assert (logits.reshape(-1) + sdf.reshape(-1)).abs().max() < 2e-5


def test_gt_decode_is_bit_reproducible_run_to_run():
    """Regression test of a race in the LDS-DMA weight rings (a wave could read a chunk another wave's DMA had not
    delivered yet): it showed as rare wrong rows on this shape — every workgroup of the attention kernel handles a
    single item, so nothing delays its first fragment reads.  Twenty decodes must agree bit for bit."""
    from slice3d_amd.synth import make_feed_dict
    fd = {k: v.cuda() for k, v in make_feed_dict(2, 48, 4500, 5, seed=5400).items()}
    m = make_model(5, "f16x3", "test")
    first = m(fd)["sdf_pred"].clone()
    for _ in range(19):
        assert torch.equal(m(fd)["sdf_pred"], first)


_gt_white = {}


def _oc_gt_white():
    from oracle import ref_cpu
    from helpers import seeded_sd_from_shapes
    from slice3d_amd.synth import make_feed_dict
    fd = make_feed_dict(1, 128, 5000, 12, seed=1235, smooth=False)
    sd = seeded_sd_from_shapes(gt_shapes())
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        r32, _ = ref_cpu.gt_forward(sd, fd, "test", 12)
        r64, _ = ref_cpu.gt_forward(sd64, {k: v.double() for k, v in fd.items()}, "test", 12)
    return {"sdf32": r32.numpy(), "sdf64": r64.numpy()}


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_gt_white_noise_images_within_the_reference_rounding_floor(prec):
    """White-noise slice images: the bound is stated against an fp64 evaluation of the oracle,
    max|hip - ref_fp64| <= max|ref_fp32 - ref_fp64| + 5e-5, median error fp32-class (see the twin test of
    Slices3DRegModel in test_gpu_parity.py).  Oracle outputs: committed (tests/golden/oracle_gt_white_s128_q5000.npz,
    written from _oc_gt_white by tests/golden/make_oracle_golden.py; S3D_LIVE_ORACLE=1 recomputes)."""
    from helpers import load_oracle_golden
    from slice3d_amd.synth import make_feed_dict
    fd = make_feed_dict(1, 128, 5000, 12, seed=1235, smooth=False)
    if "z" not in _gt_white:
        z = load_oracle_golden("gt_white_s128_q5000")
        _gt_white["z"] = _oc_gt_white() if z is None else z
    r32, r64 = torch.from_numpy(_gt_white["z"]["sdf32"]), torch.from_numpy(_gt_white["z"]["sdf64"])
    m = make_model(12, prec, "test")
    hip = m({k: v.cuda() for k, v in fd.items()})["sdf_pred"].cpu().double()
    e_hip, e_ref = (hip - r64).abs(), (r32.double() - r64).abs()
    print("GT white noise (%s): max|hip-f64| %.3e  max|ref32-f64| %.3e  median %.3e / %.3e" %
          (prec, float(e_hip.max()), float(e_ref.max()), float(e_hip.median()), float(e_ref.median())))
    assert float(e_hip.max()) <= float(e_ref.max()) + 5e-5
    assert float(e_hip.median()) < 1e-5
