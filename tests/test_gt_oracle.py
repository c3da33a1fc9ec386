"""CPU tests of the Slices3DGTModel row: the oracle restatement (oracle/ref_cpu.py gt_*) against
golden tensors captured from the REAL reference (tests/golden/make_golden_gt.py), and the host module's
state_dict contract."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_golden, seeded_sd_from_shapes

GT_CASES = ("gt1_s128_n12_q300_test", "gt2_s64_n12_q200_b2_train")


def gt_shapes():
    return {k: tuple(v) for k, v in json.load(open(os.path.join(GOLDEN, "state_dict_keys_gt.json"))).items()}


def gt_feed(g, device="cpu"):
    return {k: torch.from_numpy(g[k]).to(device) for k in
            ("img_slices", "qry_norot", "obj_rot_mat", "trans_mat_wo_rot_tp")}


@pytest.mark.parametrize("case", GT_CASES)
def test_gt_oracle_matches_reference_golden(case):
    from oracle import ref_cpu
    g = load_golden(case)
    sd = seeded_sd_from_shapes(gt_shapes())
    with torch.no_grad():
        sdf, feats = ref_cpu.gt_forward(sd, gt_feed(g), g["mode"], g["n_slices"])
    assert np.abs(sdf.numpy() - g["sdf_pred"]).max() < 1e-4
    for l, f in enumerate(feats):
        assert tuple(f.shape) == tuple(g["pyr%d_shape" % l])
        got = f.reshape(-1)[torch.from_numpy(g["pyr%d_idx" % l])].numpy()
        assert np.abs(got - g["pyr%d_val" % l]).max() < 2e-5 * max(1.0, float(np.abs(g["pyr%d_val" % l]).max()))


def test_gt_module_state_dict_contract():
    """Same keys and shapes as the reference module (released checkpoints load with strict=True)."""
    from slice3d_amd.models_gt import Slices3DGTModel
    m = Slices3DGTModel(backend="none")
    want = gt_shapes()
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want


def test_gt_module_refuses_to_compute_without_the_library():
    from slice3d_amd import _lib
    from slice3d_amd.models_gt import Slices3DGTModel
    m = Slices3DGTModel(backend="none").eval()
    with pytest.raises(_lib.S3dError):
        m.encode({"img_slices": torch.zeros(1, 36, 32, 32)})


# conv biases directly in front of a train-mode BatchNorm: exact gradient 0, both sides produce rounding noise
GT_PRE_BN_BIASES = {"img_encoder.%s.bias" % k for k in
                    ("conv1_2.0", "conv2_2.7", "conv3_3.14", "conv3_3.17", "conv4_3.24", "conv4_3.27", "conv5_3.34",
                     "conv5_3.37")}


def gt_train_sd():
    sd = seeded_sd_from_shapes(gt_shapes())
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    return sd


@pytest.mark.parametrize("case", ["gt4_train_s32_n12_q130_b2", "gt3_train_s128_n12_q96_b1"])
def test_gt_train_oracle_matches_reference_golden(case):
    """Train-mode restatement (batch-stat BN, loss, autograd gradients, running-stat updates) against one
    forward/backward of the real reference (tests/golden/make_golden_gt_train.py)."""
    from helpers import check_grads_against_golden
    from oracle import ref_cpu
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    b, s, q, ns, _ = [int(v) for v in z["meta"]]
    fd = {k: torch.from_numpy(z[k]) for k in ("img_slices", "qry_norot", "sdf", "obj_rot_mat", "trans_mat_wo_rot_tp")}
    sd = gt_train_sd()
    loss, acc, sdf, ts = ref_cpu.gt_forward_train(sd, fd, ns, 0.0)
    loss.backward()
    assert np.abs(sdf.detach().numpy() - z["sdf_pred"]).max() < 1e-4
    assert abs(float(loss.detach()) - z["losses"][0]) < 2e-5 * z["losses"][0]
    assert abs(float(acc) - z["losses"][1]) < 1e-6
    names = {str(k) for k in z["grad_names"]}
    assert names == {k for k, v in sd.items() if v.grad is not None}
    check_grads_against_golden(z, {k: sd[k].grad.reshape(-1).numpy() for k in names}, skip=GT_PRE_BN_BIASES)
    for key in z.files:
        if key.startswith("bn:"):
            k = key[3:]
            if s != 128 and ".conv_last." in k:
                continue   # that golden was made with the feat_global branch skipped (classifier needs 128^2)
            want = z[key]
            got = ts.new_stats[k].numpy() if k in ts.new_stats else sd[k].numpy()
            assert np.abs(got - want).max() < 1e-5 * max(1.0, float(np.abs(want).max())), key
