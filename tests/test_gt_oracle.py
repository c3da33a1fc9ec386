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
