"""GPU tests of the Slices3DGTModel training step (s3d_gt_train_fwd_bwd / s3d_adam_step through the C ABI) against
(1) goldens captured from the REAL reference in train mode (dropout pinned to 0; tests/golden/make_golden_gt_train.py)
and (2) autograd through the CPU oracle (gt_forward_train) on other shapes, with and without dropout.
Tolerances as in test_gpu_train.py: losses 2e-5 relative, sdf 1e-4 absolute, gradients 2e-2 relative L2 per tensor
(the L1 loss makes d loss / d sdf = sign(.)/n; typical error 1e-4)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, check_grads_against_golden
from test_gt_oracle import GT_PRE_BN_BIASES, gt_train_sd

pytestmark = pytest.mark.gpu

FEED_KEYS = ("img_slices", "qry_norot", "sdf", "obj_rot_mat", "trans_mat_wo_rot_tp")


def make_trainer(ns, **kw):
    from slice3d_amd.models_gt import Slices3DGTModel
    from slice3d_amd.trainer import HipGtTrainer
    from slice3d_amd.weights import load_seeded
    m = load_seeded(Slices3DGTModel(n_slices=ns, mode="train"), 0).cuda()
    return m, HipGtTrainer(m, **kw)


@pytest.mark.parametrize("case", ["gt4_train_s32_n12_q130_b2", "gt3_train_s128_n12_q96_b1"])
def test_gt_train_step_matches_reference_golden(case):
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    b, s, q, ns, _ = [int(v) for v in z["meta"]]
    m, tr = make_trainer(ns)
    batch = {k: torch.from_numpy(z[k]).cuda() for k in FEED_KEYS}
    losses, sdf_pred = tr.forward_backward(batch, want_outputs=True)
    torch.cuda.synchronize()
    got = losses.cpu().numpy().astype(np.float64)
    assert np.abs(sdf_pred.cpu().numpy() - z["sdf_pred"]).max() < 1e-4
    assert abs(got[0] - z["losses"][0]) < 2e-5 * z["losses"][0]
    assert abs(got[1] - z["losses"][1]) < 1e-6
    sd = dict(m.named_parameters())
    # the tensors that get a gradient are exactly the reference's
    assert set(tr.names) == {str(k) for k in z["grad_names"]}
    worst = check_grads_against_golden(z, {k: sd[k].grad.reshape(-1).cpu().numpy() for k in tr.names},
                                       skip=GT_PRE_BN_BIASES)
    state = m.state_dict()
    for key in z.files:
        if key.startswith("bn:"):
            if s != 128 and ".conv_last." in key:
                continue   # golden made with the feat_global branch skipped (see make_golden_gt_train.py)
            want = z[key]
            assert np.abs(state[key[3:]].cpu().numpy() - want).max() < 1e-5 * max(1.0, float(np.abs(want).max())), key
    print("worst sampled-gradient error / max|g| = %.2e" % worst)


def _compare_with_oracle(m, tr, sd, got_losses, loss, acc):
    assert abs(float(got_losses[0]) - float(loss.detach())) < 2e-5 * abs(float(loss.detach())) + 1e-7
    assert abs(float(got_losses[1]) - float(acc)) < 1e-6
    for k, p in m.named_parameters():
        if k not in tr.offsets:
            assert sd[k].grad is None, k
            continue
        if k in GT_PRE_BN_BIASES:
            assert float(p.grad.abs().max()) < 1e-4
            continue
        ref = sd[k].grad
        rel = float((p.grad.cpu() - ref).norm() / ref.norm())
        assert rel < 2e-2, (k, rel)


# q >= 4096 exercises the locality-sorted token order and the tiled sampling backward
@pytest.mark.parametrize("b,s,q,ns", [(1, 32, 50, 12), (2, 48, 33, 4), (2, 32, 4200, 3)])
def test_gt_train_grads_match_oracle_autograd(b, s, q, ns):
    from oracle import ref_cpu
    from slice3d_amd.synth import make_feed_dict
    m, tr = make_trainer(ns)
    fd = make_feed_dict(b, s, q, ns, seed=400 + q)
    sd = gt_train_sd()
    loss, acc, sdf, ts = ref_cpu.gt_forward_train(sd, fd, ns, 0.0)
    loss.backward()
    losses, sdf_pred = tr.forward_backward({k: v.cuda() for k, v in fd.items()}, want_outputs=True)
    assert (sdf_pred.cpu() - sdf.detach()).abs().max() < 1e-4
    _compare_with_oracle(m, tr, sd, losses.cpu().numpy(), loss, acc)


def test_gt_dropout_matches_oracle_with_identical_masks():
    """Dropout 0.1 with the oracle fed the very masks the kernels draw (s3d_dropout_mask)."""
    from oracle import ref_cpu
    from slice3d_amd.synth import make_feed_dict
    from test_gpu_train import _hip_dropout_masks
    b, s, q, ns, p = 1, 32, 40, 12, 0.1
    m, tr = make_trainer(ns, dropout=p)
    fd = make_feed_dict(b, s, q, ns, seed=77)
    got = tr.forward_backward({k: v.cuda() for k, v in fd.items()}).cpu().numpy().copy()
    masks = _hip_dropout_masks(b, q, ns, p, tr.last_seed)
    sd = gt_train_sd()
    loss, acc, sdf, ts = ref_cpu.gt_forward_train(sd, fd, ns, p, masks=masks)
    loss.backward()
    _compare_with_oracle(m, tr, sd, got, loss, acc)
    tr.dropout = 0.0
    l0 = tr.forward_backward({k: v.cuda() for k, v in fd.items()}).cpu().numpy()
    assert abs(l0[0] - got[0]) > 1e-4   # the masks are really applied


# 20000 queries: loss gradient 5e-5 per element, below f16's normal range (see backward_scale in api_train.inc)
@pytest.mark.parametrize("q,p_drop", [(200, 0.1), (20000, 0.0)])
def test_gt_f16x3_training_matches_fp32(q, p_drop):
    from slice3d_amd.synth import make_feed_dict
    fd = {k: v.cuda() for k, v in make_feed_dict(1, 64, q, 12, seed=78).items()}
    res = {}
    for prec in ("f32", "f16x3"):
        m, tr = make_trainer(12, prec=prec, dropout=p_drop, seed=5)
        losses = tr.forward_backward(fd).cpu().numpy().copy()
        res[prec] = (losses, tr.grad_flat.cpu().clone(), tr)
    la, lb = res["f32"][0], res["f16x3"][0]
    assert abs(la[0] - lb[0]) < 2e-5 * abs(la[0])
    ga, gb, tr = res["f32"][1], res["f16x3"][1], res["f32"][2]
    assert float((ga - gb).norm() / ga.norm()) < 2e-2
    gmax = max(float(ga[tr.offsets[k]:tr.offsets[k] + p.numel()].norm()) for k, p in zip(tr.names, tr.params))
    for k, p in zip(tr.names, tr.params):
        if k in GT_PRE_BN_BIASES:
            continue
        off, n = tr.offsets[k], p.numel()
        a, bb = ga[off:off + n], gb[off:off + n]
        assert float((a - bb).norm()) < 5e-2 * float(a.norm()) + 1e-4 * gmax, k


def test_gt_train_steps_reduce_the_loss_and_keep_dead_tensors():
    """A few Adam steps on one batch: the loss goes down; tensors the reference's optimiser never touches
    (classifier, conv_last BN affine, fc_global, the att_layer twin) stay bit-identical."""
    from slice3d_amd.synth import make_feed_dict
    m, tr = make_trainer(12, lr=1e-3)
    fd = {k: v.cuda() for k, v in make_feed_dict(2, 32, 256, 12, seed=79).items()}
    dead = {k: v.detach().clone() for k, v in m.named_parameters() if k not in tr.offsets}
    assert len(dead) == 20   # 118 parameter tensors, 98 of them trained (as in the reference)
    first = tr.train_step(fd)
    for _ in range(5):
        last = tr.train_step(fd)
    assert last[0] < first[0]
    for k, v in m.named_parameters():
        if k in dead:
            assert torch.equal(v, dead[k]), k
    # eval-mode inference picks up the new parameters and running statistics
    m.eval()
    out = m(fd)["sdf_pred"]
    assert torch.isfinite(out).all()
