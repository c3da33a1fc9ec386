"""RCCL on ONE rank (reference: train.py:131-132 wraps the model in DataParallel — here one process per GPU).

No multi-GPU node is available to these tests, so the transport itself cannot be exercised between devices.  What CAN
run on the single test GPU is everything that sits between this code and RCCL: librccl loads, a communicator forms with
`device_id=`, the bucketed gradient all-reduces go through the event-ordered side stream (hipEvents recorded INSIDE
s3d_train_fwd_bwd), and the sync-BN callback issues its collectives from inside the library call.  S3D_FORCE_COLLECTIVES=1
(test switch, slice3d_amd/trainer.py) keeps the trainer from skipping the exchange at world size 1; a sum over one rank is
the identity, so the gradients must not change (bit for bit where the step itself is bit-reproducible).  Each case runs in its own process (one process group per process)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


_TRAIN_SCRIPT = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
from slice3d_amd.models import Slices3DRegModel
from slice3d_amd.synth import make_feed_dict
from slice3d_amd.trainer import HipTrainer
from slice3d_amd.weights import load_seeded

calls = {"n": 0, "elems": 0}
_orig = dist.all_reduce
def counting(t, *a, **k):
    calls["n"] += 1
    calls["elems"] += t.numel()
    return _orig(t, *a, **k)
dist.all_reduce = counting

batch = {k: v.cuda() for k, v in make_feed_dict(2, 32, 300, 12, seed=8).items()}
def grads(force, overlap, sync_bn):
    if force:
        os.environ["S3D_FORCE_COLLECTIVES"] = "1"
    else:
        os.environ.pop("S3D_FORCE_COLLECTIVES", None)
    m = load_seeded(Slices3DRegModel(n_slices=12, mode="train"), 0).cuda()
    tr = HipTrainer(m, prec="f16x3", overlap_all_reduce=overlap, sync_bn=sync_bn, dropout=0.1, seed=3)
    calls["n"] = calls["elems"] = 0
    losses = tr.forward_backward(batch)
    tr.all_reduce_grads()
    torch.cuda.synchronize()
    bn = m.state_dict()["slices_generator.up1.conv.double_conv.1.running_var"].clone()
    det = torch.cat([tr.grad_flat[tr.offsets[k]:tr.offsets[k] + p.numel()] for k, p in zip(tr.names, tr.params)
                     if k.startswith(("att_decoder.", "fc_out.", "fc_p."))])     # no float atomics upstream of these
    return tr.grad_flat.clone(), losses.clone(), bn, dict(calls), tr.grad_flat.numel(), det.clone()

g0, l0, bn0, c0, n, d0 = grads(False, True, False)
assert c0["n"] == 0                                  # world size 1, no switch: nothing is exchanged
gr, lr, _, _, _, dr = grads(False, True, False)      # run-to-run noise floor of the step itself (float atomics in the sampling backward)
g1, l1, bn1, c1, _, d1 = grads(True, True, False)    # four buckets behind their events on the side stream
g2, l2, bn2, c2, _, d2 = grads(True, False, False)   # one flat all-reduce
g3, l3, bn3, c3, _, d3 = grads(True, True, True)     # + the sync-BN callback's collectives from inside the library call
rel = lambda a, b: float((a - b).norm() / b.norm())
out = {
    "repeat_rel": rel(gr, g0), "repeat_det_equal": bool(torch.equal(dr, d0)),
    "bucketed_det_equal": bool(torch.equal(d0, d1)) and bool(torch.equal(l0, l1)), "bucketed_rel": rel(g1, g0),
    "bucketed_calls": c1["n"], "bucketed_elems": c1["elems"], "n_grad": n,
    "flat_det_equal": bool(torch.equal(d0, d2)), "flat_rel": rel(g2, g0), "flat_calls": c2["n"],
    "syncbn_calls": c3["n"],
    "syncbn_rel": float((g3 - g0).norm() / g0.norm()),
    "syncbn_loss_rel": float(((l3 - l0).abs() / l0.abs().clamp_min(1e-12)).max()),
    "syncbn_stat_abs": float((bn3 - bn0).abs().max()),
    "rccl_version": str(torch.cuda.nccl.version()),
}
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def test_single_rank_rccl_carries_the_bucketed_all_reduce_and_the_sync_bn_callback():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("S3D_FORCE_COLLECTIVES", None)
    r = subprocess.run([sys.executable, "-c", _TRAIN_SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    print(out)
    # a sum over one rank is the identity: the tensors no float atomic sits upstream of (transformer, fc_out, fc_p) are
    # bit-identical with and without the exchange; the rest (the sampling backward flushes tile sums with float atomics)
    # agrees to the step's own run-to-run noise
    assert out["repeat_det_equal"] and out["bucketed_det_equal"] and out["flat_det_equal"], out
    tol = max(10 * out["repeat_rel"], 1e-6)
    assert out["bucketed_rel"] <= tol and out["flat_rel"] <= tol, out
    assert out["bucketed_calls"] == 4 and out["bucketed_elems"] == out["n_grad"]      # the four buckets tile grad_flat
    assert out["flat_calls"] == 1
    # 20 train-mode BatchNorm layers (12 encoder + 8 decoder): two collectives each in the forward (means, merged variances), one in
    # the backward, + the 4 gradient buckets
    assert out["syncbn_calls"] == 60 + 4, out
    # one-rank sync-BN is the same batch statistics through the count-merged formulas: fp32 rounding apart
    assert out["syncbn_rel"] < 1e-4 and out["syncbn_loss_rel"] < 1e-5 and out["syncbn_stat_abs"] < 1e-5, out


def test_bench_prints_its_contract_as_one_rccl_rank_under_the_launcher():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` — the driver's multi-GPU command shape at
    N = 1: the rank forms an RCCL process group, barriers and the MAX-over-ranks reduction run, one JSON line comes out."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--img-size", "64", "--n-qry", "4096", "--batch", "1", "--cpu-sample", "0", "--f16-steps", "0",
           "--c4-steps", "0", "--mesh-steps", "0", "--ldm-steps", "0", "--train-steps", "1", "--gt-train-steps", "0",
           "--pmc", "0", "--f32-steps", "0", "--noise-steps", "0"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", S3D_FORCE_COLLECTIVES="1", OMP_NUM_THREADS="8")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0
    assert d["train_samples_per_s"] > 0
