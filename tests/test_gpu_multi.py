"""The first contact of this code with RCCL between DEVICES (reference: train.py:131-132 wraps the model in
nn.DataParallel — here one process per GPU over RCCL/xGMI).

Every test skips unless the box has >= 2 GPUs (the builder's and the driver's test boxes have one), so the suite costs
nothing there; on the first multi-GPU lease they run the SAME workers tests/test_gpu_ddp.py runs with two gloo ranks on one
GPU, with the `nccl` backend and one rank per device:
  * data-parallel training against the real reference's per-shard gradients (g6 goldens), bucketed exchange on the side
    stream and the flat one; --sync_bn against the reference's full-batch step (g5 goldens);
  * the dense 256^3 reconstruction grid split into slabs + one all_gather, bit-equal to one rank doing everything;
  * `bench.py --gpus 2` (the driver's scaling command at N = 2): contract line with n_gpus == 2 and per-rank times."""
import json
import os
import subprocess
import sys

import pytest
import torch

from test_gpu_ddp import _recon_worker, _spawn, _train_worker

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (one RCCL rank per device)")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("overlap", [False, True])
def test_rccl_ddp_two_devices_matches_mean_of_reference_shard_gradients(overlap):
    msgs = _spawn(_train_worker, "g6_ddp_shards_s32_n12_q160_b2", False, overlap, "f32", "nccl")
    assert all(m.startswith("ok") for m in msgs), msgs


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_rccl_sync_bn_two_devices_reproduces_the_reference_full_batch(prec):
    msgs = _spawn(_train_worker, "g5_train_s32_n12_q128_b2", True, True, prec, "nccl")
    assert all(m.startswith("ok") for m in msgs), msgs


def test_rccl_sharded_256_cubed_grid_equals_single_rank():
    """BASELINE configs[3] over two devices: contiguous slabs of the 256^3 grid's linear index, one all_gather of the logits
    (67 MB), bit-identical to a single rank; plus a MISE run (one all_gather per refinement round)."""
    assert _spawn(_recon_worker, "nccl", ((256, 256, 0), (64, 16, 2))) == 1.0


@pytest.mark.parametrize("n", [2] + ([torch.cuda.device_count()] if torch.cuda.device_count() > 2 else []))
def test_bench_runs_one_rank_per_device_over_rccl(n):
    env = dict(os.environ)
    for k in ("S3D_BENCH_BACKEND", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--batch", "1",
                        "--n-qry", "20000", "--img-size", "128", "--cpu-sample", "0", "--train-steps", "2", "--c4-steps", "1",
                        "--c4-res", "64", "--ldm-steps", "2", "--gt-train-steps", "0", "--f16-steps", "0", "--mesh-steps", "0",
                        "--f32-steps", "0", "--noise-steps", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == n and res["scaling"] == "weak"
    assert abs(res["value"] - n * 20000 / (res["ms_per_step"] * 1e-3)) < 1e-6 * res["value"]
    lo, hi = res["ms_per_step_rank_min_max"]
    assert 0 < lo <= hi
    assert res["c4_dense_grid"]["n_gpus"] == n and res["ldm_denoise_step"]["n_gpus"] == n and res["train_samples_per_s"] > 0
