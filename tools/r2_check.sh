cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r2_tests.txt
cat gpurun_out/r2_tests.txt
python bench.py > gpurun_out/bench_r2.json 2> gpurun_out/bench_r2.err; python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_r2.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step", "train_samples_per_s", "train_ms_per_step")})
print("roofline", {k: r["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "traffic")})
print("stages", {k: round(v, 2) for k, v in r["stage_ms_per_step"].items()})
print("c4", r["c4_dense_grid"]); print("ldm", r["ldm_denoise_step"]); print("gt", r["gt_train_step"])
print("sample", r["secondary_rooflines"][0]); print("unet", r["secondary_rooflines"][1]); print("attn", r["secondary_rooflines"][2]); print("mesh", r.get("mesh_extraction"))
cb = r["cpu_baseline"]; print("cpu", {k: cb[k] for k in ("value", "cores", "host_cores", "cpu_model", "stages", "thread_probe_s_per_256_queries")})
print("parity", r["parity_vs_oracle"])
PY
tail -3 gpurun_out/bench_r2.err
