# round 4, first GPU call: the new tests, then a marker trace of the train step and an LDM kernel trace (baselines)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_rccl.py tests/test_ldm.py tests/test_gpu_dataset.py -m gpu -x -q -s 2>&1 | tail -15 > gpurun_out/r04/t_small.log
python -m pytest tests/test_gpu_train.py -m gpu -x -q -s -k "full_size" 2>&1 | tail -30 > gpurun_out/r04/t_full.log
(cd /tmp && rm -rf /tmp/pm && rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/pm -o m -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pm.log 2>&1)
python tools/marker_summary.py /tmp/pm > gpurun_out/r04/r04_train_marker_trace.md 2>&1
ls -R /tmp/pm | head -20 >> gpurun_out/r04/r04_train_marker_trace.md
tail -3 /tmp/pm.log >> gpurun_out/r04/r04_train_marker_trace.md
(cd /tmp && rm -rf /tmp/pl && rocprofv3 --kernel-trace -d /tmp/pl -o l -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 --f16-steps 0 --mesh-steps 0 --pmc 0 --steps 1 --warmup 0 --n-qry 2048 --batch 1 --ldm-steps 20 > /tmp/pl.json 2>/dev/null)
python tools/rocpd_summary.py $(find /tmp/pl -name "*.db" | head -1) > gpurun_out/r04/r04_ldm_kernel_stats_before.md
python -c "
import json; r = json.loads(open('/tmp/pl.json').read().strip().splitlines()[-1]); print(r['ldm_denoise_step'])" >> gpurun_out/r04/r04_ldm_kernel_stats_before.md
cat gpurun_out/r04/t_small.log gpurun_out/r04/t_full.log; head -40 gpurun_out/r04/r04_train_marker_trace.md
