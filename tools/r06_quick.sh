# round 6 quick loop on the GPU box: selected parity tests + the inference legs of the bench (no train / LDM / CPU legs)
#   gpurun -- 'bash tools/r06_quick.sh <tag> "<pytest -k expression>"'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-q}; KEXPR=${2:-"fused_last or golden or run_to_run"}
mkdir -p gpurun_out/$TAG
python -m pytest tests/test_gpu_parity.py -x -q -k "$KEXPR" 2>&1 | tail -5 | tee gpurun_out/$TAG/pytest.log
python bench.py --infer-only --cpu-sample 0 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python - <<P
import json
d = json.loads(open('gpurun_out/$TAG/bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
print(d['stage_ms_per_step'])
print('roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'parity', d.get('parity_vs_oracle'))
P
tail -3 gpurun_out/$TAG/bench.err
# kernel trace of the inference loop (TRACE=1)
if [ "${TRACE:-1}" = "1" ]; then
  (cd /tmp && rm -rf /tmp/p1 && rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python $GRAFT_REPO_ROOT/bench.py --infer-only --cpu-sample 0 --steps 6 --warmup 2 > /dev/null 2>&1)
  python tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) > gpurun_out/$TAG/kernel_stats.md
  head -14 gpurun_out/$TAG/kernel_stats.md | cut -c1-200
fi
