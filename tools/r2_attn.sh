cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_gt.py -q -x -m gpu -k "f16x3 or gt" 2>&1 | tail -3
python bench.py --cpu-sample 2048 --ldm-steps 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 --f16-steps 0 --mesh-steps 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('qps %.3e' % r['value'], 'ms/step %.2f' % r['ms_per_step'], 'parity %.2e' % r['parity_vs_oracle']['max_abs_err'], {k: round(v, 2) for k, v in r['stage_ms_per_step'].items()})"
