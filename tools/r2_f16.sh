cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -s -k "single_pass or f16x3_mode" 2>&1 | grep -E "passed|failed|single-pass|assert|Error" | tail
python bench.py --cpu-sample 2048 --ldm-steps 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', r['value'], r['ms_per_step']); print(r['throughput_mode_f16'])"
