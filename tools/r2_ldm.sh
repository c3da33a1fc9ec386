cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_ldm.py -q -x -m gpu 2>&1 | grep -E "passed|failed|assert|Error" | tail -6
python bench.py --cpu-sample 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 --f16-steps 0 --mesh-steps 0 --steps 1 --warmup 0 --n-qry 2048 --batch 1 --ldm-steps 20 2>/dev/null | python -c "
import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ldm_denoise_step'])"
