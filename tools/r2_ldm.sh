cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_ldm.py -q -x -m gpu -s 2>&1 | grep -E "passed|failed|LDM 128|assert|Error" | tail -6
for v in 1 0; do S3D_LDM_ATTN_F16X3=$v python bench.py --cpu-sample 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 --f16-steps 0 --steps 1 --warmup 0 --n-qry 2048 --batch 1 --ldm-steps 30 2>/dev/null | python -c "
import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('attn f16x3=$v', r['ldm_denoise_step'])"; done
