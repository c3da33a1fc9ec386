cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_train.py -q -x -m gpu 2>&1 | tail -2
python bench.py --cpu-sample 0 --ldm-steps 0 --gt-train-steps 3 --c4-steps 0 --f16-steps 0 --mesh-steps 0 --steps 2 --warmup 1 --train-steps 4 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('train samples/s %.2f  ms/step %.1f' % (r['train_samples_per_s'], r['train_ms_per_step']), 'gt', r['gt_train_step']['ms_per_step'])"
