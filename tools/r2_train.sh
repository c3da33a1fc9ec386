cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_train.py tests/test_gpu_gt_train.py tests/test_gpu_entrypoints.py -q -x -m gpu -s 2>&1 | grep -E "passed|failed|Error|error|assert|worst|autograd" | tail -25
