cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_gt_train.py tests/test_gpu_train.py tests/test_gpu_entrypoints.py -x -q -m gpu 2>&1 | tail -3
python tools/time_gt_train.py 2>&1 | tail -3
python tools/time_train_batch.py 2>&1 | grep "ms/step"
