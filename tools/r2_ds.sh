cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_dataset.py -q -x -m gpu 2>&1 | tail -15
