cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_entrypoints.py -q -x -m gpu -k bench 2>&1 | tail -12
