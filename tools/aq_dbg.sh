cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_forward_matches_oracle" 2>&1 | tail -2
