cd $GRAFT_REPO_ROOT
run() { n=0; for i in 1 2 3 4 5 6 7 8 9 10; do python -m pytest "tests/test_gpu_gt.py::test_gt_matches_oracle" -q -m gpu 2>&1 | tail -1 | grep -q failed && n=$((n+1)); done; echo "$1: $n failures of 10"; }
run default
S3D_ATTN_Q=0 run attn_q_off
S3D_ATTN_LAST=0 run attn_last_off
