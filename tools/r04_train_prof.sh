# training tests, then the step's kernel trace at the bench shape (top kernels + total)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_train.py tests/test_gpu_gt_train.py -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | grep -vE "^\s*$" | tail -${TAILN:-15}
rm -rf /tmp/pt; (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1); tail -1 /tmp/pt.log
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/r04/train_stats_now.md; head -${TOPN:-16} gpurun_out/r04/train_stats_now.md | cut -c1-150; grep "total kernel" gpurun_out/r04/train_stats_now.md
