# round-3 training check: parity tests of the train step, then a kernel trace of 3 steps at the bench shape
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_train.py -x -q -m gpu -s ${R3_K:+-k "$R3_K"} 2>&1 | grep -v Warning | tail -${R3_TAIL:-30}
rm -rf /tmp/pt; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1); tail -3 /tmp/pt.log
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/r03_train_f16x3_kernel_stats.md
head -${R3_HEAD:-30} gpurun_out/r03_train_f16x3_kernel_stats.md | cut -c1-150; tail -2 gpurun_out/r03_train_f16x3_kernel_stats.md
