cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/pt; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/r02_train_f16x3_kernel_stats.md
head -46 gpurun_out/r02_train_f16x3_kernel_stats.md | cut -c1-150; tail -2 gpurun_out/r02_train_f16x3_kernel_stats.md
