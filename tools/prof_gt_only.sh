cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/pg && S3D_GT_STEPS=3 rocprofv3 --kernel-trace -d /tmp/pg -o g -- python $GRAFT_REPO_ROOT/tools/time_gt_train.py > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pg -name "*.db" | head -1) > gpurun_out/r02_gt_train_kernel_stats.md
head -45 gpurun_out/r02_gt_train_kernel_stats.md | cut -c1-170
