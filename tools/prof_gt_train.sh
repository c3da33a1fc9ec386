cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_entrypoints.py -x -q -m gpu 2>&1 | tail -3
python tools/time_gt_train.py 2>&1 | tail -5
(cd /tmp && rm -rf /tmp/pg && S3D_GT_STEPS=3 rocprofv3 --kernel-trace -d /tmp/pg -o g -- python $GRAFT_REPO_ROOT/tools/time_gt_train.py > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pg -name "*.db" | head -1) > gpurun_out/r01_gt_train_kernel_stats.md
head -40 gpurun_out/r01_gt_train_kernel_stats.md
