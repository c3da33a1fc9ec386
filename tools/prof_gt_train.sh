cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_gt_train.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3
python tools/time_gt_train.py 2>&1 | tail -3
python tools/time_train_batch.py 2>&1 | grep "ms/step"
(cd /tmp && rm -rf /tmp/pg && S3D_GT_STEPS=3 rocprofv3 --kernel-trace -d /tmp/pg -o g -- python $GRAFT_REPO_ROOT/tools/time_gt_train.py > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pg -name "*.db" | head -1) > gpurun_out/r01_gt_train_kernel_stats.md
head -12 gpurun_out/r01_gt_train_kernel_stats.md | cut -c1-150
