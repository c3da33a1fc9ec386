# kernel trace + roctx marker trace of the training step (tools/prof_train.py: 8 steps of 4 objects x 100 k queries)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r04
rm -rf /tmp/pt; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1)
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/r04/r04_train_f16x3_kernel_stats.md
tail -1 /tmp/pt.log >> gpurun_out/r04/r04_train_f16x3_kernel_stats.md
(cd /tmp && rm -rf /tmp/pm && rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/pm -o m -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pm.log 2>&1)
python tools/marker_summary.py /tmp/pm > gpurun_out/r04/r04_train_marker_trace.md 2>&1
ls /tmp/pm >> gpurun_out/r04/r04_train_marker_trace.md
head -45 gpurun_out/r04/r04_train_f16x3_kernel_stats.md | cut -c1-150; tail -2 gpurun_out/r04/r04_train_f16x3_kernel_stats.md; head -40 gpurun_out/r04/r04_train_marker_trace.md
python -m pytest tests/test_gpu_rccl.py -m gpu -x -q -s 2>&1 | tail -8
