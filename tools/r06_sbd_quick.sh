# quick loop for train_sbd.hip: A/B against the atomic kernels at two sizes + kernel trace of the training step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-sbdq}; mkdir -p gpurun_out/$TAG
python tools/dbg_sbd.py 2 64 8192 12 f32 2>&1 | tail -7
python tools/dbg_sbd.py 4 256 100000 12 f16x3 2>&1 | tail -10
rm -rf /tmp/pt; (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1)
tail -1 /tmp/pt.log | tee gpurun_out/$TAG/train_wall.txt
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/$TAG/train_kernel_stats.md
grep -n "sample_bwd\|sbd_\|sbt_project" gpurun_out/$TAG/train_kernel_stats.md | cut -c1-200
