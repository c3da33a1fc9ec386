cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04
rm -rf /tmp/pt; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1)
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/r04/train_stats_latest.md
tail -1 /tmp/pt.log >> gpurun_out/r04/train_stats_latest.md
head -${HEADN:-16} gpurun_out/r04/train_stats_latest.md | cut -c1-150; tail -2 gpurun_out/r04/train_stats_latest.md
