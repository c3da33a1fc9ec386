cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06h; mkdir -p $O
rm -rf /tmp/pt; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1)
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > $O/r06_train_f16x3_kernel_stats.md; grep "wall time" /tmp/pt.log >> $O/r06_train_f16x3_kernel_stats.md
rm -rf /tmp/pt2; (cd /tmp && S3D_PREC=f16 rocprofv3 --kernel-trace -d /tmp/pt2 -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt2.log 2>&1)
python tools/rocpd_summary.py $(find /tmp/pt2 -name "*.db" | head -1) > $O/r06_train_f16_kernel_stats.md; grep "wall time" /tmp/pt2.log >> $O/r06_train_f16_kernel_stats.md
tail -2 $O/r06_train_f16x3_kernel_stats.md; tail -2 $O/r06_train_f16_kernel_stats.md; grep -n "sample_bwd\|sbd_" $O/r06_train_f16x3_kernel_stats.md | cut -c1-140
