cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
rm -rf /tmp/pt; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/r01_train_f16x3_kernel_stats.md
python tools/c4_dense_grid.py 2>&1 | tail -4
