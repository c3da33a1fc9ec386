# final run of round 6 (session 4): full -m gpu suite, smoke, default bench line, kernel traces of the inference bench and of the LDM step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06final; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1 > $O/build.log
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/r06_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> $O/r06_pytest_gpu.log
timeout 600 python bench.py > $O/r06_bench_line.json 2> $O/bench.err
rm -rf /tmp/pi; (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/pi -o b -- python $GRAFT_REPO_ROOT/bench.py --infer-only --cpu-sample 0 --steps 10 --warmup 2 > /tmp/pi_bench.json 2>/dev/null)
python tools/rocpd_summary.py $(find /tmp/pi -name "*.db" | head -1) > $O/r06_bench_f16x3_kernel_stats.md
echo >> $O/r06_bench_f16x3_kernel_stats.md; echo "bench line of the traced run:" >> $O/r06_bench_f16x3_kernel_stats.md; tail -c 3000 /tmp/pi_bench.json | head -c 700 >> $O/r06_bench_f16x3_kernel_stats.md
rm -rf /tmp/pl; (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/pl -o l -- python $GRAFT_REPO_ROOT/tools/time_ldm.py 1 64 > /tmp/pl.txt 2>/dev/null)
python tools/rocpd_summary.py $(find /tmp/pl -name "*.db" | head -1) > $O/r06_ldm_b1_kernel_stats.md
grep "LDM denoise" /tmp/pl.txt >> $O/r06_ldm_b1_kernel_stats.md
timeout 300 python tools/ldm_layers.py 1 64 2>/dev/null > $O/r06_ldm_layers.md
for a in "1 64" "4 64" "1 128"; do timeout 200 python tools/time_ldm.py $a | grep replay; done > $O/ldm_times.txt 2>/dev/null
cat $O/r06_pytest_gpu.log; tail -c 1200 $O/r06_bench_line.json; cat $O/ldm_times.txt
