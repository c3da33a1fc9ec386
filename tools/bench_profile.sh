cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python bench.py > gpurun_out/bench_f16x3.json 2> gpurun_out/bench_f16x3.err
tail -c 1500 gpurun_out/bench_f16x3.json
(cd /tmp && rm -rf /tmp/p1 && rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --train-steps 0 --gt-train-steps 0 --ldm-steps 0 > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) > gpurun_out/r01_bench_f16x3_kernel_stats.md
(cd /tmp && rm -rf /tmp/p2 && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o f -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --train-steps 0 --gt-train-steps 0 --ldm-steps 0 --steps 3 --warmup 1 > /dev/null 2>&1)
(cd /tmp && rm -rf /tmp/p3 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o w -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --train-steps 0 --gt-train-steps 0 --ldm-steps 0 --steps 3 --warmup 1 > /dev/null 2>&1)
cp $(find /tmp/p2 -name "*counter_collection.csv" | head -1) gpurun_out/pmc_fetch.csv
cp $(find /tmp/p3 -name "*counter_collection.csv" | head -1) gpurun_out/pmc_write.csv
ls -la gpurun_out/ | head
