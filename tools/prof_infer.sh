# kernel trace of the inference bench (no CPU baseline / train legs) -> gpurun_out/r02_bench_f16x3_kernel_stats.md
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/pi; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pi -o b -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --train-steps 0 --gt-train-steps 0 --ldm-steps 0 --c4-steps 0 --steps 10 --warmup 2 > /tmp/pi_bench.json 2>/dev/null)
python tools/rocpd_summary.py $(find /tmp/pi -name "*.db" | head -1) > gpurun_out/r02_bench_f16x3_kernel_stats.md
tail -c 600 /tmp/pi_bench.json >> gpurun_out/r02_bench_f16x3_kernel_stats.md
head -40 gpurun_out/r02_bench_f16x3_kernel_stats.md | cut -c1-170
