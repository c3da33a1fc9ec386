cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ldm.py tests/test_gpu_train.py tests/test_gpu_parity.py tests/test_gpu_gt.py tests/test_gpu_gt_train.py -q -x -m gpu --durations=12 > gpurun_out/r05_pytest5.log 2>&1
tail -22 gpurun_out/r05_pytest5.log
python bench.py --cpu-sample 0 --f16-steps 0 --f32-steps 0 --noise-steps 0 --c4-steps 0 --mesh-steps 0 --gt-train-steps 0 --pmc 0 --steps 20 --warmup 5 > gpurun_out/r05_bench_b.json 2> gpurun_out/r05_bench_b.err
tail -c 1000 gpurun_out/r05_bench_b.json; tail -3 gpurun_out/r05_bench_b.err
