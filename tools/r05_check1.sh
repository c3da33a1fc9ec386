# round 5, check 1: training goldens on the box's host CPU; LDM fused-GroupNorm tests + timing; training tests + step time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/golden
S3D_ORACLE_GOLDEN_DIR=gpurun_out/golden python tests/golden/make_oracle_golden.py train_128_16k smooth_b1_s128_q16384_n12 train_full_b4_s256 > gpurun_out/r05_golden_box.log 2>&1
cp gpurun_out/golden/oracle_*.npz tests/golden/
python -m pytest tests/test_ldm.py tests/test_gpu_train.py -q -x -m gpu --durations=10 > gpurun_out/r05_pytest4.log 2>&1
tail -25 gpurun_out/r05_pytest4.log
python bench.py --cpu-sample 0 --f16-steps 0 --f32-steps 0 --noise-steps 0 --c4-steps 0 --mesh-steps 0 --gt-train-steps 0 --pmc 0 --steps 5 --warmup 2 > gpurun_out/r05_bench_b.json 2> gpurun_out/r05_bench_b.err
tail -c 900 gpurun_out/r05_bench_b.json
