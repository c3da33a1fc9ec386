cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python tools/time_gt_train.py 2>&1 | tail -3
python bench.py > gpurun_out/bench_f16x3.json 2> gpurun_out/bench_f16x3.err; tail -c 2500 gpurun_out/bench_f16x3.json
