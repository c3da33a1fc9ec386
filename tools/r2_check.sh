cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x -s 2>&1 | grep -E "passed|failed|error|Error|white noise|dense 256|train 128|worst|assert|FAILED" | tail -40 > gpurun_out/r2_tests.txt
tail -30 gpurun_out/r2_tests.txt
python bench.py > gpurun_out/bench_r2.json 2> gpurun_out/bench_r2.err; tail -c 3000 gpurun_out/bench_r2.json; tail -5 gpurun_out/bench_r2.err
