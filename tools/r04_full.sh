# full GPU suite, the default bench line, and the train-step kernel trace (dropout 0.1 and 0)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04/t_all.log
cat gpurun_out/r04/t_all.log
python bench.py > gpurun_out/r04/bench.json 2> gpurun_out/r04/bench.err; tail -c 3000 gpurun_out/r04/bench.json
for dp in 0.1 0; do
rm -rf /tmp/pt; (cd /tmp && S3D_DROPOUT=$dp rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1)
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/r04/train_stats_dp$dp.md
tail -1 /tmp/pt.log >> gpurun_out/r04/train_stats_dp$dp.md
head -22 gpurun_out/r04/train_stats_dp$dp.md | cut -c1-140; tail -2 gpurun_out/r04/train_stats_dp$dp.md
done
