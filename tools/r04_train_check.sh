# training-path check: every train test (both models, DDP, RCCL single rank), then the step time at the bench shape
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_train.py tests/test_gpu_gt_train.py tests/test_gpu_ddp.py tests/test_gpu_rccl.py -m gpu -x -q -s ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | grep -vE "^\s*$" | tail -${TAILN:-40} > gpurun_out/r04/t_train.log
cat gpurun_out/r04/t_train.log
python tools/prof_train.py 2>&1 | tail -2
