cd $GRAFT_REPO_ROOT
S3D_SWEEP_N=200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k shape_sweep -s 2>&1 | grep -E "shape sweep|passed|failed" > gpurun_out/r05_shape_sweep.log; cat gpurun_out/r05_shape_sweep.log
S3D_LIVE_ORACLE=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gt.py tests/test_gpu_train.py -q -m gpu -s -k "full_size or white_noise or shape_sweep or dense_256 or smooth_output or 128_16k" 2>&1 | grep -E "passed|failed|full-size train|smooth-gradient|white noise|dense 256|shape sweep|train 128" > gpurun_out/r05_live_oracle.log; cat gpurun_out/r05_live_oracle.log
