cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ldm.py -q -x -m gpu > gpurun_out/r05_pytest6.log 2>&1; tail -4 gpurun_out/r05_pytest6.log
python tools/ldm_ab.py 40 > gpurun_out/r05_ldm_ab2.log 2>&1; cat gpurun_out/r05_ldm_ab2.log
bash tools/train_ab.sh build/abl/lib_noregen.so > gpurun_out/r05_train_ab.log 2>&1; cat gpurun_out/r05_train_ab.log
