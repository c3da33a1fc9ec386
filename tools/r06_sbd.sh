# round 6: the atomic-free sampling backward (train_sbd.hip): training parity tests, the slow path forced, kernel trace of the step
#   gpurun -- 'bash tools/r06_sbd.sh <tag> [pytest -k expression]'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-sbd}; KEXPR=${2:-""}
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q ${KEXPR:+-k "$KEXPR"} 2>&1 | tail -8 | tee gpurun_out/$TAG/pytest.log
S3D_SBD_SLOW_MOD=5 timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "grads_match_oracle" 2>&1 | tail -4 | tee gpurun_out/$TAG/pytest_slow.log
rm -rf /tmp/pt; (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1)
tail -2 /tmp/pt.log | tee gpurun_out/$TAG/train_wall.txt
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/$TAG/train_kernel_stats.md
grep -n "sample_bwd\|sbd_\|sbt_project\|fillBuffer" gpurun_out/$TAG/train_kernel_stats.md | cut -c1-220
tail -2 gpurun_out/$TAG/train_kernel_stats.md
