cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in build/abl/dbg_*.so; do echo $lib; S3D_HIP_LIB=$PWD/$lib python -m pytest tests/test_gpu_parity.py -q -x -s -m gpu -k "golden and f16x3 and g1_c1" 2>&1 | grep -E "^lane|passed|failed" | head -70; done
