cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
S3D_HIP_LIB=$PWD/build/abl/dbg_T.so python bench.py --cpu-sample 0 --ldm-steps 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 --f16-steps 0 --steps 2 --warmup 1 2>/dev/null | grep "^EPI" | tail -8
