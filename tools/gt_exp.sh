cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for x in 0 1 2; do echo "TILE $x"; 
S3D_CONV_TILE=$x python tools/time_train_batch.py 2>&1 | grep "ms/step" | head -2
done
