cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/time_gt_train.py 2>&1 | tail -3
python tools/time_train_batch.py 2>&1 | grep "ms/step"
S3D_WGRAD3_WGS=2048 python tools/time_train_batch.py 2>&1 | grep "ms/step"
