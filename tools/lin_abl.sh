# timing ablations of the streaming row-linear kernel (historical: the CV_ABL_* switches were deleted with the experiment, results in profiles/r03_lin_store_pattern.md): does the
# access pattern of the MFMA fragment layout (16-byte pieces at a 32-byte stride on the load side, 64-byte runs per row on the
# store side) hold the kernel below the rate of the full-line kernels?  Ablated variants give wrong results; only time is read.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in build/abl/lib_cv_LOADPERM.so build/abl/lib_cv_FULLLINE.so build/abl/lib_cv_LOADPERM_FULLLINE.so; do
  echo "=== ${lib:-product build}"
  export S3D_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib}
  python tools/bench_lin.py 2>&1 | tail -7
done
