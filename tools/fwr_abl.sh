# timing ablations of ffn_wgrad_rec_kernel (build/abl/lib_fwr_*.so, FWR_ABL_* macros): kernel-trace averages over 3 train steps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in "" $(ls build/abl/lib_fwr_*.so); do
  rm -rf /tmp/pa; (cd /tmp && S3D_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} rocprofv3 --kernel-trace -d /tmp/pa -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /dev/null 2>&1)
  echo "== ${lib:-base}"; python tools/rocpd_summary.py $(find /tmp/pa -name "*.db" | head -1) | grep -E "ffn_wgrad_rec_kernel|total kernel" | cut -c1-120
done
