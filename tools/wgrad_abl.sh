# timing ablations of wgrad_lin_f16x3_kernel (build/abl/lib_wl_*.so: tools/build_experiment.sh tools/patches/wgrad_lin_ablations.patch ... -DWL_ABL_NOLOAD): kernel time of the training
# step's in_proj weight gradient with the global loads / the MFMAs removed.  Wrong results; only the time is read.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in "" build/abl/lib_wl_NOLOAD.so build/abl/lib_wl_NOMFMA.so; do
  echo "=== ${lib:-product build}"
  export S3D_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib}
  [ -z "$lib" ] && unset S3D_HIP_LIB
  rm -rf /tmp/pt; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1)
  python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) | grep -E "wgrad_lin|total kernel" | cut -c1-140
done
