# timing ablations of the TRAINING forward FFN (ffn_layer_f16x3_pipe_kernel<2|3>): build/abl/lib_ff_*.so from
# tools/patches/ffn_train_fwd_ablations.patch (no activity bits / no operand images / no U store).  Wrong gradients; only the time is read.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for dr in 0.1 0; do
for lib in "" build/abl/lib_ff_NOM.so build/abl/lib_ff_NOIMG.so build/abl/lib_ff_NOU.so build/abl/lib_ff_ALL.so; do
  echo "=== dropout $dr ${lib:-product build}"
  export S3D_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib}
  [ -z "$lib" ] && unset S3D_HIP_LIB
  rm -rf /tmp/pt; (cd /tmp && S3D_DROPOUT=$dr timeout 600 rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1)
  python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) | grep -E "pipe_kernelILi[234]|total kernel" | cut -c1-150
done; done
