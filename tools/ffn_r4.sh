# EXPERIMENT: the pipelined FFN kernel with PIPE_R = 4 (64 rows per wave, one workgroup per CU, 451
# VGPRs: every weight byte is read from LDS 4x per CU instead of 8x) — time, clock and socket power next to the product build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in "" build/abl/lib_ffn_r4.so; do
  echo "=== ${lib:-product build (PIPE_R = 2, two workgroups per CU)}"
  export S3D_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib}
  [ -z "$lib" ] && unset S3D_HIP_LIB
  python -m pytest tests/test_gpu_parity.py -m gpu -q -k "f16x3 and (full_size_256 or golden)" 2>&1 | tail -1
  python tools/ffn_data_power.py 80 2>/dev/null | grep -E "seeded random \(the bench\)|all zero|^\| decoder"
done
