# experiment libraries: tools/build_experiment.sh tools/patches/ffn_fp8_two_product_experiments.patch <out.so> -DFFN_G1_FP8=1 / -DFFN_G2_TWO=1|2
# EXPERIMENT: GEMM2 of the inference FFN with two of the three split products
# (build/abl/lib_ffn_two1.so drops W2_lo*h_hi, lib_ffn_two2.so drops W2_hi*h_lo): parity on the 200-shape sweep, the
# full-size and white-noise tests, then FFN time / clock / power.  Output -> profiles/r03_ffn_two_product_gemm2.md
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in "" build/abl/lib_ffn_two1.so build/abl/lib_ffn_two2.so; do
  echo "=== ${lib:-base (three products)}"
  export S3D_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib}
  [ -z "$lib" ] && unset S3D_HIP_LIB
  S3D_SWEEP_N=${SWEEP_N:-200} python -m pytest tests/test_gpu_parity.py -m gpu -s -q -k "f16x3 and (shape_sweep or full_size_256 or white_noise or golden)" 2>&1 | grep -E "passed|failed|worst|max|Error|assert" | cut -c1-200 | tail -12
  python tools/ffn_data_power.py 60 2>/dev/null | grep -E "seeded random \(the bench\)|^\| decoder"
done
