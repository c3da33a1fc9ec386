# experiment libraries: tools/build_experiment.sh tools/patches/ffn_fp8_two_product_experiments.patch <out.so> -DFFN_G1_FP8=1 / -DFFN_G2_TWO=1|2
# A/B of the inference FFN's first GEMM: cross terms on the scaled fp8 MFMA (product build) vs three f16 products
# (build/abl/lib_g1_f16.so = the same sources with -DFFN_G1_FP8=0): parity tests, stage times, error against the oracle
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in "" build/abl/lib_g1_f16.so; do
  echo "=== ${lib:-product build (fp8 cross terms in GEMM1)}"
  export S3D_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib}
  [ -z "$lib" ] && unset S3D_HIP_LIB
  python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f16x3" 2>&1 | tail -2
  python bench.py --train-steps 0 --pmc 0 --cpu-sample 0 --ldm-steps 0 --gt-train-steps 0 --c4-steps 0 --f16-steps 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('q/s %.0f  ms/step %.2f  ffn %.2f  parity %.3e  frac %.3f' % (d['value'], d['ms_per_step'], d['stage_ms_per_step']['ffn_layer'], d['parity_vs_oracle']['max_abs_err'], d['roofline']['frac']))"
done
