# A/B: FFN pipe kernel with 4 waves per workgroup (two weight streams per CU) vs 8 (one) on the bench workload
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--cpu-sample 2048 --ldm-steps 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 --f16-steps 0 --mesh-steps 0 --steps 10 --warmup 3"
for lib in "" build/abl/lib_w8.so "" build/abl/lib_w8.so; do
  S3D_HIP_LIB=${lib:+$PWD/$lib} python bench.py $B 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s' % '${lib:-base(4 waves)}', 'qps %.3e' % r['value'], 'ffn ms/launch %.3f' % r['roofline']['avg_launch_ms'], 'frac %.4f' % r['roofline']['frac'], 'parity %.2e' % r['parity_vs_oracle']['max_abs_err'], {k: round(v, 2) for k, v in r['stage_ms_per_step'].items() if k in ('attn_layer','ffn_layer','ffn_final')})"
done
