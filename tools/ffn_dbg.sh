cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--cpu-sample 2048 --ldm-steps 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 --steps 10 --warmup 3"
for lib in "" gpurun_in_r4.so ""  gpurun_in_r4.so; do
  S3D_HIP_LIB=${lib:+$PWD/$lib} python bench.py $B 2>gpurun_out/ab.err | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('LIB=$lib', 'ms/step %.2f' % r['ms_per_step'], 'ffn ms/launch %.3f' % r['roofline']['avg_launch_ms'], 'parity %.2e' % r['parity_vs_oracle']['max_abs_err'], {k: round(v, 2) for k, v in r['stage_ms_per_step'].items()})
"
done
