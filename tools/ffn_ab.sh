# A/B of the decoder FFN kernels on the bench workload + parity of the pipelined one (tools/ffn_ab.sh)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
B="--cpu-sample 2048 --ldm-steps 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 --steps 10 --warmup 3"
for pipe in 1 0 1; do
  S3D_FFN_PIPE=$pipe python bench.py $B 2>gpurun_out/ab.err | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PIPE=$pipe', 'qps %.3e' % r['value'], 'ms/step %.2f' % r['ms_per_step'], 'ffn ms/launch %.3f' % r['roofline']['avg_launch_ms'], 'frac %.4f' % r['roofline']['frac'], 'parity %.2e' % r['parity_vs_oracle']['max_abs_err'], {k: round(v, 2) for k, v in r['stage_ms_per_step'].items()})
"
done
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "f16x3" 2>&1 | tail -3
