# A/B of an environment switch on the inference loop, alternating arms on one box:
#   gpurun -- 'bash tools/r06_ab.sh S3D_LAST_OVERLAP 3'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
VAR=$1; N=${2:-3}
for i in $(seq $N); do
  for v in 0 1; do
    env $VAR=$v python bench.py --infer-only --cpu-sample 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('$VAR=$v  ms/step %.3f  attn %.3f ffn %.3f final %.3f sample %.3f' % (d['ms_per_step'], s['attn_layer'], s['ffn_layer'], s['ffn_final'], s['sample_tokens']))"
  done
done
