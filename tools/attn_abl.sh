# timing ablations of attn_layer_q_kernel (build/abl/lib_*.so built by tools/build_experiment.sh from tools/patches/attn_q_ablations.patch with -DAQ_ABL_NOBAR etc.): stage time of the bench workload
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--cpu-sample 0 --ldm-steps 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 --f16-steps 0 --mesh-steps 0 --steps 10 --warmup 3"
for lib in "" $(ls build/abl/lib_*.so); do
  S3D_HIP_LIB=${lib:+$PWD/$lib} python bench.py $B 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s' % '${lib:-base}', 'ms/step %.2f' % r['ms_per_step'], {k: round(v, 2) for k, v in r['stage_ms_per_step'].items() if k in ('attn_layer', 'ffn_layer', 'sample_tokens')})"
done
