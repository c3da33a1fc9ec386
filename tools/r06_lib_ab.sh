# inference loop with alternative builds of the library (S3D_HIP_LIB), alternating on one box:
#   gpurun -- 'bash tools/r06_lib_ab.sh 2 product build/abl/lib_x.so ...'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    if [ "$L" = product ]; then unset S3D_HIP_LIB; else export S3D_HIP_LIB=$GRAFT_REPO_ROOT/$L; fi
    python bench.py --infer-only --cpu-sample 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('%-28s ms/step %.3f  attn %.3f ffn %.3f final %.3f sample %.3f unet %.3f | ffn launch %.4f ms' % ('$L', d['ms_per_step'], s['attn_layer'], s['ffn_layer'], s['ffn_final'], s['sample_tokens'], s['unet_encode'], d['roofline']['avg_launch_ms']))"
  done
done
