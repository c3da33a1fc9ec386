# train step (bench.py's train leg, 4 objects x 100 k queries, dropout 0.1) with the product library against a variant, same box,
# alternating:  tools/train_ab.sh build/abl/lib_noregen.so
cd $GRAFT_REPO_ROOT
VAR=$1
for rep in 1 2 3; do
for lib in "" $VAR; do
  S3D_HIP_LIB=${lib:+$PWD/$lib} python bench.py --cpu-sample 0 --f16-steps 0 --f32-steps 0 --noise-steps 0 --c4-steps 0 --mesh-steps 0 --ldm-steps 0 --gt-train-steps 0 --pmc 0 --steps 1 --warmup 1 --train-steps 12 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('%-28s train %.2f ms/step' % ('${lib:-product}', r['train_ms_per_step']))
"
done
done
