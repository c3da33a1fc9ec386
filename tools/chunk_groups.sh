# decode pass size: the previous product (16384 groups per pass), one pass with one FFN launch (32768), the product (32768 with FFN launches of 16384)
# variant libraries: copies of the tree built with `make EXTRA=-DS3D_CHUNK_GROUPS=...` (lib_cg32768: one FFN launch per pass) and the previous commit's build (lib_base)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--cpu-sample 0 --ldm-steps 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 --f16-steps 0 --mesh-steps 0 --pmc 0 --steps 20 --warmup 3"
for r in 1 2 3; do
for lib in build/abl/lib_base.so build/abl/lib_cg32768.so ""; do
  S3D_HIP_LIB=${lib:+$PWD/$lib} python bench.py $B 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s' % '${lib:-product}', 'qps %.3fM  ms/step %.2f' % (r['value'] / 1e6, r['ms_per_step']), {k: round(v, 2) for k, v in r['stage_ms_per_step'].items() if v > 0}, 'frac', round(r['roofline']['frac'], 4), 'launches', r['roofline']['launches'], 'parity', r.get('parity_vs_oracle', {}).get('max_abs_err'))"
done; done
