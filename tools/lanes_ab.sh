# Two decode lanes (attention of one half beside the FFN of the other, api.hip) against everything on one stream, same box:
#   ms per step, stage times (HIP events on each lane's stream: they overlap when lanes = 2), clock and power of both arms
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lanes in 1 2; do
  S3D_DECODE_LANES=$lanes python bench.py --cpu-sample 0 --f16-steps 0 --f32-steps 0 --c4-steps 0 --mesh-steps 0 --ldm-steps 0 --train-steps 0 --gt-train-steps 0 --pmc 0 --noise-steps 40 --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline())
w=r['white_noise']
print('lanes=$lanes  ms/step %.3f  q/s %.4g  stages %s  smooth(40 steps) %.3f ms %s MHz %s W' % (r['ms_per_step'], r['value'], {k: round(v,2) for k,v in r['stage_ms_per_step'].items()}, w['smooth']['ms_per_step'], w['smooth']['sclk_mhz'], w['smooth']['power_w']))
"
done
done
