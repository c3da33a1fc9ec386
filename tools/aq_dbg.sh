cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python bench.py --cpu-sample 0 --ldm-steps 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 --f16-steps 0 --steps 5 --warmup 2 2>gpurun_out/b.err | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r['value']); print(r['secondary_rooflines'][2]); print(r['mesh_extraction'])"
tail -3 gpurun_out/b.err
