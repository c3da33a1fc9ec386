# kernel trace of the LDM denoise step (bench.py's ldm leg, 20 steps + 2 warm-up) -> gpurun_out/r02/r02_ldm_kernel_stats.md
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
(cd /tmp && rm -rf /tmp/pl && rocprofv3 --kernel-trace -d /tmp/pl -o l -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --train-steps 0 --gt-train-steps 0 --c4-steps 0 --f16-steps 0 --steps 1 --warmup 0 --n-qry 2048 --batch 1 --ldm-steps 20 > /tmp/pl.json 2>/dev/null)
python tools/rocpd_summary.py $(find /tmp/pl -name "*.db" | head -1) > gpurun_out/r02/r02_ldm_kernel_stats.md
python -c "
import json; r = json.loads(open('/tmp/pl.json').read().strip().splitlines()[-1]); print(r['ldm_denoise_step'])" >> gpurun_out/r02/r02_ldm_kernel_stats.md
head -32 gpurun_out/r02/r02_ldm_kernel_stats.md | cut -c1-170
