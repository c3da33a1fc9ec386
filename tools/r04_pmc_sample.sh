# cache counters of sample_tokens_kernel (is the gather L1-hit, L2-hit or fabric traffic?)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04
BENCH="$GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --train-steps 0 --gt-train-steps 0 --ldm-steps 0 --c4-steps 0 --f16-steps 0 --mesh-steps 0 --pmc 0 --steps 2 --warmup 1"
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
rm -rf /tmp/pc; (cd /tmp && rocprofv3 --pmc $set --output-format csv -d /tmp/pc -o c -- python $BENCH > /dev/null 2>&1)
python - <<'PY'
import csv, glob, collections
fs = glob.glob('/tmp/pc/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])) if fs else []:
    k = r['Kernel_Name']
    if 'sample_tokens' in k or 'attn_layer_q' in k:
        acc[k[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, dd in acc.items():
    print(k, {c: '%.4g' % (sum(v) / len(v)) for c, v in dd.items()}, len(next(iter(dd.values()))))
PY
done 2>&1 | tee gpurun_out/r04/pmc_sample_tokens.txt
