# SQ counter pass over the decoder kernels (tools/time_attn.py workload); prints per-kernel averages
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/psq; (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/psq -o s -- python $GRAFT_REPO_ROOT/tools/time_attn.py > /dev/null 2>&1)
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/psq/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'ffn_layer' in k or 'attn_layer' in k:
        acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print('   %-28s %.4g (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
