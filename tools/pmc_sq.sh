# SQ counter passes over the decoder kernels (tools/time_attn.py workload); prints per-kernel averages
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/psq; (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/psq -o s -- python $GRAFT_REPO_ROOT/tools/time_attn.py > /dev/null 2>&1)
rm -rf /tmp/psq2; (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --output-format csv -d /tmp/psq2 -o s -- python $GRAFT_REPO_ROOT/tools/time_attn.py > /dev/null 2>&1)
python - <<'PY'
import csv, glob, collections
for d in ('/tmp/psq', '/tmp/psq2'):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print('no csv in', d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if 'ffn_layer' in k or 'attn_layer' in k or 'sample_tokens' in k:
            acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, dd in sorted(acc.items()):
        print(k)
        for c, v in dd.items():
            print('   %-28s %.4g (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
