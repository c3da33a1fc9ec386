# SQ counter pass over the training step kernels (tools/prof_train.py, B=1 C2 shape); per-kernel averages
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/psq; (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/psq -o s -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /dev/null 2>&1)
rm -rf /tmp/psq2; (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/psq2 -o s -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /dev/null 2>&1)
python - <<'PY'
import csv, glob, collections
for d in ('/tmp/psq', '/tmp/psq2'):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print('no csv in', d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if any(s in k for s in ('ffn_wgrad_rec_kernel', 'ffn_bwd_dx', 'ffn_layer_f16x3', 'wgrad_lin', 'attn_core', 'sample_bwd', 'conv_igemm_f16x3_kernel<4, 4, 2, 2, 1>')):
            acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, dd in sorted(acc.items()):
        print(k)
        for c, v in dd.items():
            print('   %-28s %.4g (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
