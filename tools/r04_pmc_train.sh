# SQ counter pass over the training step's large kernels (tools/prof_train.py: 8 steps of 4 objects x 100 k queries): matrix-pipe occupancy per launch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04
rm -rf /tmp/psq; (cd /tmp && timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d /tmp/psq -o s -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /dev/null 2>&1)
python - > gpurun_out/r04/r04_train_sq_counters.md <<'PY'
import csv, glob, collections
print("# SQ counters of the training step's large kernels, round 4 (rocprofv3 --pmc, tools/prof_train.py: 4 objects x 100 k queries, f16x3, dropout 0.1)\n")
print("Per-launch averages over the launches of the 8 steps.  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).\n")
fs = glob.glob('/tmp/psq/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])) if fs else []:
    k = r['Kernel_Name']
    if any(s in k for s in ('ffn_wgrad_rec_kernel', 'ffn_layer_f16x3', 'wgrad_lin', 'attn_layer_q', 'attn_bwd_q', 'conv3x3_lds_f16x3_kernel<4', 'lin_stream', 'sample_bwd', 'attn_mix0')):
        acc[k[:64]][r['Counter_Name']].append(float(r['Counter_Value']))
print("| kernel | launches | MFMA pipe busy % | VALU instructions per MFMA-busy cycle x 16 | LDS-wait share of wave cycles % |\n|---|---|---|---|---|")
for k, dd in sorted(acc.items()):
    n = len(dd.get('GRBM_GUI_ACTIVE', []))
    if not n: continue
    m = sum(dd['SQ_VALU_MFMA_BUSY_CYCLES']) / n; g = sum(dd['GRBM_GUI_ACTIVE']) / n
    busy = 100.0 * m / (g / 8 * 1024)
    valu = sum(dd['SQ_INSTS_VALU']) / n
    lds = 100.0 * (sum(dd['SQ_WAIT_INST_LDS']) / n) / max(sum(dd['SQ_WAVE_CYCLES']) / n, 1)
    print("| `%s` | %d | %.1f | %.2f | %.1f |" % (k, n, busy, 16.0 * valu / max(m, 1), lds))
PY
cat gpurun_out/r04/r04_train_sq_counters.md
