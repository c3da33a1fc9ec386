# SQ counter pass over the LDM step's kernels (tools/time_attn48.py + one eager step): per-kernel averages of the new round-6 kernels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06final; mkdir -p $O
rm -rf /tmp/psq; (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/psq -o s -- python $GRAFT_REPO_ROOT/tools/time_attn48.py > /dev/null 2>&1)
rm -rf /tmp/psq3; (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/psq3 -o s -- python $GRAFT_REPO_ROOT/tools/ldm_out.py /tmp/y.pt 1 > /dev/null 2>&1)
python - > $O/r06_ldm_sq_counters.md <<'PY'
import csv, glob, collections
print("# SQ counters of the round-6 LDM kernels (`tools/r06_ldm_sq.sh`; per-launch averages; matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) as in the other sq_counters files)\n")
for d, what, keys in (('/tmp/psq', 'tools/time_attn48.py (1 x 1 024 and 4 x 1 024 and 1 x 4 096 tokens, 8 heads of 48; 1 x 4 096 of 24)', ('la_attention', 'la_pack', 'la_merge', 'qkv_attention')),
                      ('/tmp/psq3', 'one eager denoise step, batch 1 (warm-up included: tools/ldm_out.py)', ('conv3x3_small', 'conv3x3_lds', 'small_linear', 'conv_splitk'))):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print('no csv in', d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if any(x in k for x in keys):
            acc[k[:75]][r['Counter_Name']].append(float(r['Counter_Value']))
    print("## %s\n" % what)
    print("| kernel | launches | wave cycles | waiting (any) | waiting for issue | LDS wait | matrix pipe busy |")
    print("|---|---|---|---|---|---|---|")
    for k, dd in sorted(acc.items()):
        m = lambda c: sum(dd[c]) / len(dd[c]) if dd.get(c) else float('nan')
        wc = m('SQ_WAVE_CYCLES')
        print("| `%s` | %d | %.3g | %.0f %% | %.0f %% | %.0f %% | %.1f %% |" % (k, len(dd['SQ_WAVE_CYCLES']), wc, 100 * m('SQ_WAIT_ANY') / wc, 100 * m('SQ_WAIT_INST_ANY') / wc,
              100 * m('SQ_WAIT_INST_LDS') / wc, 100 * m('SQ_VALU_MFMA_BUSY_CYCLES') / (m('GRBM_GUI_ACTIVE') / 8 * 1024)))
    print()
PY
cat $O/r06_ldm_sq_counters.md | cut -c1-200
