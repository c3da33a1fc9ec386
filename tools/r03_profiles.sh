# Round-3 profile artefacts (copied into profiles/ by hand afterwards): kernel trace, HBM PMC passes, SQ counters of the
# dominant kernels, clock/power under load.  One rocprofv3 --pmc set per run (no trace domains mixed in).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r03
BENCH="$GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --train-steps 0 --gt-train-steps 0 --ldm-steps 0 --c4-steps 0 --f16-steps 0 --mesh-steps 0 --pmc 0"
(cd /tmp && rm -rf /tmp/p1 && rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python $BENCH --steps 10 --warmup 2 > /tmp/p1.json 2>/dev/null)
python tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) > gpurun_out/r03/r03_bench_f16x3_kernel_stats.md
echo >> gpurun_out/r03/r03_bench_f16x3_kernel_stats.md; echo "bench line of the traced run:" >> gpurun_out/r03/r03_bench_f16x3_kernel_stats.md; tail -c 2500 /tmp/p1.json >> gpurun_out/r03/r03_bench_f16x3_kernel_stats.md
(cd /tmp && rm -rf /tmp/p2 && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o f -- python $BENCH --steps 3 --warmup 1 > /dev/null 2>&1)
(cd /tmp && rm -rf /tmp/p3 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o w -- python $BENCH --steps 3 --warmup 1 > /dev/null 2>&1)
cp $(find /tmp/p2 -name "*counter_collection.csv" | head -1) gpurun_out/r03/pmc_fetch.csv
cp $(find /tmp/p3 -name "*counter_collection.csv" | head -1) gpurun_out/r03/pmc_write.csv
python tools/pmc_summary.py gpurun_out/r03/pmc_fetch.csv gpurun_out/r03/pmc_write.csv f16x3 gpurun_out/r03/r03_bench_f16x3_pmc_hbm.md --json gpurun_out/r03/pmc_traffic.json
rm -rf /tmp/psq; (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/psq -o s -- python $BENCH --steps 3 --warmup 1 > /dev/null 2>&1)
rm -rf /tmp/psq2; (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --output-format csv -d /tmp/psq2 -o s -- python $BENCH --steps 3 --warmup 1 > /dev/null 2>&1)
python - > gpurun_out/r03/r03_bench_f16x3_sq_counters.md <<'PY'
import csv, glob, collections
print("# SQ counters of the decoder kernels (rocprofv3 --pmc, two passes of 8 counters, `bench.py --steps 3 --warmup 1`, inference legs only)\n")
print("Per-launch averages.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_VALU_MFMA_BUSY_CYCLES and")
print("SQ_BUSY_CYCLES count cycles (MI355X_MICROARCH.md).  MFMA pipe busy below = SQ_VALU_MFMA_BUSY_CYCLES (16 cycles per four-pass MFMA,")
print("summed over the SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): GRBM_GUI_ACTIVE comes back summed over the 8 XCDs.\n")
for d in ('/tmp/psq', '/tmp/psq2'):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print('no csv in', d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if 'ffn_layer' in k or 'attn_layer' in k or 'sample_tokens' in k or 'conv3x3_lds' in k:
            acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, dd in sorted(acc.items()):
        print("\n`%s`\n" % k)
        print("| counter | avg per launch | launches |\n|---|---|---|")
        for c, v in dd.items():
            print('| %s | %.5g | %d |' % (c, sum(v) / len(v), len(v)))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in dd and 'GRBM_GUI_ACTIVE' in dd:
            m = sum(dd['SQ_VALU_MFMA_BUSY_CYCLES']) / len(dd['SQ_VALU_MFMA_BUSY_CYCLES'])
            g = sum(dd['GRBM_GUI_ACTIVE']) / len(dd['GRBM_GUI_ACTIVE'])
            print("\nMFMA pipe busy = %.1f %% of the launch's SIMD cycles" % (100.0 * m / (g / 8 * 1024)))
        if 'SQ_INSTS_VALU' in dd and 'SQ_INSTS_MFMA' in dd:
            print("\nVALU (non-MFMA) instructions per MFMA = %.2f" % ((sum(dd['SQ_INSTS_VALU']) / len(dd['SQ_INSTS_VALU'])) / (sum(dd['SQ_INSTS_MFMA']) / len(dd['SQ_INSTS_MFMA'])) - 1.0))
PY
bash tools/clock_probe.sh > gpurun_out/r03/r03_clock_power_raw.txt 2>&1
head -12 gpurun_out/r03/r03_bench_f16x3_pmc_hbm.md | cut -c1-160; head -60 gpurun_out/r03/r03_bench_f16x3_sq_counters.md; cat gpurun_out/r03/r03_clock_power_raw.txt
# ---- training step: kernel trace + SQ counters of its dominant kernels ----
rm -rf /tmp/pt; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/r03/r03_train_f16x3_kernel_stats.md
rm -rf /tmp/psq; (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/psq -o s -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /dev/null 2>&1)
rm -rf /tmp/psq2; (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/psq2 -o s -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /dev/null 2>&1)
python - > gpurun_out/r03/r03_train_sq_counters.md <<'PY'
import csv, glob, collections
print("# SQ counters of the training step's dominant kernels (rocprofv3 --pmc, two passes, `tools/prof_train.py`: 3 steps, 4 objects x 100 k queries)\n")
print("Per-launch averages (each FFN kernel: two 5.2 M-row layers + the 0.4 M-row last layer per step).  MFMA pipe busy =")
print("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).\n")
for d in ('/tmp/psq', '/tmp/psq2'):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print('no csv in', d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if any(s in k for s in ('ffn_wgrad_rec_kernel', 'ffn_layer_f16x3', 'ffn_rec_images', 'wgrad_lin', 'attn_core', 'sample_bwd', 'ln_bwd')):
            acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, dd in sorted(acc.items()):
        print("\n`%s`\n" % k)
        print("| counter | avg per launch | launches |\n|---|---|---|")
        for c, v in dd.items():
            print('| %s | %.5g | %d |' % (c, sum(v) / len(v), len(v)))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in dd and 'GRBM_GUI_ACTIVE' in dd:
            m = sum(dd['SQ_VALU_MFMA_BUSY_CYCLES']) / len(dd['SQ_VALU_MFMA_BUSY_CYCLES'])
            g = sum(dd['GRBM_GUI_ACTIVE']) / len(dd['GRBM_GUI_ACTIVE'])
            if m > 0:
                print("\nMFMA pipe busy = %.1f %% of the launch's SIMD cycles" % (100.0 * m / (g / 8 * 1024)))
        if 'SQ_INSTS_VALU' in dd and 'SQ_INSTS_MFMA' in dd and sum(dd['SQ_INSTS_MFMA']) > 0:
            print("\nVALU (non-MFMA) instructions per MFMA = %.2f" % ((sum(dd['SQ_INSTS_VALU']) / len(dd['SQ_INSTS_VALU'])) / (sum(dd['SQ_INSTS_MFMA']) / len(dd['SQ_INSTS_MFMA'])) - 1.0))
PY
grep -E "MFMA pipe busy|^\`" gpurun_out/r03/r03_train_sq_counters.md | head -40
