# round-6 final artefacts: full GPU suite, default bench line, kernel traces (bench inference loop, train step, LDM batch 1),
# HBM counters and SQ counters of the inference kernels, marker trace.  Outputs -> gpurun_out/r06f/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
if [ -z "$S3D_PROFILES_ONLY" ]; then   # S3D_PROFILES_ONLY=1: only the traces and counters
python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; tail -22 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; wc -c $O/bench.json; tail -c 2200 $O/bench.json
fi
BENCH="$GRAFT_REPO_ROOT/bench.py --infer-only"
(cd /tmp && rm -rf /tmp/p1 && rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python $BENCH --steps 10 --warmup 2 > /tmp/p1.json 2>/dev/null)
python tools/rocpd_summary.py $(find /tmp/p1 -name "*.db" | head -1) > $O/r06_bench_f16x3_kernel_stats.md
echo >> $O/r06_bench_f16x3_kernel_stats.md; echo "bench line of the traced run:" >> $O/r06_bench_f16x3_kernel_stats.md; tail -c 2500 /tmp/p1.json >> $O/r06_bench_f16x3_kernel_stats.md
(cd /tmp && rm -rf /tmp/p2 && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p2 -o f -- python $BENCH --steps 3 --warmup 1 > /dev/null 2>&1)
(cd /tmp && rm -rf /tmp/p3 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p3 -o w -- python $BENCH --steps 3 --warmup 1 > /dev/null 2>&1)
cp $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $O/pmc_fetch.csv
cp $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $O/pmc_write.csv
python tools/pmc_summary.py $O/pmc_fetch.csv $O/pmc_write.csv f16x3 $O/r06_bench_f16x3_pmc_hbm.md --json $O/pmc_traffic.json
rm -rf /tmp/psq; (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/psq -o s -- python $BENCH --steps 3 --warmup 1 > /dev/null 2>&1)
python - > $O/r06_bench_f16x3_sq_counters.md <<'PY'
import csv, glob, collections
print("# SQ counters of the decoder kernels, round 6 (rocprofv3 --pmc, `bench.py --steps 3 --warmup 1`, inference legs only)\n")
print("Per-launch averages.  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).\n")
fs = glob.glob('/tmp/psq/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])) if fs else []:
    k = r['Kernel_Name']
    if 'ffn_layer' in k or 'attn_layer' in k or 'attn_last' in k or 'sample_tokens' in k or 'conv3x3_lds' in k:
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
PY
rm -rf /tmp/pt; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1)
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > $O/r06_train_f16x3_kernel_stats.md; grep "wall time" /tmp/pt.log >> $O/r06_train_f16x3_kernel_stats.md
(cd /tmp && rm -rf /tmp/pm && rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/pm -o m -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pm.log 2>&1)
python tools/marker_summary.py /tmp/pm > $O/r06_train_marker_trace.md 2>&1
rm -rf /tmp/pl; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pl -o l -- python $GRAFT_REPO_ROOT/tools/time_ldm.py 1 64 > /tmp/pl.log 2>&1)
python tools/rocpd_summary.py $(find /tmp/pl -name "*.db" | head -1) > $O/r06_ldm_b1_kernel_stats.md; grep "LDM denoise" /tmp/pl.log >> $O/r06_ldm_b1_kernel_stats.md
bash tools/clock_probe.sh > $O/r06_clock_power_raw.txt 2>&1
head -12 $O/r06_bench_f16x3_kernel_stats.md | cut -c1-150; head -14 $O/r06_train_f16x3_kernel_stats.md | cut -c1-150; tail -2 $O/r06_train_f16x3_kernel_stats.md
