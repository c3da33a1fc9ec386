# SQ / TA counters of selected kernels of the training step (tools/prof_train.py) (two --pmc passes), per-launch averages
#   gpurun -- 'bash tools/r06_pmc_kernels.sh <tag> "sample_bwd_dense"'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-pmc}; PAT=${2:-"sample_bwd_dense"}
O=gpurun_out/$TAG; mkdir -p $O
BENCH="$GRAFT_REPO_ROOT/tools/prof_train.py"
rm -rf /tmp/pq1 /tmp/pq2 /tmp/pq3
(cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pq1 -o s -- python $BENCH > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/pq2 -o s -- python $BENCH > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_WAVES --output-format csv -d /tmp/pq3 -o s -- python $BENCH > /dev/null 2>&1)
python - "$PAT" > $O/sq_counters.md <<'PY'
import csv, glob, collections, re, sys
pat = re.compile(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('/tmp/pq1', '/tmp/pq2', '/tmp/pq3'):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    for r in csv.DictReader(open(fs[0])) if fs else []:
        k = r['Kernel_Name']
        if pat.search(k):
            acc[k[:80]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, dd in sorted(acc.items()):
    print("\n`%s`\n" % k)
    print("| counter | avg per launch | launches |\n|---|---|---|")
    for c, v in dd.items():
        print('| %s | %.5g | %d |' % (c, sum(v) / len(v), len(v)))
PY
cat $O/sq_counters.md
