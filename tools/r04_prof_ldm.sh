# kernel trace of the batch-1 LDM denoise step alone (tools/time_ldm.py: 13 eager steps + 13 graph steps)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r04
rm -rf /tmp/pl; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pl -o l -- python $GRAFT_REPO_ROOT/tools/time_ldm.py 1 64 > /tmp/pl.log 2>&1)
python tools/rocpd_summary.py $(find /tmp/pl -name "*.db" | head -1) > gpurun_out/r04/ldm_b1_stats.md
cat /tmp/pl.log | tail -4 >> gpurun_out/r04/ldm_b1_stats.md
python - <<'PY' >> gpurun_out/r04/ldm_b1_stats.md
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/pl/**/*.db', recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# the last graph replay = the last N dispatches after the final big gap; print gaps statistics of the last 400 dispatches
last = rows[-450:]
gaps = [last[i+1][1] - last[i][2] for i in range(len(last)-1)]
busy = sum(e - s for _, s, e in last)
print("\nlast 450 dispatches: span %.3f ms, kernel-busy %.3f ms, mean gap %.2f us, gaps > 3us: %d" % ((last[-1][2]-last[0][1])/1e6, busy/1e6, sum(gaps)/len(gaps)/1e3, sum(g > 3000 for g in gaps)))
PY
head -30 gpurun_out/r04/ldm_b1_stats.md | cut -c1-150; tail -8 gpurun_out/r04/ldm_b1_stats.md
