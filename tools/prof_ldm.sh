cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/time_ldm.py 1; python tools/time_ldm.py 4
rm -rf /tmp/pl; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pl -o b -- python $GRAFT_REPO_ROOT/tools/time_ldm.py 1 > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pl -name "*.db" | head -1) > gpurun_out/ldm_kernels.md
awk -F'|' 'NR>2 {printf "%-60s %5s calls %8s ms avg %8s us\n", substr($2,1,60), $3, $4, $5}' gpurun_out/ldm_kernels.md | head -16
