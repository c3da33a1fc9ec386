# kernel trace of the inference bench; prints conv kernels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/pi; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pi -o b -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --train-steps 0 --gt-train-steps 0 > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pi -name "*.db" | head -1) > gpurun_out/infer_kernels.md
grep -i "conv\|pool" gpurun_out/infer_kernels.md | awk -F'|' '{printf "%-60s %5s calls %8s ms avg %8s us grid %9s\n", substr($2,1,60), $3, $4, $5, $9}'
