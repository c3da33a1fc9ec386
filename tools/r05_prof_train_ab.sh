cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05
for tag in product noregen; do
  lib=""; [ $tag = noregen ] && lib=$PWD/build/abl/lib_noregen.so
  rm -rf /tmp/pt; (cd /tmp && S3D_HIP_LIB=$lib rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1)
  python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/r05/train_stats_$tag.md
  tail -1 /tmp/pt.log >> gpurun_out/r05/train_stats_$tag.md
  echo "== $tag"; head -16 gpurun_out/r05/train_stats_$tag.md | cut -c1-140; tail -2 gpurun_out/r05/train_stats_$tag.md
done
