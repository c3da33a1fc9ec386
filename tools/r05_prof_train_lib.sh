# kernel trace of the train step with a variant library:  tools/r05_prof_train_lib.sh <tag> <lib or ""> [kernel filter]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05
tag=$1; lib=$2; flt=${3:-sample_bwd}
rm -rf /tmp/pt; (cd /tmp && S3D_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} rocprofv3 --kernel-trace -d /tmp/pt -o tr -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /tmp/pt.log 2>&1)
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/r05/train_stats_$tag.md
tail -1 /tmp/pt.log >> gpurun_out/r05/train_stats_$tag.md
echo "== $tag"; grep -E "$flt|total kernel" gpurun_out/r05/train_stats_$tag.md | cut -c1-160; tail -1 gpurun_out/r05/train_stats_$tag.md
