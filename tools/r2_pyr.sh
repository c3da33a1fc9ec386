cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/pp; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pp -o b -- python $GRAFT_REPO_ROOT/tools/time_pyr.py > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pp -name "*.db" | head -1) | grep -i "sample_pyr\|sort\|hist\|scan\|scatter\|rank\|fill\|copy\|total" | cut -c1-200
