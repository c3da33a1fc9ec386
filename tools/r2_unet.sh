cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_ldm.py -q -x -m gpu -k "golden or unet or conv or ldm" 2>&1 | tail -2
rm -rf /tmp/ul; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/ul -o u -- python $GRAFT_REPO_ROOT/tools/unet_layers_run.py 4 > /dev/null 2>&1)
python tools/unet_layers.py $(find /tmp/ul -name "*.db" | head -1) 4 > gpurun_out/unet_layers.md; cut -c1-150 gpurun_out/unet_layers.md
