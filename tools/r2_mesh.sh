cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_mesh.py tests/test_gpu_parity.py -q -x -m gpu -k "mesh or mise or marching or generator or slab" 2>&1 | tail -15
python tools/time_mesh_device.py 2>&1 | tail -6
