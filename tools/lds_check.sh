cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_gt_train.py tests/test_gpu_train.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
(cd /tmp && rm -rf /tmp/pg && S3D_GT_STEPS=3 rocprofv3 --kernel-trace -d /tmp/pg -o g -- python $GRAFT_REPO_ROOT/tools/time_gt_train.py > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pg -name "*.db" | head -1) | grep -E "wgrad_conv3|wgrad_lin" | cut -c1-120
(cd /tmp && rm -rf /tmp/pt && rocprofv3 --kernel-trace -d /tmp/pt -o g -- python $GRAFT_REPO_ROOT/tools/prof_train.py > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) | grep -E "wgrad_conv3|wgrad_lin|ffn_wgrad_rec_kernel" | cut -c1-120
(cd /tmp && rm -rf /tmp/pb && rocprofv3 --kernel-trace -d /tmp/pb -o g -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --train-steps 0 --gt-train-steps 0 --ldm-steps 0 > /tmp/b.json 2>/dev/null)
python tools/rocpd_summary.py $(find /tmp/pb -name "*.db" | head -1) | grep -E "attn_layer" | cut -c1-120
python -c "
import json; r=json.load(open('/tmp/b.json')); print(r['value'], r['stage_ms_per_step'])"
