cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
S3D_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --batch 1 --n-qry 20000 --img-size 128 --train-steps 2 --c4-res 64 --ldm-steps 1 --gt-train-steps 1 --f16-steps 1 2>gpurun_out/bench2.err | tail -c 1800
echo; tail -5 gpurun_out/bench2.err
