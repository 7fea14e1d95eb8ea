set -x
run() { PMG_BENCH_SHARE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 --steps $2 --warmup 5 --grid-n $3 --no-mgmc > gpurun_out/reh_$1.json 2> gpurun_out/reh_$1.err; echo "rc=$?"; grep -o '"value": [0-9.]*' gpurun_out/reh_$1.json | head -1; grep "pmg error" gpurun_out/reh_$1.err | head -2; }
run 29541 50 128
run 29542 50 512
run 29543 5 512
