# bench.py with N ranks sharing ONE GPU (rehearsal of the multi-GPU path): NG=4 bash tools/bench_share2.sh
export PMG_BENCH_SHARE_DEVICE=1
timeout -k 10 500 python bench.py --gpus ${NG:-2} --steps 10 --warmup 5 --no-cpu-baseline 2> gpurun_out/n2.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n_gpus', d['n_gpus'], 'ms_per_step', round(d['ms_per_step'],3), d.get('halo_check'))
for k, v in d.items():
    if k.startswith('secondary'): print(k, v)
"; tail -3 gpurun_out/n2.err
