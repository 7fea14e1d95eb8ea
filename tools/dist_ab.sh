# A/B of the z-slab V-cycle kernels on ONE GPU shared by two ranks (PMG_ST27_PAIR_SLAB = 1 | 0): bash tools/dist_ab.sh
export PMG_BENCH_SHARE_DEVICE=1
for v in 1 0; do PMG_ST27_PAIR_SLAB=$v timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 5 --no-cpu-baseline 2> gpurun_out/ab_$v.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pair_slab $v', 'n_gpus', d['n_gpus'], 'ms_per_step', round(d['ms_per_step'],3), {k: (v.get('ms_per_sample') if isinstance(v, dict) else v) for k, v in d.items() if k.startswith('secondary')})
" || exit 1; done
