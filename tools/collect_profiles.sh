#!/bin/bash
# here (after tools/final_profiles.sh ran on the GPU box): copy what is judged from gpurun_out/ into profiles/
tag=${1:-r04}
set -e
cp "$(ls -t gpurun_out/$tag/trace/*/*_kernel_stats.csv | head -1)" profiles/${tag}_rocprofv3_kernel_stats.csv   # newest: gpurun merges, it does not delete
cp "$(ls -t gpurun_out/$tag/trace_full/*/*_kernel_stats.csv | head -1)" profiles/${tag}_rocprofv3_kernel_stats_full_bench.csv
cp gpurun_out/${tag}_summary.csv gpurun_out/${tag}_summary.json gpurun_out/${tag}_bench_under_rocprof.json gpurun_out/${tag}_cycles_summary.json gpurun_out/${tag}_valubench.txt profiles/
tail -1 gpurun_out/${tag}_bench.json > profiles/${tag}_bench.json
ls -la profiles/${tag}_*
