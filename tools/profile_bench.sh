#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the SAME bench command,
# condensed into gpurun_out/<tag>_summary.{json,csv}.  Usage: tools/profile_bench.sh <tag> [bench args]
set -e
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/$tag
rm -rf $out
mkdir -p $out
steps=${STEPS:-20}     # the driver runs bench.py --steps 20 --warmup 5: profile THAT command (STEPS / WARMUP override)
warmup=${WARMUP:-5}
args="--steps $steps --warmup $warmup --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $args --no-mgmc > $out/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_full -- python3 bench.py $args > $out/bench_trace_full.log 2>&1  # incl. the V-cycle lines
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $args --no-mgmc > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $args --no-mgmc > $out/bench_write.log 2>&1
python3 tools/summarize_profile.py $out/trace $out/pmc_fetch $out/pmc_write gpurun_out/${tag}_summary 512 $((2 * steps))
grep -h '"metric"' $out/bench_trace.log > gpurun_out/${tag}_bench_under_rocprof.json || true
