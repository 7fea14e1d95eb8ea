#!/bin/bash
# GPU box (via gpurun): per-sample kernel time and HBM traffic of bench.py's secondary workloads.
# usage: tools/profile_cycles.sh <tag> [key ...]   ->  gpurun_out/<tag>_cycles_summary.json (copy to profiles/)
set -e
tag=$1; shift
keys=${@:-"mgmc_257_5 mgmc_513_6 mgmc_lowrank_257_5_k3 sell_sweep_377089 sell_sweep_1505793 mgmc_aij_377089"}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/${tag}_cycles
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
S=5
for k in $keys; do
  rocprofv3 --kernel-trace --output-format csv -d $out/${k}_trace -- python3 $root/tools/cyclebench.py $k $S > $out/${k}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${k}_fetch -- python3 $root/tools/cyclebench.py $k $S > $out/${k}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${k}_write -- python3 $root/tools/cyclebench.py $k $S > $out/${k}_write.log 2>&1
  echo "[profile_cycles] $k done"
done
python3 $root/tools/summarize_cycles.py $out $root/gpurun_out/${tag}_cycles_summary.json $(for k in $keys; do echo $k:$S; done)
find $out -name '*.csv' -size +2M -delete   # keep the merge-back small
