#!/bin/bash
# GPU box: kernel trace of the low-rank V-cycle (257^3, k = 3), chain of small kernels and PMG_LRC_FUSED=1: per-kernel averages
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in fused unfused; do
  out=$root/gpurun_out/lrc_trace_$mode; rm -rf $out
  if [ $mode = fused ]; then export PMG_LRC_FUSED=1; else unset PMG_LRC_FUSED; fi
  rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/tools/cyclebench.py mgmc_lowrank_257_5_k3 10 > $out.log 2>&1
  echo "== $mode"; python3 $root/tools/kstat.py $out lrc fill_normal | head -20
  find $out -name '*.csv' -size +2M -delete
done
