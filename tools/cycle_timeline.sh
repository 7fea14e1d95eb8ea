#!/bin/bash
# GPU box: launch-by-launch timelines (tools/ktimeline.py) of one sample of each V-cycle workload under rocprofv3 --kernel-trace
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for key in ${@:-mgmc_257_5 mgmc_513_6 mgmc_lowrank_257_5_k3 mgmc_aij_377089}; do
  out=$root/gpurun_out/timeline_$key; rm -rf $out
  rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/tools/cyclebench.py $key 10 > $out.log 2>&1
  python3 $root/tools/ktimeline.py $out 10 > $out.txt
  echo "== $key"; tail -1 $out.txt
  rm -rf $out
done
