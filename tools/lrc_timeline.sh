#!/bin/bash
# GPU box: launch-by-launch timeline of one low-rank V-cycle sample (257^3, k = 3) under rocprofv3 --kernel-trace
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/lrc_timeline; rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/tools/cyclebench.py ${1:-mgmc_lowrank_257_5_k3} 10 > $out.log 2>&1
python3 $root/tools/ktimeline.py $out 10 > $out.txt
tail -3 $out.txt
find $out -name '*.csv' -size +2M -delete
