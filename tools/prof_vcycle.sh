#!/bin/bash
# usage (on the GPU box, through gpurun): tools/prof_vcycle.sh <tag>   -> gpurun_out/vc_<tag>.csv (per-kernel summary)
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_vc_$tag
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_vc_$tag -o vc -- python3 $GRAFT_REPO_ROOT/tools/vcyclebench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_vc_$tag.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_kernels.py $GRAFT_REPO_ROOT/gpurun_out/prof_vc_$tag/vc_results.db $GRAFT_REPO_ROOT/gpurun_out/vc_$tag.csv > /dev/null
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_vc_$tag
grep "V-cycle" $GRAFT_REPO_ROOT/gpurun_out/prof_vc_$tag.log
grep -E "st27|q1_|grid_residual|tri_gemv" $GRAFT_REPO_ROOT/gpurun_out/vc_$tag.csv | head -40
