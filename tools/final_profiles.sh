#!/bin/bash
# GPU box: everything profiles/rNN_* is made of, on the build that is in the tree (usage: tools/final_profiles.sh r04)
tag=${1:-r04}
root=$GRAFT_REPO_ROOT
cd $root
tools/profile_bench.sh $tag > gpurun_out/${tag}_profile_bench.log 2>&1; echo "[final] profile_bench rc=$?"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "[final] bench rc=$?"
tools/profile_cycles.sh $tag > gpurun_out/${tag}_profile_cycles.log 2>&1; echo "[final] profile_cycles rc=$?"
tools/pmc_st27.sh > gpurun_out/${tag}_st27_phase_counters_raw.txt 2>&1; echo "[final] pmc_st27 rc=$?"
tools/pmc_headline.sh > gpurun_out/${tag}_headline_counters_raw.txt 2>&1; echo "[final] pmc_headline rc=$?"
tools/valubench 200 8 > gpurun_out/${tag}_valubench.txt 2>&1; echo "[final] valubench rc=$?"
