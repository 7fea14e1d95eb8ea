#!/bin/bash
# GPU box: SQ counters of the class-stencil phase kernel at the 257^3 level (tools/st27bench.py), production build and any
# A/B build under build/ab_*; separate rocprofv3 passes per counter group (a pass takes at most a few counters).
root=$GRAFT_REPO_ROOT
for lib in "" $(ls -d $root/build/ab_*/libparmgmc_hip.so 2>/dev/null); do
  echo "=== ${lib:-production}"
  for grp in "SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
    PMG_LIBRARY=$lib $root/tools/pmc_one.sh st27 "$grp" "st27_pair_phase_kernel<true, false, false, false, (false|true)>" tools/st27bench.py 2>&1 | grep phase_kernel
  done
done
