#!/bin/bash
# GPU box: SQ counters of the 512^3 noisy colour sweep (tools/kbench.py), production build and any A/B build under build/ab_*;
# separate rocprofv3 passes per counter group.
root=$GRAFT_REPO_ROOT
for lib in "" $(ls -d $root/build/ab_*/libparmgmc_hip.so 2>/dev/null); do
  echo "=== ${lib:-production}"
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE"; do
    PMG_LIBRARY=$lib $root/tools/pmc_one.sh head "$grp" "grid_color_sweep_kernel<true, true, false, false, false>" tools/kbench.py --n 512 --omega 1.0 --reps 20 --mode noisy --no-copy 2>&1 | grep sweep_kernel
  done
done
