#!/bin/bash
# GPU box: run a tool with every A/B library build (build/ab_*) and the production one.  usage: tools/ab_run.sh <python tool> [args]
for lib in "" $(ls -d build/ab_*/libparmgmc_hip.so 2>/dev/null); do
  echo "=== ${lib:-production}"
  PMG_LIBRARY=$lib python "$@" 2>&1 | grep -v amdgpu.ids
done
