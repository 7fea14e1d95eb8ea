#!/bin/bash
# Development: an A/B build of the library with extra kernel flags, beside the production build.
#   tools/ab_build.sh <tag> "<extra hipcc flags>"   ->  build/ab_<tag>/libparmgmc_hip.so   (use with PMG_LIBRARY=...)
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/ab_$tag
mkdir -p $out
cd $root/parmgmc_amd/csrc
objs=""
for f in kernels_*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-parameter $* -c $f -o $out/${f%.hip}.o
  objs="$objs $out/${f%.hip}.o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libparmgmc_hip.so $objs pmg_*.o -L/opt/rocm/lib -lamdhip64 -lm -ldl -lrt -Wl,-rpath,/opt/rocm/lib
echo $out/libparmgmc_hip.so
