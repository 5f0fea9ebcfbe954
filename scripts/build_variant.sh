#!/bin/bash
# usage: bash scripts/build_variant.sh NAME "-DFPCDR_...=..."   -> fpc_diffrend_amd/libfpcdr_NAME.so (A/B builds for scripts/ab.sh)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../fpc_diffrend_amd/csrc"
B=_build_$NAME
mkdir -p $B
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function -Wno-pass-failed $*"
pids=()
for s in abi rasterize interpolate texture antialias blend loss objective clip adam; do
  /opt/rocm/bin/hipcc $FLAGS -c $s.hip -o $B/$s.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfpcdr_$NAME.so $B/*.o
rm -rf $B
echo built libfpcdr_$NAME.so
