#!/bin/bash
# usage: bash scripts/build_rast_variant.sh NAME "-D..."  -> fpc_diffrend_amd/libfpcdr_NAME.so with rasterize.hip rebuilt under the switches
set -e
NAME=$1; shift
cd "$(dirname "$0")/../fpc_diffrend_amd/csrc"
mkdir -p _b_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed "$@" -c rasterize.hip -o _b_$NAME/rasterize.o
objs=$(ls _build/*.o | grep -v rasterize.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfpcdr_$NAME.so $objs _b_$NAME/rasterize.o
rm -rf _b_$NAME
echo built libfpcdr_$NAME.so
