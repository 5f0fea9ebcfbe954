#!/bin/bash
# usage: bash scripts/build_obj_variant.sh NAME "-D..."  -> fpc_diffrend_amd/libfpcdr_NAME.so with objective.hip rebuilt under the switches
# (the other objects come from csrc/_build: run make first)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../fpc_diffrend_amd/csrc"
mkdir -p _b_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed "$@" -c objective.hip -o _b_$NAME/objective.o
objs=$(ls _build/*.o | grep -v "_tc.o" | grep -v objective.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfpcdr_$NAME.so $objs _b_$NAME/objective.o
rm -rf _b_$NAME
echo built libfpcdr_$NAME.so
