#!/bin/bash
# usage: bash scripts/build_one_variant.sh NAME SRC(.hip, no suffix) "-D..." ...  -> fpc_diffrend_amd/libfpcdr_NAME.so with that ONE source
# rebuilt under the switches (the other objects come from csrc/_build: run make first)
set -e
NAME=$1; SRC=$2; shift; shift
cd "$(dirname "$0")/../fpc_diffrend_amd/csrc"
mkdir -p _b_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed "$@" -c $SRC.hip -o _b_$NAME/$SRC.o
objs=$(ls _build/*.o | grep -v "_tc.o" | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfpcdr_$NAME.so $objs _b_$NAME/$SRC.o
rm -rf _b_$NAME
echo built libfpcdr_$NAME.so
