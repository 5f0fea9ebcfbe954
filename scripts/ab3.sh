#!/bin/bash
# usage: bash scripts/ab3.sh NAME...  -- scripts/time_objective_parts.py (cfg3, gradients both / one / none) per library variant
for n in "$@"; do
  if [ "$n" = default ]; then unset FPCDR_LIB_PATH; else export FPCDR_LIB_PATH=$PWD/fpc_diffrend_amd/libfpcdr_$n.so; fi
  echo "$n $(timeout -k 10 200 python scripts/time_objective_parts.py 2>/dev/null | tail -1)"
done
