#!/bin/bash
# usage: bash scripts/ab2.sh "NAME[:ENV=VAL,...]"...  -- scripts/prof_objective.py timings (HIP events of the two objective calls) per variant
mkdir -p gpurun_out
for spec in "$@"; do
  n=${spec%%:*}; envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  if [ "$n" = default ]; then lib=""; else lib="FPCDR_LIB_PATH=$PWD/fpc_diffrend_amd/libfpcdr_$n.so"; fi
  out=$(env $lib $envs timeout -k 10 200 python scripts/prof_objective.py --ops 0 2>/dev/null | tail -1) || { echo "FAILED $spec"; exit 1; }
  echo "$spec $out"
done
