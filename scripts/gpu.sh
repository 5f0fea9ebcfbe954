#!/bin/bash
# usage: bash scripts/gpu.sh TIMEOUT 'command'  -- gpurun, retried only while the pod has no free GPU slot (exit code 3: nothing ran, nothing charged)
T=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
