#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in default mipw4 mipw5; do
  if [ $v = default ]; then unset FPCDR_LIB_PATH; else export FPCDR_LIB_PATH=$PWD/fpc_diffrend_amd/libfpcdr_$v.so; fi
  echo "== $v $(timeout -k 10 200 python scripts/dbg/mip_step.py 2>&1 | tail -1)"
done
