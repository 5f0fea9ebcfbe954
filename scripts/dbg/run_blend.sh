#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_fit.py -x -q 2>&1 | tail -2
for v in default slab256 slab128; do
  if [ $v = default ]; then unset FPCDR_LIB_PATH; else export FPCDR_LIB_PATH=$PWD/fpc_diffrend_amd/libfpcdr_$v.so; fi
  echo "== $v"; timeout -k 10 120 python scripts/time_blend.py 2>&1 | tail -2
done
unset FPCDR_LIB_PATH
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_b -- python3 scripts/prof_objective.py --ops 0 > gpurun_out/tl_b.log 2>&1 && python scripts/step_timeline.py gpurun_out/tl_b --all > gpurun_out/r4_timeline_b.txt
grep -n "k_init_objective\|k_blend\|step:" gpurun_out/r4_timeline_b.txt
