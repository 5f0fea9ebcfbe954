#!/bin/bash
# one GPU call: targeted tests, stream-overlap variants, step timeline, bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1
timeout -k 10 700 python -m pytest tests/test_gpu_objective.py tests/test_gpu_fit.py tests/test_gpu_gaps.py tests/test_gpu_large.py tests/test_gpu_dist.py -x -q > gpurun_out/t_$T.log 2>&1 || { tail -30 gpurun_out/t_$T.log; exit 1; }
tail -3 gpurun_out/t_$T.log
timeout -k 10 400 python scripts/dbg/step_variants.py 2>&1 | tail -6
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$T -- python3 scripts/prof_objective.py --ops 0 > gpurun_out/tl_$T.log 2>&1 && python scripts/step_timeline.py gpurun_out/tl_$T --all > gpurun_out/r4_timeline_$T.txt
timeout -k 10 400 python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_$T.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
