#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --workload cfg2 --no-cpu-baseline > gpurun_out/r04_bench_cfg2.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload cfg5 --no-cpu-baseline > gpurun_out/r04_bench_cfg5.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --mip --no-cpu-baseline --no-reference-shaped-step > gpurun_out/r04_bench_cfg3_mip.json 2>/dev/null || exit 1
for w in cfg2 cfg5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-reference-shaped-step > /dev/null 2>&1 || exit 1
  python scripts/summarize_rocprof.py gpurun_out/ps_$w > gpurun_out/r04_rocprof_stats_$w.txt && rm -rf gpurun_out/ps_$w
done
python - <<'PY'
import json
for w in ("cfg2", "cfg5", "cfg3_mip"):
    d = json.loads(open(f"gpurun_out/r04_bench_{w}.json").read().strip().splitlines()[-1]); print(w, d["ms_per_step"], d["value"])
PY
