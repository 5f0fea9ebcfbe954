#!/bin/bash
# Official measurement of a round on the GPU box: bench line, rocprofv3 kernel stats, and the two PMC passes.
# usage: bash scripts/measure_round.sh TAG      (outputs under gpurun_out/, to be copied into profiles/)
set -e
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/${TAG}_bench_cfg3_1gpu.json 2> gpurun_out/${TAG}_bench.err
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/prof_stats.err
python scripts/summarize_rocprof.py gpurun_out/prof_stats > gpurun_out/${TAG}_rocprof_stats_cfg3.txt
rm -rf gpurun_out/prof_stats
echo "stats done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > /dev/null 2> gpurun_out/prof_fetch.err
echo "fetch done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > /dev/null 2> gpurun_out/prof_write.err
python scripts/summarize_rocprof.py gpurun_out/prof_fetch gpurun_out/prof_write > gpurun_out/${TAG}_rocprof_pmc_cfg3.txt
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write
echo "pmc done"
