#!/bin/bash
# Official measurement of a round on the GPU box: bench line, rocprofv3 kernel stats, and the PMC passes (each counter set in a
# pass of its own, with --kernel-trace only), FIRST, so that the bench line of the same call can pair them.   usage: bash scripts/measure_round.sh TAG [quick|stats]
# Outputs under gpurun_out/ (to be copied into profiles/): TAG_bench_cfg3_1gpu.json, TAG_rocprof_stats_cfg3.txt,
# TAG_rocprof_pmc_cfg3.txt, TAG_counters.json
set -e
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
if [ "$2" != stats ]; then
# PMC passes over scripts/prof_objective.py (seven fit steps at cfg3, nothing else): one counter set per pass
pass() {   # name, counters...   (a failed pass ends the script: no partial counters file for bench.py to trust)
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -- python3 scripts/prof_objective.py > /dev/null 2> gpurun_out/pmc_$name.err || { echo "pass $name FAILED"; exit 1; }
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass sq3 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE
python scripts/summarize_rocprof.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3 > gpurun_out/${TAG}_rocprof_pmc_cfg3.txt
python scripts/make_counters_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3 > gpurun_out/${TAG}_counters.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3
# the bench line pairs its own timings with these counters (same kernel sources, durations within 10 %): it reads the file from profiles/
cp gpurun_out/${TAG}_counters.json profiles/${TAG}_counters_cfg3.json
echo "pmc done"
fi

if [ "$2" != quick ]; then
  if [ "$2" != stats ]; then
  timeout -k 10 500 python bench.py > gpurun_out/${TAG}_bench_cfg3_1gpu.json 2> gpurun_out/${TAG}_bench.err
  echo "bench done"
  fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-shaped-step > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/prof_stats.err
  python scripts/summarize_rocprof.py gpurun_out/prof_stats > gpurun_out/${TAG}_rocprof_stats_cfg3.txt
  rm -rf gpurun_out/prof_stats
  echo "stats done"
  # (the profiled run leaves out the one-image reference-shaped steps: their launches of the same kernels would dilute the averages)
fi
