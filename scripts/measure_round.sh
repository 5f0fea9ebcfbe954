#!/bin/bash
# Official measurement of a round on the GPU box: bench line, rocprofv3 kernel stats, and the PMC passes (each counter set in a
# pass of its own, with --kernel-trace only), FIRST, so that the bench line of the same call can pair them.
#   usage: bash scripts/measure_round.sh TAG [quick|stats|all] [WORKLOAD [bench / prof_objective flags, e.g. --mip]]
# Outputs under gpurun_out/ (to be copied into profiles/): TAG_bench_<W>.json, TAG_rocprof_stats_<W>.txt, TAG_rocprof_pmc_<W>.txt, and
# profiles/TAG_counters_<W>.json (W = cfg3, cfg2, cfg5, cfg3_mip, cfg3_c3: the file name bench.counters_file expects)
set -e
TAG=${1:-rXX}
MODE=${2:-all}
WL=${3:-cfg3}
for _ in 1 2 3; do [ $# -gt 0 ] && shift; done
EXTRA="$*"
W=$WL
MIP=0; CH=1
case " $EXTRA " in *" --mip "*) W=${W}_mip; MIP=1;; esac
case " $EXTRA " in *" --channels 3 "*) W=${W}_c3; CH=3;; esac
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
if [ "$MODE" != stats ]; then
# PMC passes over scripts/prof_objective.py (seven fit steps at cfg3, nothing else): one counter set per pass
pass() {   # name, counters...   (a failed pass ends the script: no partial counters file for bench.py to trust)
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -- python3 scripts/prof_objective.py --workload $WL $EXTRA > /dev/null 2> gpurun_out/pmc_$name.err || { echo "pass $name FAILED"; exit 1; }
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass sq3 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE
python scripts/summarize_rocprof.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3 > gpurun_out/${TAG}_rocprof_pmc_${W}.txt
IMAGES=$(python3 -c "import bench; c, f, _ = bench.workload_config('$WL'); print(f * 9)")
python scripts/make_counters_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3 workload=$WL images=$IMAGES channels=$CH mip=$MIP > gpurun_out/${TAG}_counters_${W}.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3
# the bench line pairs its own timings with these counters (same kernel sources, durations within 10 %): it reads the file from profiles/
cp gpurun_out/${TAG}_counters_${W}.json profiles/${TAG}_counters_${W}.json
echo "pmc done"
fi

if [ "$MODE" != quick ]; then
  if [ "$MODE" != stats ]; then
  if [ "$W" = cfg3 ]; then      # the default line, as the driver runs it
    timeout -k 10 500 python bench.py > gpurun_out/${TAG}_bench_${W}.json 2> gpurun_out/${TAG}_bench_${W}.err
  else
    timeout -k 10 500 python bench.py --workload $WL $EXTRA --no-cpu-baseline --no-reference-shaped-step > gpurun_out/${TAG}_bench_${W}.json 2> gpurun_out/${TAG}_bench_${W}.err
  fi
  echo "bench done"
  fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --workload $WL $EXTRA --steps 5 --warmup 2 --no-cpu-baseline --no-reference-shaped-step > gpurun_out/${TAG}_bench_under_rocprof_${W}.json 2> gpurun_out/prof_stats.err
  python scripts/summarize_rocprof.py gpurun_out/prof_stats > gpurun_out/${TAG}_rocprof_stats_${W}.txt
  rm -rf gpurun_out/prof_stats
  echo "stats done"
  # (the profiled run leaves out the one-image reference-shaped steps: their launches of the same kernels would dilute the averages)
fi
