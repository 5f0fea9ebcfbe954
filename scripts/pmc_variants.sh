#!/bin/bash
# usage: bash scripts/pmc_variants.sh "COUNTERS..." NAME...  -- one rocprofv3 --pmc pass of scripts/prof_objective.py per library variant
# (default = the in-tree build): mean counter values per dispatch of the objective's kernels + the HIP-event times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CTRS=$1; shift
for n in "$@"; do
  if [ "$n" = default ]; then unset FPCDR_LIB_PATH; else export FPCDR_LIB_PATH=$PWD/fpc_diffrend_amd/libfpcdr_$n.so; fi
  rm -rf gpurun_out/pmcv_$n
  timeout -k 10 200 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d gpurun_out/pmcv_$n -- python3 scripts/prof_objective.py --ops 0 > gpurun_out/pmcv_$n.json 2> gpurun_out/pmcv_$n.err || { echo "FAILED $n"; exit 1; }
  echo "== $n $(tail -1 gpurun_out/pmcv_$n.json)"
  python scripts/summarize_rocprof.py gpurun_out/pmcv_$n | grep -E "k_bins_list|k_aa_fix_list|k_render_aa_bwd|k_setup|k_sil2|k_shade|k_fix"
  rm -rf gpurun_out/pmcv_$n
done
