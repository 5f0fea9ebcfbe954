#!/bin/bash
# usage: bash scripts/prof_variants.sh NAME...  -- rocprofv3 kernel stats of scripts/prof_objective.py for the default build and each variant
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for n in default "$@"; do
  if [ "$n" = default ]; then unset FPCDR_LIB_PATH; else export FPCDR_LIB_PATH=$PWD/fpc_diffrend_amd/libfpcdr_$n.so; fi
  rm -rf gpurun_out/pv_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pv_$n -- python3 scripts/prof_objective.py --ops 0 > gpurun_out/pv_$n.json 2> gpurun_out/pv_$n.err || echo "FAILED $n"
  python scripts/summarize_rocprof.py gpurun_out/pv_$n | grep -E "k_bins|k_aa_fix|k_render_aa_bwd|k_setup|k_sil2|k_occ|k_shade|k_fix|k_list|k_init|k_tex_reduce|k_objective" > gpurun_out/pv_$n.txt
  rm -rf gpurun_out/pv_$n
  echo "== $n $(cat gpurun_out/pv_$n.json)"; cat gpurun_out/pv_$n.txt
done
