#!/bin/bash
# The rest of a round's measurement set (run after scripts/measure_round.sh TAG all cfg3 / cfg2 / cfg5 / cfg3 --mip).
#   usage: bash scripts/measure_rest.sh TAG      -> gpurun_out/TAG_* (copy what is to be judged into profiles/)
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
E=gpurun_out/${TAG}_rest.err; : > $E
python bench.py --workload cfg5 --frames-per-gpu 32 --no-cpu-baseline --steps 20 > gpurun_out/${TAG}_bench_cfg5_32frames.json 2>> $E && echo cfg5-32
python bench.py --channels 3 --no-cpu-baseline --no-reference-shaped-step > gpurun_out/${TAG}_bench_cfg3_c3.json 2>> $E && echo c3
python scripts/coverage_sweep.py > gpurun_out/${TAG}_coverage_sweep.json 2>> $E && echo sweep
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 scripts/prof_objective.py --ops 0 > /dev/null 2>> $E; python scripts/step_timeline.py gpurun_out/tl --all > gpurun_out/${TAG}_step_timeline.txt; rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 scripts/prof_objective.py --ops 0 --frames-per-step 1 --views-per-step 1 --workload ref --graph 1 --steps 12 > /dev/null 2>> $E; python scripts/step_timeline.py gpurun_out/tl --all > gpurun_out/${TAG}_ref_step_timeline_graph.txt; rm -rf gpurun_out/tl
# wave-cycles of k_shade by phase: a -DFPCDR_OPROF build of objective.hip (scripts/build_obj_variant.sh oprof -DFPCDR_OPROF, built in the container)
if [ -f fpc_diffrend_amd/libfpcdr_oprof.so ]; then
  FPCDR_LIB_PATH=$PWD/fpc_diffrend_amd/libfpcdr_oprof.so python scripts/prof_phases.py > gpurun_out/${TAG}_shade_phase_cycles.txt 2>> $E && echo oprof
fi
echo done
