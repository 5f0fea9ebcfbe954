cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --workload cfg2 --no-cpu-baseline > gpurun_out/r05_bench_cfg2.json 2> gpurun_out/rest.err && echo cfg2
python bench.py --workload cfg5 --no-cpu-baseline --steps 20 > gpurun_out/r05_bench_cfg5.json 2>> gpurun_out/rest.err && echo cfg5
python bench.py --workload cfg5 --frames-per-gpu 32 --no-cpu-baseline --steps 20 > gpurun_out/r05_bench_cfg5_32frames.json 2>> gpurun_out/rest.err && echo cfg5-32
python bench.py --mip --no-cpu-baseline --no-reference-shaped-step > gpurun_out/r05_bench_cfg3_mip.json 2>> gpurun_out/rest.err && echo mip
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 scripts/prof_objective.py --ops 0 > /dev/null 2>> gpurun_out/rest.err; python scripts/step_timeline.py gpurun_out/tl --all > gpurun_out/r05_step_timeline.txt; rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 scripts/prof_objective.py --ops 0 --frames-per-step 1 --views-per-step 1 --workload ref --graph 1 --steps 12 > /dev/null 2>> gpurun_out/rest.err; python scripts/step_timeline.py gpurun_out/tl --all > gpurun_out/r05_ref_step_timeline_graph.txt; rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pm -- python3 scripts/prof_mip.py > /dev/null 2>> gpurun_out/rest.err; python scripts/summarize_rocprof.py gpurun_out/pm > gpurun_out/r05_rocprof_stats_mip.txt; rm -rf gpurun_out/pm
echo done
