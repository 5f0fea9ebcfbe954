# usage: bash scripts/ab.sh NAME...   -- bench each fpc_diffrend_amd/libfpcdr_NAME.so (and the default build first)
mkdir -p gpurun_out
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_default.json 2>/dev/null && python scripts/kt.py gpurun_out/ab_default.json
for n in "$@"; do
  FPCDR_LIB_PATH=$PWD/fpc_diffrend_amd/libfpcdr_$n.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$n.json 2>/dev/null && python scripts/kt.py gpurun_out/ab_$n.json
done
