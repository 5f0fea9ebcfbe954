#!/bin/bash
# usage: bash scripts/kernel_avg.sh KERNEL_SUBSTRING [LIBNAME...]   -- mean duration of the full-size launches of one kernel in
# scripts/prof_objective.py (rocprofv3 --kernel-trace), for the default build and each fpc_diffrend_amd/libfpcdr_LIBNAME.so
K=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in "" "$@"; do
  if [ -n "$n" ]; then export FPCDR_LIB_PATH=$PWD/fpc_diffrend_amd/libfpcdr_$n.so; else unset FPCDR_LIB_PATH; fi
  rm -rf gpurun_out/ka
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ka -- python3 scripts/prof_objective.py --ops 0 > /dev/null 2> gpurun_out/ka.err || exit 1
  python - "$K" "${n:-default}" <<'PY'
import csv, glob, sys
k, name = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob("gpurun_out/ka/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            rows.append((int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
g = max(x for x, _ in rows)
sel = [t for x, t in rows if 2 * x >= g]
print("%-10s %s: %d launches, mean %.1f us, min %.1f us" % (name, k, len(sel), sum(sel) / len(sel), min(sel)))
PY
done
rm -rf gpurun_out/ka
