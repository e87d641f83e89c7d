#!/bin/bash
# tools/ab_prof.sh tag... : rocprofv3 kernel stats of `bench.py $AB_ARGS` for each library variant (GPU box only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in "$@"; do
  export CIRCKIT_LIB=$(pwd)/circkit_amd/libcirckit_hip_$v.so     # read by circkit_amd/api.py; the in-tree library stays as built
  tools/prof_run.sh abprof_$v ${AB_ARGS} > /dev/null 2>&1
  echo "== $v"
  python - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/abprof_${v}_kernel_stats.csv")))[:6]:
    if "canon" in r["Name"] or "xxh3" in r["Name"] or "uniq" in r["Name"]:
        print("  %-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
