#!/bin/bash
# tools/prof_script.sh <tag> <python script> : rocprofv3 kernel stats of a probe script (GPU box), top kernels printed
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python $R/$1 > $R/gpurun_out/$TAG.log 2>&1
f=$(find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:8]:
    print("  %-64s calls %4s avg %9.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
