#!/bin/bash
# usage (GPU box, repo root): tools/prof_run.sh <tag> [bench args...]  -> gpurun_out/<tag>_kernel_stats.csv
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python $R/bench.py --no-cpu --no-cli --no-others "$@" > $R/gpurun_out/$TAG.log 2>&1
f=$(find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
cat $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
