#!/bin/bash
# GPU box: per-kernel time of library variants on one box.  usage: tools/r02_prof_variants.sh "<variants>" <kernel-substring> [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
variants=$1; pat=$2; shift 2
for v in $variants; do
  export CIRCKIT_LIB=$R/circkit_amd/libcirckit_hip_$v.so     # read by circkit_amd/api.py; the in-tree library stays as built
  tools/prof_run.sh pv_$v "$@" > /dev/null 2>&1
  echo "== $v"; grep -E "$pat" gpurun_out/pv_${v}_kernel_stats.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-60,100-400 | awk -F, '{printf "%-70s calls %s avg %.1f us\n", substr($1,1,70), $(NF-2), $(NF)/1000}'
done
