#!/bin/bash
# runs tools/gpu_debug_small.py with each libcirckit_hip_<tag>.so variant swapped in (GPU box only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in "$@"; do
  export CIRCKIT_LIB=$(pwd)/circkit_amd/libcirckit_hip_$v.so     # read by circkit_amd/api.py; the in-tree library stays as built
  echo "== $v"; timeout -k 10 100 python tools/gpu_debug_small.py 2>&1 | grep "records"
done
