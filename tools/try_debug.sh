#!/bin/bash
# runs tools/gpu_debug_small.py with each libcirckit_hip_<tag>.so variant swapped in (GPU box only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in "$@"; do
  cp circkit_amd/libcirckit_hip_$v.so circkit_amd/libcirckit_hip.so
  echo "== $v"; timeout -k 10 100 python tools/gpu_debug_small.py 2>&1 | grep "records"
done
