#!/bin/bash
# runs a probe script with each libcirckit_hip_<tag>.so variant swapped in (GPU box only): tools/try_probe.sh script.py tag...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
script=$1; shift
for v in "$@"; do
  cp circkit_amd/libcirckit_hip_$v.so circkit_amd/libcirckit_hip.so
  echo "== $v"; timeout -k 10 200 python $script 2>&1 | grep -v amdgpu.ids
done
