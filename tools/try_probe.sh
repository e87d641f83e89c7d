#!/bin/bash
# runs a probe script with each libcirckit_hip_<tag>.so variant swapped in (GPU box only): tools/try_probe.sh script.py tag...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
script=$1; shift
for v in "$@"; do
  export CIRCKIT_LIB=$(pwd)/circkit_amd/libcirckit_hip_$v.so     # read by circkit_amd/api.py; the in-tree library stays as built
  echo "== $v"; timeout -k 10 200 python $script 2>&1 | grep -v amdgpu.ids
done
