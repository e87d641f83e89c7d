#!/bin/bash
# Times bench.py with each libcirckit_hip_<tag>.so variant swapped in (GPU box only; restores nothing: scratch copy).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in "$@"; do
  export CIRCKIT_LIB=$(pwd)/circkit_amd/libcirckit_hip_$v.so     # read by circkit_amd/api.py; the in-tree library stays as built
  for i in 1 2; do
    echo -n "$v: "; python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms_per_step=%.3f frac=%.3f' % (d['ms_per_step'], d['roofline']['frac']))"
  done
done
