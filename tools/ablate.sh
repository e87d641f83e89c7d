#!/bin/bash
# Timing ablations of the streaming kernel (results are WRONG under these flags; timing only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in 0 1 2 4 8 16 32 63 ; do
  echo -n "flags=$f  "
  CIRCKIT_DEBUG_FLAGS=$f python $R/bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms_per_step=%.3f' % d['ms_per_step'])"
done
