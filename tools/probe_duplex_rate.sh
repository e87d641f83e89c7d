#!/bin/bash
# GPU box: how often does a fresh process end up with the two directions of its host batches NOT overlapping?
# (tools/probe_stream_luck.py K, RUNS fresh processes; a 1 GB call takes ~24.8 ms when they overlap, 38-40 when not)
R=${GRAFT_REPO_ROOT:-$(pwd)}
runs=${1:-40}; k=${2:-0}
fast=0; slow=0
for i in $(seq $runs); do
  ms=$(timeout -k 10 120 python $R/tools/probe_stream_luck.py $k 2>&1 | grep -o "streams: [0-9.]* ms" | grep -o "[0-9.]*" | head -1)
  if python3 -c "import sys; sys.exit(0 if float('$ms') < 30 else 1)"; then fast=$((fast+1)); else slow=$((slow+1)); echo "slow: $ms ms (run $i)"; fi
done
echo "K=$k: $fast fast, $slow slow of $runs"
