#!/bin/bash
# GPU box: does a process that has just released a lot of device memory change how the NEXT process's host batches overlap their
# two copy directions?  (The first bench.py after a `pytest -m gpu` run showed 38 ms per 1 GB call instead of 23.7, twice.)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for gb in 0 40 150 150; do
  python - <<PY
import torch
if $gb:
    x = [torch.empty($gb * (1 << 28) // 8, dtype=torch.int32, device="cuda").fill_(1) for _ in range(8)]
    torch.cuda.synchronize()
PY
  for i in 1 2 3; do
    echo "after a process that held $gb GB, run $i: $(timeout -k 10 120 python $R/tools/probe_stream_luck.py 0 2>&1 | grep -o 'streams: [0-9.]* ms')"
  done
done
