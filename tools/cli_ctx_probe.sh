#!/bin/bash
# GPU box: `circkit canonicalize` on 5 GB into /dev/null and a tmpfs file with one and with two contexts taking alternate chunks
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-5000000}
python3 - <<PY
import numpy as np
rng = np.random.default_rng(1)
N, L = $N, 1000
with open("/dev/shm/in.fasta", "wb") as f:
    for s in range(0, N, 100000):
        m = min(100000, N - s)
        blk = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(m, L))]
        f.write(b"".join(b">r%d\n" % (s + i) + blk[i].tobytes() + b"\n" for i in range(m)))
PY
for rep in 1 2 3; do
  for k in 1 2; do
    for sink in /dev/null /dev/shm/out.fasta; do
      rm -f /dev/shm/out.fasta
      s=$(date +%s.%N); CIRCKIT_CLI_CTXS=$k CIRCKIT_CLI_TIMING=1 $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o $sink 2> /tmp/t.txt; e=$(date +%s.%N)
      python3 -c "print('ctxs %d  %-20s wall %.3f s = %.2f M records/s' % ($k, '$sink', $e - $s, $N / ($e - $s) / 1e6), end='  ')"
      grep "main() to" /tmp/t.txt | sed 's/.*pipeline \([0-9.]*\) s.*gpu \([0-9.]*\) .*file write \([0-9.]*\)/pipeline \1  gpu busy \2  file write \3/'
    done
  done
done
rm -f /dev/shm/in.fasta /dev/shm/out.fasta
