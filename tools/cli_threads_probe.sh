#!/bin/bash
# GPU box: the CLI's device stage against the number of parser / emit threads (CPU and memory contention), 5 GB into /dev/null.
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
nproc
for th in 16 12 8 6 4 16 12 8; do
  s=$(date +%s.%N); CIRCKIT_CLI_TIMING=1 $R/circkit_amd/circkit canonicalize --threads $th /dev/shm/in.fasta -o /dev/null 2> /tmp/t.txt; e=$(date +%s.%N)
  python3 -c "print('threads %2d  wall %.3f s' % ($th, $e - $s), end='  ')"
  grep "main() to" /tmp/t.txt | sed 's/.*pipeline \([0-9.]*\) s.*parse+pack \([0-9.]*\) .*gpu \([0-9.]*\) .*emit \([0-9.]*\) .*/pipeline \1  parse \2  gpu \3  emit \4/'
done
rm -f /dev/shm/in.fasta
