#!/bin/bash
# GPU box: what the CLI's exit is made of after a 5 GB run (CIRCKIT_CLI_TIMING=2 tears down piece by piece and times it).
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
for mode in 2 1 2 1; do
  s=$(date +%s.%N); CIRCKIT_CLI_TIMING=$mode $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o /dev/null 2> /tmp/t.txt; e=$(date +%s.%N)
  python3 -c "print('timing mode $mode: wall %.3f s' % ($e - $s))"
  grep "main() to\|teardown" /tmp/t.txt | cut -c1-90
done
rm -f /dev/shm/in.fasta
