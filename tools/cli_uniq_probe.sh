#!/bin/bash
# GPU box: `circkit uniq` (default: hashes of the canonical forms, original records out) and `uniq --canonicalize` on 5M x 1 kb
# (every record distinct: all of them are written) next to `canonicalize`, into /dev/null; wall + the binary's stage clock
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
run() {
  label=$1; shift
  s=$(date +%s.%N); CIRCKIT_CLI_TIMING=1 "$@" 2> /tmp/cli_timing.txt; e=$(date +%s.%N)
  python3 -c "print('%-34s %.3f s wall -> %.2f M records/s' % ('$label', $e - $s, $N / ($e - $s) / 1e6))"
  grep "busy" /tmp/cli_timing.txt | sed "s/^/    /" | cut -c1-250
}
for rep in 1 2; do
  run "canonicalize" $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o /dev/null
  run "uniq" $R/circkit_amd/circkit uniq /dev/shm/in.fasta -o /dev/null
  run "uniq --canonicalize" $R/circkit_amd/circkit uniq --canonicalize /dev/shm/in.fasta -o /dev/null
done
rm -f /dev/shm/in.fasta
