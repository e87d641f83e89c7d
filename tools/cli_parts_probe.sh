#!/bin/bash
# GPU box: the CLI's GPU stage with host batches in one part / in parts (5M x 1 kb into /dev/null)
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 - <<PY
import numpy as np
rng = np.random.default_rng(1)
N, L = 5000000, 1000
with open("/dev/shm/in.fasta", "wb") as f:
    for s in range(0, N, 100000):
        m = min(100000, N - s)
        blk = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(m, L))]
        f.write(b"".join(b">r%d\n" % (s + i) + blk[i].tobytes() + b"\n" for i in range(m)))
PY
for p in 1 2 4 8 1 4; do
  echo "== CIRCKIT_HOST_BATCH_PARTS=$p"
  CIRCKIT_HOST_BATCH_PARTS=$p CIRCKIT_CLI_TIMING=1 $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o /dev/null 2>&1 | grep busy | sed 's/.*pipeline/pipeline/'
done
rm -f /dev/shm/in.fasta
