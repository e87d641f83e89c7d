#!/bin/bash
# End-to-end CLI timing on the GPU box: 1M x 1 kb synthetic FASTA (1 GB) from /dev/shm.
R=${GRAFT_REPO_ROOT:-$(pwd)}
python - <<'PY'
import numpy as np
rng = np.random.default_rng(1)
N, L = 1_000_000, 1000
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(N, L))]
with open("/dev/shm/in.fasta", "wb") as f:
    for s in range(0, N, 100000):
        blk = seq[s:s + 100000]
        rows = [b">r%d\n" % (s + i) + blk[i].tobytes() + b"\n" for i in range(len(blk))]
        f.write(b"".join(rows))
PY
ls -la /dev/shm/in.fasta
for i in 1 2; do s=$(date +%s.%N); CIRCKIT_CLI_TIMING=1 $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o /dev/shm/out.fasta; e=$(date +%s.%N); python3 -c "print('canonicalize: %.3f s wall' % ($e - $s))"; done
s=$(date +%s.%N); CIRCKIT_CLI_TIMING=1 $R/circkit_amd/circkit uniq -c /dev/shm/in.fasta -o /dev/shm/out2.fasta --table /dev/shm/t.csv; e=$(date +%s.%N); python3 -c "print('uniq -c: %.3f s wall' % ($e - $s))"
ls -la /dev/shm/out.fasta /dev/shm/out2.fasta /dev/shm/t.csv
rm -f /dev/shm/in.fasta /dev/shm/out.fasta /dev/shm/out2.fasta /dev/shm/t.csv
