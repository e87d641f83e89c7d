#!/bin/bash
# GPU box: CLI end to end on plasmid-sized records (log-uniform 1..300 kb, ~1.5 GB FASTA in tmpfs), checked against the
# oracle's CLI restatement on the first records.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 - <<PY
import numpy as np
rng = np.random.default_rng(2)
N = 30000
lens = np.exp(np.log(1000) + rng.random(N) * (np.log(300000) - np.log(1000))).astype(np.int64)
with open("/dev/shm/long.fasta", "wb") as f:
    for i, L in enumerate(lens):
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(L))].tobytes()
        f.write(b">p%d\n" % i); f.write(s); f.write(b"\n")
print("records", N, "bases", int(lens.sum()))
PY
rm -f /dev/shm/long.out; s=$(date +%s.%N); CIRCKIT_CLI_TIMING=1 $R/circkit_amd/circkit canonicalize /dev/shm/long.fasta -o /dev/shm/long.out; e=$(date +%s.%N)
python3 -c "print('canonicalize: %.3f s wall' % ($e - $s))"
python3 - <<PY
import sys
sys.path.insert(0, "$R")
from oracle import oracle as O
O.build()
data = open("/dev/shm/long.fasta", "rb").read(60_000_000)
cut = data.rfind(b"\n>")
exp = O.cli_canonicalize(data[:cut + 1])
got = open("/dev/shm/long.out", "rb").read(len(exp))
print("first %d output bytes equal the oracle's: %s" % (len(exp), got == exp))
PY
rm -f /dev/shm/long.fasta /dev/shm/long.out
