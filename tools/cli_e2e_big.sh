#!/bin/bash
# Steady-state CLI throughput: N x 1 kb synthetic FASTA in tmpfs (default 5M records = 5 GB).
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
ls -la /dev/shm/in.fasta
# (a fresh output file each time: truncating the previous run's 5 GB in tmpfs costs 0.2-0.3 s that are not the tool's)
for t in 0 4 1; do rm -f /dev/shm/out.fasta; s=$(date +%s.%N); CIRCKIT_CLI_TIMING=1 $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o /dev/shm/out.fasta $([ $t -gt 0 ] && echo -t $t); e=$(date +%s.%N); python3 -c "print('canonicalize -t $t (0 = default): %.3f s wall -> %.2f M records/s' % ($e - $s, $N / ($e - $s) / 1e6))"; done
rm -f /dev/shm/out.fasta; s=$(date +%s.%N); $R/circkit_amd/circkit uniq /dev/shm/in.fasta -o /dev/shm/out.fasta; e=$(date +%s.%N); python3 -c "print('uniq (raw output): %.3f s wall -> %.2f M records/s' % ($e - $s, $N / ($e - $s) / 1e6))"
cmp <(head -c 100000000 /dev/shm/in.fasta | grep -c ">") <(head -c 100000000 /dev/shm/out.fasta | grep -c ">") && echo "record counts agree on the first 100 MB"
rm -f /dev/shm/in.fasta /dev/shm/out.fasta
