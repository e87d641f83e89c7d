#!/bin/bash
# GPU box: `circkit canonicalize` on 5 GB into /dev/null with and without the supervisor process, runs spaced by a second and
# back to back (does the previous worker's teardown, now in the background, slow the next run down?)
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
  python3 -c "import re;t=open('/tmp/cli_timing.txt').read();m=re.search(r'pipeline ([0-9.]+) s',t);print('%-44s %.3f s wall -> %.2f M records/s   pipeline %s' % ('$label', $e - $s, $N / ($e - $s) / 1e6, m.group(1) if m else '?'))"
}
for rep in 1 2 3; do run "supervisor, spaced" $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o /dev/null; sleep 1; done
for rep in 1 2 3; do run "supervisor, back to back" $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o /dev/null; done
sleep 1
for rep in 1 2 3; do CIRCKIT_CLI_NO_SUPERVISOR=1 run "one process" $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o /dev/null; done
sleep 1
for rep in 1 2; do run "supervisor, file, spaced" $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o /dev/shm/out.fasta; sleep 1; rm -f /dev/shm/out.fasta; done
for rep in 1 2; do CIRCKIT_CLI_NO_SUPERVISOR=1 run "one process, file" $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o /dev/shm/out.fasta; rm -f /dev/shm/out.fasta; done
rm -f /dev/shm/in.fasta /dev/shm/out.fasta
