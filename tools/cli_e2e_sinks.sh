#!/bin/bash
# GPU box: `circkit canonicalize` wall time on N x 1 kb synthetic FASTA in tmpfs (default 5M records = 5 GB) into different
# sinks -- a tmpfs file, /dev/null, a pipe consumer -- to separate the pipeline's own ceiling (parse + H2D + kernels + D2H +
# emit) from the single-file write rate.  CIRCKIT_CLI_TIMING=1 prints the stages' busy seconds.
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
run() {   # label, command...
  label=$1; shift
  s=$(date +%s.%N); CIRCKIT_CLI_TIMING=1 "$@" 2> /tmp/cli_timing.txt; e=$(date +%s.%N)
  python3 -c "print('%-34s %.3f s wall -> %.2f M records/s' % ('$label', $e - $s, $N / ($e - $s) / 1e6))"
  grep "busy\|close output" /tmp/cli_timing.txt | sed "s/^/    /"
}
for rep in 1 2; do
  rm -f /dev/shm/out.fasta
  run "tmpfs file" $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o /dev/shm/out.fasta
  run "/dev/null" $R/circkit_amd/circkit canonicalize /dev/shm/in.fasta -o /dev/null
  run "pipe | cat > /dev/null" bash -c "$R/circkit_amd/circkit canonicalize /dev/shm/in.fasta | cat > /dev/null"
  run "pipe | wc -c" bash -c "$R/circkit_amd/circkit canonicalize /dev/shm/in.fasta | wc -c > /dev/null"
done
rm -f /dev/shm/in.fasta /dev/shm/out.fasta
