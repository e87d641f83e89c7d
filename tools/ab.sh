#!/bin/bash
# tools/ab.sh ROUNDS tag1 tag2 ... : round-robin timing of libcirckit_hip_<tag>.so variants (GPU box only), so that
# clock / thermal drift hits every variant alike; prints min and median ms per variant.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rounds=$1; shift
: > gpurun_out/ab_raw.txt
for r in $(seq $rounds); do
  for v in "$@"; do
    export CIRCKIT_LIB=$R/circkit_amd/libcirckit_hip_$v.so     # read by circkit_amd/api.py; the in-tree library stays as built
    ms=$(python bench.py --steps 10 --warmup 2 --no-cpu --no-e2e --no-copy ${AB_ARGS} 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['roofline']['kernel_ms'])")
    echo "$v $ms" >> gpurun_out/ab_raw.txt
  done
done
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/ab_raw.txt"):
    v, ms = l.split(); d[v].append(float(ms))
for v, x in d.items():
    print("%-12s min %.3f  median %.3f  max %.3f  (n=%d)" % (v, min(x), statistics.median(x), max(x), len(x)))
PY
