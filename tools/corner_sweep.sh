#!/bin/bash
# GPU box: the combinations of workload / outputs / N / record length the tables do not list -- looking for a slow corner
run() {
  timeout -k 10 150 python bench.py "$@" --no-cpu --no-e2e --no-copy --steps 8 --warmup 6 > gpurun_out/cs.json 2> gpurun_out/cs.err || { echo "FAILED: $*"; tail -3 gpurun_out/cs.err; return; }
  python -c "
import json; d=json.load(open('gpurun_out/cs.json')); r=d['roofline']; print('%-70s %8.3f ms  frac %.3f' % ('$*', d['ms_per_step'], r['frac']))"
}
run --workload uniq --hash-only --n-frac 0.01
run --workload uniq --n-frac 0.001
run --n-frac 0.001
run --n-frac 0.1
run --workload uniq --n-frac 0.1
run --length 1500 --records 6000000
run --length 1500 --records 6000000 --workload uniq
run --length 1500 --records 6000000 --n-frac 0.01
run --length 1500 --records 6000000 --workload uniq --n-frac 0.01
run --length 200 --records 20000000
run --length 200 --records 20000000 --workload uniq
run --length 200 --records 20000000 --workload uniq --hash-only
run --length 300 --records 20000000 --workload uniq
run --length 2500 --records 4000000
run --length 2500 --records 4000000 --workload uniq
run --length 5000 --records 2000000 --workload uniq --n-frac 0.01
run --workload mixed --with-hash --hash-only
run --workload mixed --with-hash --hash-only --n-frac 0.01
run --workload mixed --max-len 200000
run --workload mixed --max-len 200000 --with-hash
