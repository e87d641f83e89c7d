#!/bin/bash
run() {
  timeout -k 10 150 python bench.py "$@" --no-cpu --no-e2e --no-copy --steps 8 --warmup 6 > gpurun_out/cs.json 2> gpurun_out/cs.err || { echo "FAILED: $*"; tail -3 gpurun_out/cs.err; return; }
  python -c "
import json; d=json.load(open('gpurun_out/cs.json')); r=d['roofline']; print('%-70s %8.3f ms  frac %.3f' % ('$*', d['ms_per_step'], r['frac']))"
}
run --length 200 --records 20000000
run --length 200 --records 20000000 --workload uniq
run --length 200 --records 20000000 --workload uniq --hash-only
run --length 150 --records 20000000 --workload uniq --n-frac 0.01
run --length 300 --records 20000000 --workload uniq
