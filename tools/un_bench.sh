#!/bin/bash
# GPU box: uniq on the fixed-length batch with 1 % N (bench line), libraries through CIRCKIT_LIB
for lib in "$@"; do
  if [ "$lib" = base ]; then unset CIRCKIT_LIB; else export CIRCKIT_LIB=$PWD/circkit_amd/libcirckit_hip_$lib.so; fi
  for rep in 1 2; do
    timeout -k 10 120 python bench.py --workload uniq --n-frac 0.01 --no-cpu --no-e2e --no-copy > gpurun_out/ub.json 2> gpurun_out/ub.err || { echo "bench failed"; tail -5 gpurun_out/ub.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ub.json')); r=d['roofline']; print('$lib uniq n1pct', round(d['ms_per_step'],3), 'frac', round(r['frac'],4))"
  done
done
