#!/bin/bash
# GPU box: bench lines of the N workloads for variant libraries (CIRCKIT_LIB), same box: tools/ruled_ab.sh base <tag> ...
for lib in "$@"; do
  if [ "$lib" = base ]; then unset CIRCKIT_LIB; else export CIRCKIT_LIB=$PWD/circkit_amd/libcirckit_hip_$lib.so; fi
  for args in "--n-frac 0.01" "--n-frac 0.1" "--workload uniq --n-frac 0.1" "--workload mixed --n-frac 0.01" "--workload mixed --n-frac 0.1"; do
    timeout -k 10 150 python bench.py $args --no-cpu --no-e2e --no-copy --steps 10 --warmup 6 > gpurun_out/rab.json 2> gpurun_out/rab.err || { echo "FAILED $lib $args"; tail -3 gpurun_out/rab.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/rab.json')); r=d['roofline']; print('%-8s %-40s %8.3f ms  frac %.3f' % ('$lib', '$args', d['ms_per_step'], r['frac']))"
  done
done
