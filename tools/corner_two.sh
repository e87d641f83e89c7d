#!/bin/bash
# GPU box: bytes-only batches of two-word records by workgroup size of the two-word streaming build
for lib in "$@"; do
  if [ "$lib" = base ]; then unset CIRCKIT_LIB; else export CIRCKIT_LIB=$PWD/circkit_amd/libcirckit_hip_$lib.so; fi
  for L in 1200 1500 2000; do
    timeout -k 10 150 python bench.py --length $L --records $((9000000000 / L)) --no-cpu --no-e2e --no-copy --steps 8 --warmup 6 > gpurun_out/ct.json 2> gpurun_out/ct.err || { echo "FAILED $lib $L"; tail -3 gpurun_out/ct.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/ct.json')); r=d['roofline']; print('%-8s L=%-5d %8.3f ms  frac %.3f' % ('$lib', $L, d['ms_per_step'], r['frac']))"
  done
done
