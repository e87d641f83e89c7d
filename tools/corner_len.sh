#!/bin/bash
# GPU box: canonicalize (bytes only) by record length, libraries through CIRCKIT_LIB
for lib in "$@"; do
  if [ "$lib" = base ]; then unset CIRCKIT_LIB; else export CIRCKIT_LIB=$PWD/circkit_amd/libcirckit_hip_$lib.so; fi
  for L in 100 200 300 400 500 700; do
    timeout -k 10 150 python bench.py --length $L --records $((4000000000 / L)) --no-cpu --no-e2e --no-copy --steps 8 --warmup 6 > gpurun_out/cl.json 2> gpurun_out/cl.err || { echo "FAILED $lib $L"; tail -3 gpurun_out/cl.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/cl.json')); r=d['roofline']; print('%-10s L=%-5d %8.3f ms  frac %.3f  %.2f G records/s' % ('$lib', $L, d['ms_per_step'], r['frac'], d['value']/1e9))"
  done
done
