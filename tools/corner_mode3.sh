#!/bin/bash
# GPU box: records of 1009..2032 symbols through the two-word streaming build (the library) and through the mixed-length kernels
# (a build with the mode forced to 3), bytes and bytes + XXH3
run() {
  timeout -k 10 150 python bench.py "$@" --no-cpu --no-e2e --no-copy --steps 8 --warmup 6 > gpurun_out/cm.json 2> gpurun_out/cm.err || { echo "FAILED: $*"; tail -3 gpurun_out/cm.err; return; }
  python -c "
import json; d=json.load(open('gpurun_out/cm.json')); r=d['roofline']; print('%-8s %-60s %8.3f ms  frac %.3f' % ('$LIBTAG', '$*', d['ms_per_step'], r['frac']))"
}
for lib in base mode3; do
  LIBTAG=$lib
  if [ "$lib" = base ]; then unset CIRCKIT_LIB; else export CIRCKIT_LIB=$PWD/circkit_amd/libcirckit_hip_$lib.so; fi
  run --length 1200 --records 7000000
  run --length 1500 --records 6000000
  run --length 2000 --records 4500000
  run --length 1200 --records 7000000 --workload uniq
  run --length 1500 --records 6000000 --workload uniq
  run --length 2000 --records 4500000 --workload uniq
done
