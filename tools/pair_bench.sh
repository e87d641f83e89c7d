#!/bin/bash
# GPU box: the uniq step with canonical bytes and hash-only (bench lines, no CPU leg) -- the quick look at a streaming-build change
for w in "" "--hash-only"; do
  timeout -k 10 120 python bench.py --workload uniq $w --no-cpu --no-e2e > gpurun_out/pb.json 2> gpurun_out/pb.err || { echo "bench failed"; tail -5 gpurun_out/pb.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/pb.json')); r=d['roofline']; print('uniq $w', round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'kernel_ms', r.get('kernel_ms'))"
done
