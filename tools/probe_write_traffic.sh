#!/bin/bash
# GPU box: WRITE_SIZE of the mixed-length N kernel per launch at several N fractions (does the write traffic beyond the payload
# follow the share of records with an N in the winning window?)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for f in 0.001 0.003 0.01 0.03; do
  rm -rf $R/gpurun_out/wt_$f
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/wt_$f -- python $R/bench.py --workload mixed --n-frac $f --steps 2 --warmup 1 --no-cpu --no-e2e --no-copy --no-cli --no-others > /dev/null 2>&1
  echo "n-frac $f: $(python $R/tools/pmc_summary.py $R/gpurun_out/wt_$f | grep "canon_mixed_n_kernel\|canon_kernel<4>" | sed 's/(anonymous namespace):://; s/(ck::CanonArgs.*)  *WRITE/ WRITE/' | tr '\n' ';')"
done
