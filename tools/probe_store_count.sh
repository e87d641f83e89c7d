#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for cfg in "500 3000 0.01" "500 3000 0.0" "100 3000 0.01" "1000 3000 0.01" "500 9000 0.01"; do
  set -- $cfg
  rm -rf $R/gpurun_out/sc
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_WR WRITE_SIZE --output-format csv -d $R/gpurun_out/sc -- python3 $R/tools/probe_store_count.py $1 $2 $3 > /tmp/sc.log 2>&1
  echo "short $1 long $2 nfrac $3: $(grep records /tmp/sc.log)"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/sc | grep "canon_mixed" | sed 's/(anonymous namespace):://; s/(ck::CanonArgs[^)]*)//' | awk '{print "   ", $1, $(NF-2), $NF}'
done
