#!/bin/bash
# GPU box: LDS counters of the headline kernel for library variants.  usage: tools/r02_pmc_lds.sh "<variants>"
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in $1; do
  export CIRCKIT_LIB=$R/circkit_amd/libcirckit_hip_$v.so     # read by circkit_amd/api.py; the in-tree library stays as built
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pl_$v -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/pl_$v.log 2>&1)
  echo "== $v"; python tools/pmc_summary.py gpurun_out/pl_$v | grep "StreamCfg<16, 2, 1, 1>, false, false, false" | awk '{print $(NF-2), $NF}'
done
