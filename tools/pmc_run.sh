#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <tag> [bench args...]
# Collects the SQ / TCC counter passes for bench.py separately (rocprofv3 --pmc only; no trace domains).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-copy --no-cli --no-others $*"
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/$TAG/p1 -- $B > $R/gpurun_out/$TAG.p1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/$TAG/p2 -- $B > $R/gpurun_out/$TAG.p2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/p3 -- $B > $R/gpurun_out/$TAG.p3.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/p4 -- $B > $R/gpurun_out/$TAG.p4.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/$TAG | grep -E "canon_stream_kernel|canon_fast_kernel|canon_kernel" > $R/gpurun_out/$TAG.summary.txt
cat $R/gpurun_out/$TAG.summary.txt
