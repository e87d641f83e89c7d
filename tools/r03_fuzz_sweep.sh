#!/bin/bash
# tools/r03_fuzz_sweep.sh (GPU box): every profile of tools/gpu_fuzz.py on fresh seeds; stops at the first mismatch.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
out=gpurun_out/fuzz_sweep.txt
: > $out
run() { echo "== seed $1 count $2 profile $3" >> $out; timeout -k 10 400 python tools/gpu_fuzz.py $1 $2 $3 >> $out 2>&1 || { echo FAILED >> $out; tail -5 $out; exit 1; }; tail -1 $out; }
B=${FUZZ_SEED:-7000}
run $((B+1)) 300000 short && run $((B+2)) 300000 short && run $((B+3)) 150000 long && run $((B+4)) 200000 nrich && \
run $((B+5)) 30000 longn && run $((B+6)) 30000 prefixn && run $((B+7)) 400 team && run $((B+8)) 20000 leanties && \
run $((B+9)) 20000 leanties && run $((B+10)) 30000 longn && echo "sweep done" | tee -a $out
