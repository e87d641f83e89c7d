#!/bin/bash
# GPU box: the timed steps of bench.py right after a process that held 150 GB of device memory, and again later.
R=${GRAFT_REPO_ROOT:-$(pwd)}
b() { python $R/bench.py --no-cpu --no-e2e "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('   step %.3f ms  kernel %.3f  frac %.4f  copy %.0f GB/s  of copy %.3f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('copy_ceiling_gbps', 0), r.get('frac_of_copy', 0)))"; }
echo "quiet box:"; b; b
for rep in 1 2; do
python - <<PY
import torch
x = [torch.empty(150 * (1 << 28) // 8, dtype=torch.int32, device="cuda").fill_(1) for _ in range(8)]
torch.cuda.synchronize()
PY
echo "right after a process that held 150 GB:"; b; b; b
done
