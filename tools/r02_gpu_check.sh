#!/bin/bash
# GPU box: the round's standard check -- GPU tests, then one bench line per workload -> gpurun_out/r02_*
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02_pytest.log || { cat gpurun_out/r02_pytest.log; exit 1; }
cat gpurun_out/r02_pytest.log
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err && cat gpurun_out/r02_bench.json
python bench.py --workload uniq > gpurun_out/r02_bench_uniq.json 2> gpurun_out/r02_bench_uniq.err && cat gpurun_out/r02_bench_uniq.json
python bench.py --workload mixed > gpurun_out/r02_bench_mixed.json 2> gpurun_out/r02_bench_mixed.err && cat gpurun_out/r02_bench_mixed.json
