#!/bin/bash
# GPU box: the round's standard check -- GPU tests, then one bench line per workload -> gpurun_out/r02_*
# (full pytest output goes to a file as it is produced: a crash must not take the name of the running test with it)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -v ${PYTEST_ARGS} > gpurun_out/r02_pytest.log 2>&1
rc=$?
tail -25 gpurun_out/r02_pytest.log
[ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err && cat gpurun_out/r02_bench.json &&
python bench.py --workload uniq > gpurun_out/r02_bench_uniq.json 2> gpurun_out/r02_bench_uniq.err && cat gpurun_out/r02_bench_uniq.json &&
python bench.py --workload mixed > gpurun_out/r02_bench_mixed.json 2> gpurun_out/r02_bench_mixed.err && cat gpurun_out/r02_bench_mixed.json
python bench.py --n-frac 0.01 > gpurun_out/r02_bench_n1pct.json 2> gpurun_out/r02_bench_n1pct.err && cat gpurun_out/r02_bench_n1pct.json &&
python bench.py --workload mixed --n-frac 0.01 > gpurun_out/r02_bench_mixed_n1pct.json 2> gpurun_out/r02_bench_mixed_n1pct.err && cat gpurun_out/r02_bench_mixed_n1pct.json
