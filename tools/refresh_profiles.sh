#!/bin/bash
# GPU box, repo root: regenerates the judged artefacts under gpurun_out/refresh/ (copy them into profiles/ afterwards).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
tools/pmc_run.sh refresh_pmc > /dev/null
cp gpurun_out/refresh_pmc.summary.txt $O/r01_pmc_counters.txt
python tools/make_traffic.py $O/r01_pmc_counters.txt "StreamCfg<16, 2, 1, 1>, false, false" $O/traffic.json > /dev/null
cp $O/traffic.json profiles/traffic.json        # so that the bench line below carries the fresh figure
tools/prof_run.sh refresh_stats > /dev/null
cp gpurun_out/refresh_stats_kernel_stats.csv $O/r01_kernel_stats.csv
tools/prof_run.sh refresh_stats_uniq --workload uniq > /dev/null
cp gpurun_out/refresh_stats_uniq_kernel_stats.csv $O/r01_kernel_stats_uniq.csv
tools/prof_run.sh refresh_stats_mixed --workload mixed > /dev/null
cp gpurun_out/refresh_stats_mixed_kernel_stats.csv $O/r01_kernel_stats_mixed.csv
python bench.py > $O/r01_bench.json
python bench.py --workload uniq --no-cpu > $O/r01_bench_uniq.json
python bench.py --workload mixed --no-cpu > $O/r01_bench_mixed.json
cat $O/r01_bench.json; head -3 $O/r01_kernel_stats.csv | cut -c1-160
