#!/bin/bash
# GPU box, repo root: regenerates the judged artefacts of round 2 under gpurun_out/refresh/ (copy them into profiles/).
# One bench line + one rocprofv3 --kernel-trace --stats summary per workload, and the PMC passes of the headline.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
run() {   # tag, bench args...
  tag=$1; shift
  python bench.py "$@" > $O/r02_bench_$tag.json 2> $O/r02_bench_$tag.err || { echo "bench $tag failed"; tail -3 $O/r02_bench_$tag.err; }
  tools/prof_run.sh refresh_$tag "$@" > /dev/null 2>&1
  cp gpurun_out/refresh_${tag}_kernel_stats.csv $O/r02_kernel_stats_$tag.csv
}
# PMC passes first: the bench lines below quote their traffic (this box's copy of profiles/traffic.json)
tools/pmc_run.sh refresh_pmc > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/refresh_pmc > $O/r02_pmc_counters.txt
tools/pmc_run.sh refresh_pmc_uniq --workload uniq > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/refresh_pmc_uniq > $O/r02_pmc_counters_uniq.txt
tools/pmc_run.sh refresh_pmc_mixed --workload mixed > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/refresh_pmc_mixed > $O/r02_pmc_counters_mixed.txt
python tools/make_traffic.py $O/traffic.json \
  $O/r02_pmc_counters.txt "StreamCfg<16, 2, 1, 1>, false, false, false, false" "canonicalize 10000000 x 1000" 10000000 \
  $O/r02_pmc_counters_uniq.txt "StreamCfg<16, 2, 1, 1>, true, false, false, false" "uniq 10000000 x 1000" 10000000 \
  $O/r02_pmc_counters_mixed.txt "canon_mixed_kernel" "mixed 1000000 x 1000" 1000000 > /dev/null
cp $O/traffic.json $R/profiles/traffic.json
run canonicalize
run uniq --workload uniq
run mixed --workload mixed
run canonicalize_n1pct --n-frac 0.01
run mixed_n1pct --workload mixed --n-frac 0.01
ls -la $O
# the same three workloads with the steps dealt to three ctx/stream lanes (bench lines only: overlapping launches make
# per-kernel durations of a trace meaningless)
python bench.py --streams 3 --no-cpu > $O/r02_bench_canonicalize_streams3.json 2>/dev/null
python bench.py --workload uniq --streams 3 --no-cpu > $O/r02_bench_uniq_streams3.json 2>/dev/null
python bench.py --workload mixed --streams 3 --no-cpu > $O/r02_bench_mixed_streams3.json 2>/dev/null
ls $O | wc -l
