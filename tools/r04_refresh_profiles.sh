#!/bin/bash
# GPU box, repo root: regenerates the judged artefacts of round 4 under gpurun_out/refresh/ (copy them into profiles/).
# Per workload: the PMC passes (traffic.json quotes them), one bench line, one rocprofv3 --kernel-trace --stats summary.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
pmc() {   # tag, bench args...
  tag=$1; shift
  tools/pmc_run.sh refresh_pmc_$tag "$@" > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/refresh_pmc_$tag > $O/r04_pmc_counters_$tag.txt
  echo "pmc $tag: $(wc -l < $O/r04_pmc_counters_$tag.txt) lines"
}
run() {   # tag, bench args...
  tag=$1; shift
  python bench.py "$@" > $O/r04_bench_$tag.json 2> $O/r04_bench_$tag.err || { echo "bench $tag failed"; tail -3 $O/r04_bench_$tag.err; }
  tools/prof_run.sh refresh_$tag "$@" --no-e2e --no-copy > /dev/null 2>&1
  cp gpurun_out/refresh_${tag}_kernel_stats.csv $O/r04_kernel_stats_$tag.csv
  python -c "import json;d=json.load(open('$O/r04_bench_$tag.json'));r=d['roofline'];print('$tag', round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'of copy', round(r.get('frac_of_copy',0),3), 'traffic', r['traffic'])"
}
pmc canonicalize
pmc uniq --workload uniq
pmc mixed --workload mixed
pmc canonicalize_n1pct --n-frac 0.01
pmc mixed_n1pct --workload mixed --n-frac 0.01
pmc mixed_uniq --workload mixed --with-hash
pmc uniq_hash_only --workload uniq --hash-only
pmc mixed_uniq_n1pct --workload mixed --with-hash --n-frac 0.01
python tools/make_traffic.py $O/traffic.json \
  $O/r04_pmc_counters_canonicalize.txt "StreamCfg<16, 2, 1, 1>, false, false, false, false" "canonicalize 10000000 x 1000" 10000000 \
  $O/r04_pmc_counters_uniq.txt "StreamCfg<8, 2, 2, 1>, true, false, false, false" "uniq 10000000 x 1000" 10000000 \
  $O/r04_pmc_counters_mixed.txt "canon_mixed_kernel" "mixed 1000000 x 1000" 1000000 \
  $O/r04_pmc_counters_canonicalize_n1pct.txt "StreamCfg<4, 2, 1, 1>, false, false, false, true" "canonicalize 10000000 x 1000 n0.01" 10000000 \
  $O/r04_pmc_counters_mixed_n1pct.txt "canon_mixed_n_kernel" "mixed 1000000 x 1000 n0.01" 1000000 \
  $O/r04_pmc_counters_mixed_uniq.txt "canon_mixed_h_kernel" "mixed 1000000 x 1000 hash" 1000000 \
  $O/r04_pmc_counters_uniq_hash_only.txt "StreamCfg<8, 2, 2, 1>, true, false, false, false" "uniq 10000000 x 1000 hash-only" 10000000 \
  $O/r04_pmc_counters_mixed_uniq_n1pct.txt "canon_mixed_nh_kernel" "mixed 1000000 x 1000 n0.01 hash" 1000000 > /dev/null
cp $O/traffic.json $R/profiles/traffic.json
run canonicalize
run uniq --workload uniq
run mixed --workload mixed
run canonicalize_n1pct --n-frac 0.01
run mixed_n1pct --workload mixed --n-frac 0.01
run mixed_uniq --workload mixed --with-hash
run uniq_hash_only --workload uniq --hash-only
run mixed_uniq_n1pct --workload mixed --with-hash --n-frac 0.01
# the multi-GPU uniq exchange rehearsed on one GPU (launcher + RCCL, one rank)
CIRCKIT_BENCH_FORCE_DIST=1 python bench.py --workload uniq --no-cpu --no-e2e > $O/r04_bench_uniq_forced_exchange.json 2>/dev/null
# the same three workloads with the steps dealt to three ctx/stream lanes (bench lines only: overlapping launches make
# per-kernel durations of a trace meaningless)
python bench.py --streams 3 --no-cpu --no-e2e > $O/r04_bench_canonicalize_streams3.json 2>/dev/null
python bench.py --workload uniq --streams 3 --no-cpu --no-e2e > $O/r04_bench_uniq_streams3.json 2>/dev/null
python bench.py --workload mixed --streams 3 --no-cpu --no-e2e > $O/r04_bench_mixed_streams3.json 2>/dev/null
ls $O | wc -l
