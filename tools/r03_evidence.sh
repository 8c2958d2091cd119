#!/bin/bash
# round 3 evidence run (GPU box): everything DESIGN.md / profiles/r03 quote, on the final code.  usage: tools/r03_evidence.sh [part ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03_evidence
mkdir -p $O
cd $R
PARTS=${@:-"prof other timeline shard grid bulge cli ingest c2 wave stress"}
for P in $PARTS; do
case $P in
prof)   timeout 1500 bash tools/collect_profiles.sh r03_evidence/prof > $O/collect.log 2>&1; tail -3 $O/collect.log ;;
other)  timeout 600 bash tools/pmc_other_kernels.sh > $O/pmc_other_kernels.txt 2>&1; tail -30 $O/pmc_other_kernels.txt ;;
timeline) timeout 300 bash tools/timeline.sh > $O/timeline_step.txt 2>&1; timeout 300 bash tools/timeline_lists.sh > $O/timeline_lists.txt 2>&1; tail -4 $O/timeline_step.txt ;;
shard)  for n in 2 4 8; do timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) 2>/dev/null | tail -1; done | tee $O/shard_step.txt
        echo "# the plan a shard got before plan_cost saw its span (10 + 10, radii 1 + 2):" | tee -a $O/shard_step.txt
        timeout 300 python tools/shard_step.py --shards 8 --rank 4 --plan-a 10 --plan-r1 1 2>/dev/null | tail -1 | tee -a $O/shard_step.txt ;;
grid)   timeout 900 python tools/timing_grid.py --out $O/timing_grid.json > $O/timing_grid.md 2> $O/timing_grid.err; tail -22 $O/timing_grid.md ;;
bulge)  timeout 600 python tools/bulge_scale.py --guides 10000 --brute-guides 300 --out $O/bulge_scale.json 2>&1 | tail -4 ;;
cli)    timeout 900 python tools/cli_wall.py --out $O/cli_wall_chr22_scale.json 2>&1 | tail -3; timeout 1500 python tools/cli_wall.py --mbases 3100 --contigs 24 --big-guides 100000 --out $O/cli_wall_hg38_scale.json 2>&1 | tail -3 ;;
ingest) timeout 900 python tools/ingest_scale.py --out $O/ingest_hg38_scale.json 2>&1 | tail -3 ;;
c2)     timeout 300 python bench.py --targets 4.5e6 --guides 1000 --steps 50 --warmup 5 --cpu-seconds 10 --no-skewed --no-c2 > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 1500 $O/bench_c2.json ;;
wave)   # per-wave cycle counts of the compare launch (dev build: tools/build_variant.sh WORK stats_q -DFFH_WAVE_STATS)
        for q in 1 0; do echo "hg38-scale step, FFH_WORK_QUEUE=$q"; FFH_WORK_QUEUE=$q FFH_LIBRARY=$R/flashfry_amd/lib/ab/stats_q.so timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 3 --warmup 1 2>&1 | grep "wave stats" | tail -1; done > $O/wave_stats.txt
        echo "repeat-structured workload, bounded (six slabs)" >> $O/wave_stats.txt
        FFH_LIBRARY=$R/flashfry_amd/lib/ab/stats_q.so timeout 600 python tools/skewed_ab.py 2>&1 | grep "wave stats" | tail -6 >> $O/wave_stats.txt; cat $O/wave_stats.txt ;;
skewed) timeout 600 bash tools/skewed_timeline.sh > $O/skewed_timeline.txt 2>&1; tail -40 $O/skewed_timeline.txt ;;
stress) timeout 1300 python tools/stress_parity.py 1200 > $O/stress_parity.txt 2>&1; tail -3 $O/stress_parity.txt ;;
esac
done
ls -la $O
