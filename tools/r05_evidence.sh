#!/bin/bash
# round 5 evidence run (GPU box): everything DESIGN.md / profiles/r05 quote, on the final code.  usage: tools/r05_evidence.sh [part ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r05_evidence
mkdir -p $O
cd $R
PARTS=${@:-"prof trip other timeline shard skewed c2 lists"}
for P in $PARTS; do
case $P in
tests)  timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -14 $O/pytest_gpu.log ;;
prof)   timeout 1800 bash tools/collect_profiles.sh r05_evidence/prof > $O/collect.log 2>&1; tail -3 $O/collect.log ;;
trip)   FFH_LIBRARY=$R/flashfry_amd/lib/ab/trip.so timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 2 --warmup 1 2> $O/trip_stats.err > /dev/null
        grep "trip stats" $O/trip_stats.err | tail -1 | tee $O/compare_trip_stats.txt ;;
other)  timeout 900 bash tools/pmc_other_kernels.sh > $O/pmc_other_kernels.txt 2>&1; tail -40 $O/pmc_other_kernels.txt ;;
timeline) timeout 300 bash tools/timeline.sh > $O/timeline_step.txt 2>&1; timeout 300 bash tools/timeline.sh --targets 4.5e6 --guides 1000 > $O/timeline_c2.txt 2>&1
        timeout 300 bash tools/timeline_lists.sh > $O/timeline_lists.txt 2>&1; tail -4 $O/timeline_step.txt ;;
shard)  for n in 1 2 4 8; do timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) --comm 2>/dev/null | grep '^{'; done | tee $O/shard_step.txt ;;
skewed) timeout 600 bash tools/skewed_timeline.sh > $O/skewed_timeline.txt 2>&1; tail -30 $O/skewed_timeline.txt ;;
c2)     timeout 300 python bench.py --targets 4.5e6 --guides 1000 --steps 50 --warmup 5 --cpu-seconds 10 --no-skewed --no-c2 > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.json ;;
lists)  timeout 600 python tools/lists_ab.py 2>/dev/null | tail -1 | tee $O/lists.json ;;
grid)   timeout 1500 python tools/timing_grid.py > $O/timing_grid.md 2> $O/timing_grid.err; tail -12 $O/timing_grid.md ;;
r03)    timeout 900 python tools/ingest_scale.py > $O/ingest_scale.txt 2>&1; tail -5 $O/ingest_scale.txt; timeout 900 python tools/cli_wall.py > $O/cli_wall.txt 2>&1; tail -8 $O/cli_wall.txt ;;
stress) timeout 2300 bash tools/stress_sweep.sh ${STRESS_SECS:-1200} 5 gpurun_out/r05_evidence/stress ;;
esac
done
ls -la $O
