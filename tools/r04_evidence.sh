#!/bin/bash
# round 4 evidence run (GPU box): everything DESIGN.md / profiles/r04 quote, on the final code.  usage: tools/r04_evidence.sh [part ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r04_evidence
mkdir -p $O
cd $R
PARTS=${@:-"tests prof other timeline shard skewed c2"}
for P in $PARTS; do
case $P in
tests)  timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log ;;
prof)   timeout 1800 bash tools/collect_profiles.sh r04_evidence/prof > $O/collect.log 2>&1; tail -3 $O/collect.log ;;
other)  timeout 600 bash tools/pmc_other_kernels.sh > $O/pmc_other_kernels.txt 2>&1; tail -30 $O/pmc_other_kernels.txt ;;
timeline) timeout 300 bash tools/timeline.sh > $O/timeline_step.txt 2>&1; timeout 300 bash tools/timeline_lists.sh > $O/timeline_lists.txt 2>&1; tail -4 $O/timeline_step.txt ;;
shard)  for n in 1 2 4 8; do timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) --comm 2>/dev/null | grep '^{'; done | tee $O/shard_step.txt ;;
skewed) timeout 600 bash tools/skewed_timeline.sh > $O/skewed_timeline.txt 2>&1; tail -30 $O/skewed_timeline.txt ;;
c2)     timeout 300 python bench.py --targets 4.5e6 --guides 1000 --steps 50 --warmup 5 --cpu-seconds 10 --no-skewed --no-c2 > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.json ;;
stress) timeout 2300 bash tools/stress_sweep.sh ${STRESS_SECS:-1500} 5 gpurun_out/r04_evidence/stress ;;
esac
done
ls -la $O
