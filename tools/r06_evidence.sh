#!/bin/bash
# round 6 evidence run (GPU box): everything DESIGN.md / profiles/r06 quote, on the final code, in the order that matters if the call is cut
# short.  usage: tools/r06_evidence.sh [part ...]     (default: every part; each part has its own timeout)
#   tests     the -m gpu suite (the new tests of round 6 among them: heapwatch, slice exchange, pipe, zero-copy lists)
#   ab        the in-process parity sweep under tools/heapwatch.c, odd workers destroying their streams as rounds 1-5 did (FFH_STREAM_DESTROY=1),
#             even workers with the pooled streams of round 6: damaged chunks per side
#   bench     the default bench line (+ --pipelined)
#   prof      rocprofv3 --kernel-trace --stats + the PMC passes of the compare kernel (tools/collect_profiles.sh)
#   shard     one rank's step of a 1 / 2 / 4 / 8-way run through the (one-rank) communicator
#   lists     the list-delivering discover: copying form against FFH_LIST_ZERO_COPY=1
#   sweep     the randomised parity sweep, both checker modes, on the final code
#   setupq    A/B of FFH_ROW_SETUP_Q (the row set-up priced when the suffix image's parts per candidate are chosen: tools/lds_conflict_model.py)
#   gather    FETCH_SIZE and the TCC request counters on a random 8-byte gather of known footprint (66 or 132 B per hit?)
#   c2 skewed timeline other grid r03   as in tools/r05_evidence.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r06_evidence
mkdir -p $O
cd $R
PARTS=${@:-"tests ab bench prof shard lists c2 skewed timeline other sweep"}
for P in $PARTS; do
case $P in
tests)  timeout 2700 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -16 $O/pytest_gpu.log ;;
ab)     AB_DESTROY=1 timeout $(( ${AB_SECS:-240} + 200 )) bash tools/r06_stress_sanitized.sh ${AB_SECS:-240} 12 0 gpurun_out/r06_evidence/stress_ab > $O/stress_ab.log 2>&1; tail -40 $O/stress_ab.log
        for side in 0 1; do echo "destroy=$side: $(cat $O/stress_ab/hw_*_destroy$side.log | grep -E '^all [0-9]+ cases agree' | awk '{c += $2} END {print c + 0}') cases, $(cat $O/stress_ab/hw_*_destroy$side.log | grep -oE 'heapwatch [0-9]+ damaged' | awk '{d += $2} END {print d + 0}') damaged chunks"; done | tee $O/stress_ab_sides.txt ;;
bench)  timeout 900 python bench.py --pipelined > $O/bench_default.log 2> $O/bench_default.err; tail -1 $O/bench_default.log > $O/bench_default.json; tail -c 1500 $O/bench_default.json ;;
prof)   timeout 1800 bash tools/collect_profiles.sh r06_evidence/prof > $O/collect.log 2>&1; tail -3 $O/collect.log ;;
shard)  for n in 1 2 4 8; do timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) --comm 2>/dev/null | grep '^{'; done | tee $O/shard_step.txt ;;   # (one rank: the exchange by slices needs world > 1 -- tests/test_gpu_comm.py runs it over the copy transport)
lists)  for z in 0 1; do echo "FFH_LIST_ZERO_COPY=$z $(FFH_LIST_ZERO_COPY=$z timeout 600 python tools/lists_ab.py 2>/dev/null | tail -1)"; done | tee $O/lists.txt ;;
c2)     timeout 300 python bench.py --targets 4.5e6 --guides 1000 --steps 50 --warmup 5 --cpu-seconds 10 --no-skewed --no-c2 > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.json ;;
skewed) timeout 600 bash tools/skewed_timeline.sh > $O/skewed_timeline.txt 2>&1; tail -30 $O/skewed_timeline.txt ;;
timeline) timeout 300 bash tools/timeline.sh > $O/timeline_step.txt 2>&1; timeout 300 bash tools/timeline.sh --targets 4.5e6 --guides 1000 > $O/timeline_c2.txt 2>&1; tail -4 $O/timeline_step.txt ;;
other)  timeout 900 bash tools/pmc_other_kernels.sh > $O/pmc_other_kernels.txt 2>&1; tail -40 $O/pmc_other_kernels.txt ;;
sweep)  timeout $(( ${STRESS_SECS:-600} + 300 )) bash tools/stress_sweep.sh ${STRESS_SECS:-600} 5 gpurun_out/r06_evidence/stress ;;
setupq) bash tools/build_variant.sh WORK setup0 > /dev/null 2>&1; bash tools/build_variant.sh WORK setup5 FFH_ROW_SETUP_Q=5 > /dev/null 2>&1
        for v in setup0 setup5 setup0 setup5; do echo "$v $(FFH_LIBRARY=$R/flashfry_amd/lib/ab/$v.so timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c 'import sys, json; d = json.loads(sys.stdin.read()); print(d["ms_per_step"], d["breakdown_ms"])')"; done | tee $O/ab_row_setup.txt ;;
gather) timeout 1200 bash tools/pmc_gather_calibration.sh > $O/gather_calibration.log 2>&1; tail -30 $O/gather_calibration.log ;;
grid)   timeout 1500 python tools/timing_grid.py > $O/timing_grid.md 2> $O/timing_grid.err; tail -12 $O/timing_grid.md ;;
r03)    timeout 900 python tools/ingest_scale.py > $O/ingest_scale.txt 2>&1; tail -5 $O/ingest_scale.txt; timeout 900 python tools/cli_wall.py --mbases 3100 --big-guides 100000 > $O/cli_wall.txt 2>&1; tail -12 $O/cli_wall.txt ;;
esac
done
ls -la $O
