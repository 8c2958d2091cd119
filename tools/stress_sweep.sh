#!/bin/bash
# tools/stress_sweep.sh <seconds> <workers per mode> [out dir] (SEED_BASE=4000: the workers' seeds are SEED_BASE + 100 k + mode): the randomised parity sweep (tools/stress_parity.py) in both checker
# modes at once -- <workers> processes with the in-process oracle and <workers> with the oracle isolated in a process of its own, every
# one with its own seed, sharing the box's GPU -- plus the replay of round 3's unexplained case (seed 99, from case 5165 on) in both
# modes.  FFH_POOL_DEBUG=1 (canaries + poison on the page-locked result blocks) in all of them.  One summary at the end.
secs=${1:-600}; nw=${2:-5}; out=${3:-gpurun_out/stress}
mkdir -p $out
export FFH_POOL_DEBUG=1
pids=()
for mode in inproc isolated; do
  timeout $((secs + 900)) python tools/stress_parity.py 240 99 5165 --oracle $mode --quiet > $out/replay99_$mode.log 2>&1 &
  pids+=($!)
  for k in $(seq 1 $nw); do
    seed=$((${SEED_BASE:-4000} + 100 * k + ( $( [ $mode = inproc ] && echo 1 || echo 2 ) )))
    timeout $((secs + 900)) python tools/stress_parity.py $secs $seed --oracle $mode --quiet > $out/${mode}_$seed.log 2>&1 &
    pids+=($!)
  done
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
echo "---- summary (exit $rc) ----" | tee $out/summary.txt
for f in $out/*.log; do echo "$(basename $f): $(grep -c '^MISMATCH' $f) mismatches; $(grep -E '^all [0-9]+ cases agree' $f || tail -3 $f | tr '\n' ' ')"; done | tee -a $out/summary.txt
grep -h "pool debug" $out/*.log | sort | uniq -c | tee -a $out/summary.txt
for mode in inproc isolated; do
  echo "$mode total: $(cat $out/${mode}_*.log $out/replay99_$mode.log | grep -E '^all [0-9]+ cases agree' | awk '{s += $2} END {print s + 0}') cases" | tee -a $out/summary.txt
done
exit $rc
