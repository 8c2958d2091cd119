#!/bin/bash
# round 5: the in-process half of the randomised parity sweep alone, twice as many workers (the one disagreement of the round -- the
# checker's in-process database answering differently when asked twice -- only ever showed with the checker in the library's process),
# the checker's database checksummed around every step (tools/stress_parity.py), half of the workers under glibc's MALLOC_CHECK_=3.
# usage: tools/r05_stress_inproc.sh <seconds> <workers> [out dir]
secs=${1:-540}; nw=${2:-12}; out=${3:-gpurun_out/r05_evidence/stress_inproc}
mkdir -p $out
export FFH_POOL_DEBUG=1
pids=()
for k in $(seq 1 $nw); do
  seed=$((${SEED_BASE:-6000} + 100 * k + 1))
  if [ $((k % 2)) = 0 ]; then MC=3; else MC=0; fi
  MALLOC_CHECK_=$MC timeout $((secs + 600)) python tools/stress_parity.py $secs $seed --oracle inproc --quiet > $out/inproc_${seed}_mc$MC.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
echo "---- summary (exit $rc) ----" | tee $out/summary.txt
for f in $out/*.log; do echo "$(basename $f): $(grep -c '^MISMATCH' $f) mismatches, $(grep -c 'ORACLE DATABASE CHANGED' $f) database changes; $(grep -E '^all [0-9]+ cases agree' $f || tail -4 $f | tr '\n' ' ')"; done | tee -a $out/summary.txt
grep -h -A12 "^MISMATCH\|ORACLE DATABASE CHANGED" $out/*.log | head -60 | tee -a $out/summary.txt
echo "total: $(cat $out/*.log | grep -E '^all [0-9]+ cases agree' | awk '{s += $2} END {print s + 0}') cases" | tee -a $out/summary.txt
exit 0
