#!/bin/bash
# why prefix-heavy plans are slow on a bin shard (wave statistics), work queue against fixed stride
mkdir -p gpurun_out/r03m
O=gpurun_out/r03m
S=$PWD/flashfry_amd/lib/ab
for args in "" "--plan-a 10 --plan-r1 1" "--plan-a 10 --plan-r1 2" "--plan-a 11 --plan-r1 2" "--plan-a 9 --plan-r1 1"; do
  echo "== shard 8 $args"
  FFH_LIBRARY=$S/stats_dyn.so timeout 300 python tools/shard_step.py --shards 8 --rank 4 $args 2>&1 | grep "wave stats\|shards" | tail -2
done 2>&1 | tee -a $O/ab.txt
echo "== shard 2 auto, static queue" | tee -a $O/ab.txt
FFH_LIBRARY=$S/static_queue.so timeout 300 python tools/shard_step.py --shards 2 --rank 1 2>&1 | tail -1 | tee -a $O/ab.txt
echo "== shard 2 auto, work queue" | tee -a $O/ab.txt
timeout 300 python tools/shard_step.py --shards 2 --rank 1 2>&1 | tail -1 | tee -a $O/ab.txt
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'])" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run queue X=1
  run static FFH_LIBRARY=$S/static_queue.so
done
echo "== skewed, work queue" | tee -a $O/ab.txt
timeout 600 python tools/skewed_ab.py 2>&1 | grep "bounding" | tee -a $O/ab.txt
echo "== skewed, static" | tee -a $O/ab.txt
FFH_LIBRARY=$S/static_queue.so timeout 600 python tools/skewed_ab.py 2>&1 | grep "bounding" | tee -a $O/ab.txt
echo "== skewed, work queue, wave stats" | tee -a $O/ab.txt
FFH_LIBRARY=$S/stats_dyn.so timeout 600 python tools/skewed_ab.py 2>&1 | grep "wave stats" | tail -6 | tee -a $O/ab.txt
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
