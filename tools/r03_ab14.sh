#!/bin/bash
# work queue in chunks + entries split by candidates: parity first, then A/B against the fixed stride on one box
mkdir -p gpurun_out/r03n
O=gpurun_out/r03n
S=$PWD/flashfry_amd/lib/ab
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x > $O/pytest1.log 2>&1; echo "pytest rc=$?" >> $O/pytest1.log; tail -3 $O/pytest1.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'], 'tiles', d['plan']['tiles'])" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run chunk8 X=1
  run static FFH_LIBRARY=$S/static_queue.so
  run chunk4 FFH_LIBRARY=$S/chunk4.so
  run chunk16 FFH_LIBRARY=$S/chunk16.so
done
FFH_LIBRARY=$S/stats_q.so timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 3 --warmup 1 2>&1 | grep "wave stats" | tail -1 | tee -a $O/ab.txt
for lib in "" $S/static_queue.so; do
  echo "== skewed ${lib:-chunk8}" | tee -a $O/ab.txt
  FFH_LIBRARY=$lib timeout 600 python tools/skewed_ab.py 2>&1 | grep "bounding" | cut -c1-330 | tee -a $O/ab.txt
done
echo "== skewed, wave stats" | tee -a $O/ab.txt
FFH_LIBRARY=$S/stats_q.so timeout 600 python tools/skewed_ab.py 2>&1 | grep "wave stats" | tail -6 | tee -a $O/ab.txt
for n in 8 4 2; do
  for args in "" "--plan-a 10 --plan-r1 1"; do
    echo "== shard $n $args" | tee -a $O/ab.txt
    timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) $args 2>/dev/null | tail -1 | tee -a $O/ab.txt
  done
done
FFH_LIBRARY=$S/static_queue.so timeout 300 python tools/shard_step.py --shards 8 --rank 4 2>/dev/null | tail -1 | tee -a $O/ab.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
