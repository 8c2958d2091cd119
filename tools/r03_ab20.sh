#!/bin/bash
# where did the queue's gain go: original flush loop against the batched one, queue on / off, wave statistics
mkdir -p gpurun_out/r03t
O=gpurun_out/r03t
S=$PWD/flashfry_amd/lib/ab
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'], 'tiles', d['plan']['tiles'])" | tee -a $O/ab.txt
}
for rep in 1 2 3; do
  run batch_queue X=1
  run batch_static FFH_WORK_QUEUE=0
  run orig_queue FFH_LIBRARY=$S/flush_orig.so
  run orig_static FFH_LIBRARY=$S/flush_orig.so FFH_WORK_QUEUE=0
done
for lib in stats_q stats_orig; do for q in 1 0; do
  echo "== $lib queue $q" | tee -a $O/ab.txt
  FFH_WORK_QUEUE=$q FFH_LIBRARY=$S/$lib.so timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 3 --warmup 1 2>&1 | grep "wave stats" | tail -1 | tee -a $O/ab.txt
done; done
