#!/bin/bash
# 16 work queues per side: chunk size A/B; guides per epilogue wave
mkdir -p gpurun_out/r03o
O=gpurun_out/r03o
S=$PWD/flashfry_amd/lib/ab
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest1.log 2>&1; echo "pytest rc=$?" >> $O/pytest1.log; tail -2 $O/pytest1.log
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
  run chunk32 FFH_LIBRARY=$S/chunk32.so
  run epi2 FFH_LIBRARY=$S/epi2.so
  run epi4 FFH_LIBRARY=$S/epi4.so
done
for lib in "" $S/chunk16.so $S/epi2.so $S/epi4.so; do
  echo "== shard 8 ${lib:-chunk8}" | tee -a $O/ab.txt
  FFH_LIBRARY=$lib timeout 300 python tools/shard_step.py --shards 8 --rank 4 2>/dev/null | tail -1 | tee -a $O/ab.txt
done
for lib in "" $S/chunk16.so $S/chunk4.so; do
  echo "== skewed ${lib:-chunk8}" | tee -a $O/ab.txt
  FFH_LIBRARY=$lib timeout 600 python tools/skewed_ab.py 2>&1 | grep "bounding" | cut -c1-200 | tee -a $O/ab.txt
done
