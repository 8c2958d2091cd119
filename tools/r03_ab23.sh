#!/bin/bash
# the end of a work list dealt entry by entry (kQueueTail sixteenths)
mkdir -p gpurun_out/r03w
O=gpurun_out/r03w
S=$PWD/flashfry_amd/lib/ab
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest1.log 2>&1; echo "pytest rc=$?" >> $O/pytest1.log; tail -2 $O/pytest1.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'], 'tiles', d['plan']['tiles'])" | tee -a $O/ab.txt
}
for rep in 1 2 3; do
  run tail2 X=1
  run tail0 FFH_LIBRARY=$S/tail0.so
  run tail1 FFH_LIBRARY=$S/tail1.so
  run tail4 FFH_LIBRARY=$S/tail4.so
done
echo "== skewed unbounded + bounded, tail2" | tee -a $O/ab.txt
timeout 600 python tools/skewed_ab.py 2>&1 | grep "bounding" | cut -c1-220 | tee -a $O/ab.txt
