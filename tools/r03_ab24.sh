#!/bin/bash
# every queue counter in a cache line of its own: chunk sizes again
mkdir -p gpurun_out/r03x
O=gpurun_out/r03x
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
  run lines_c16 X=1
  run oneline_c16 FFH_LIBRARY=$S/oneline.so
  run lines_c8 FFH_LIBRARY=$S/c8.so
  run lines_c4 FFH_LIBRARY=$S/c4.so
  run lines_c2 FFH_LIBRARY=$S/c2.so
  run lines_c4q64 FFH_LIBRARY=$S/c4q64.so
done
for lib in "" $S/c4.so; do
echo "== skewed ${lib:-lines_c16}" | tee -a $O/ab.txt
FFH_LIBRARY=$lib timeout 600 python tools/skewed_ab.py 2>&1 | grep "bounding" | cut -c1-220 | tee -a $O/ab.txt
done
