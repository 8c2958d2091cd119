#!/bin/bash
mkdir -p gpurun_out/r03v
O=gpurun_out/r03v
S=$PWD/flashfry_amd/lib/ab
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'], 'tiles', d['plan']['tiles'])" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run heavy_first X=1
  run before FFH_LIBRARY=$S/before_lpt.so
done
for lib in "" $S/before_lpt.so; do
  echo "== skewed ${lib:-heavy_first}" | tee -a $O/ab.txt
  FFH_LIBRARY=$lib timeout 600 python tools/skewed_ab.py 2>&1 | grep "bounding 1" | cut -c1-220 | tee -a $O/ab.txt
done
