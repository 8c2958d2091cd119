#!/bin/bash
# round 4, GPU call 11: hit-buffer space reserved in chunks of at least 512 / 1024 slots from a wave's first flush on (fewer atomics on the one hit cursor)
mkdir -p gpurun_out/r04
for rep in 1 2 3; do
  for v in libflashfry_hip.so ab/chunk512.so ab/chunk1024.so; do
    FFH_LIBRARY=$PWD/flashfry_amd/lib/$v timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, d['hits']['raw'])" | tee -a gpurun_out/r04/ab11.txt
  done
done
