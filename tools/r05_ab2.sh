#!/bin/bash
# round 5, A/B 2: the fused candidate-list launches (scans inside their consumers, one-launch work list) against the commit before
mkdir -p gpurun_out/r05
for rep in 1 2; do
  for v in ab/head.so libflashfry_hip.so; do
    FFH_LIBRARY=$PWD/flashfry_amd/lib/$v timeout 900 python bench.py --no-traffic --cpu-seconds 0 --no-verify --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d.get('skewed', {})
print('$v', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'c2', round(d.get('c2', {}).get('ms_per_step', 0), 4), 'skewed', round(s.get('ms_per_step', 0), 3), {k: round(v, 3) for k, v in s.get('breakdown_ms', {}).items()})" | tee -a gpurun_out/r05/ab2.txt
  done
done
