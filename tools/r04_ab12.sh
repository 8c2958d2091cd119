#!/bin/bash
# round 4, GPU call 12: minimum hit-buffer chunk 256 / 384 / 512 on the hg38-scale step and the repeat-structured workload
mkdir -p gpurun_out/r04
for rep in 1 2; do
  for v in libflashfry_hip.so ab/chunk256.so ab/chunk384.so ab/chunk512.so; do
    FFH_LIBRARY=$PWD/flashfry_amd/lib/$v timeout 900 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-c2 --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d.get('skewed', {})
print('$v', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'skewed', round(s.get('ms_per_step', 0), 3), {k: round(v, 3) for k, v in s.get('breakdown_ms', {}).items()}, s.get('raw_hits'))" | tee -a gpurun_out/r04/ab12.txt
  done
done
