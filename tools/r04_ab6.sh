#!/bin/bash
# round 4, GPU call 6: LDS fetch depth of the group loop (pieces of the next group requested one group ahead); queue chunk 4 for medium lists on the repeat-structured workload
mkdir -p gpurun_out/r04
for rep in 1 2; do
  for v in libflashfry_hip.so ab/e3.so ab/e4.so ab/e6.so; do
    FFH_LIBRARY=$PWD/flashfry_amd/lib/$v timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()})" | tee -a gpurun_out/r04/ab6.txt
  done
done
for rep in 1 2; do
  for q in -1 0 4; do
    if [ $q = -1 ]; then unset FFH_WORK_QUEUE; else export FFH_WORK_QUEUE=$q; fi
    timeout 900 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-c2 --steps 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d.get('skewed', {})
print('FFH_WORK_QUEUE=$q', round(d['ms_per_step'], 3), 'skewed', s.get('ms_per_step'), s.get('breakdown_ms'))" | tee -a gpurun_out/r04/ab6.txt
  done
done
