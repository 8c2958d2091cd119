#!/bin/bash
# round 4, GPU call 13: the adopted reservation rule (first flush exact, then >= 512) against the tree before it, larger minima, and a timing-only build
# without any cursor atomic (results unusable) as the lower bound; parity of the adopted rule
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r04/pytest_gpu_13.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_gpu_13.log
tail -3 gpurun_out/r04/pytest_gpu_13.log
for rep in 1 2; do
  for v in ab/pre_chunk.so libflashfry_hip.so ab/cmin1024.so ab/cmin2048.so ab/noatomic.so; do
    FFH_LIBRARY=$PWD/flashfry_amd/lib/$v timeout 900 python bench.py --no-traffic --cpu-seconds 0 --no-verify --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d.get('skewed', {})
print('$v', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'c2', round(d.get('c2', {}).get('ms_per_step', 0), 4), 'skewed', round(s.get('ms_per_step', 0), 3), {k: round(v, 3) for k, v in s.get('breakdown_ms', {}).items()})" | tee -a gpurun_out/r04/ab13.txt
  done
done
