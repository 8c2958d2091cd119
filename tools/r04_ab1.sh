#!/bin/bash
# round 4, first GPU call on the rewritten compare kernel: the -m gpu suite, then r03 library against the working tree on one box
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_gpu_1.log
tail -5 gpurun_out/r04/pytest_gpu_1.log
for rep in 1 2; do
  for lib in flashfry_amd/lib/ab/r03.so flashfry_amd/lib/libflashfry_hip.so; do
    FFH_LIBRARY=$PWD/$lib timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'launch', d['roofline'].get('launch_ms'))" | tee -a gpurun_out/r04/ab1.txt
  done
done
