#!/bin/bash
# round 4, GPU call 8: a row's set-up look-ups requested ahead (marker word two rows, bucket table one row): parity, then A/B
mkdir -p gpurun_out/r04
FFH_LIBRARY=$PWD/flashfry_amd/lib/ab/rowpf.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mismatch or split or enzyme or work_queues or 19mer" > gpurun_out/r04/pytest_gpu_8.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_gpu_8.log
tail -3 gpurun_out/r04/pytest_gpu_8.log
for rep in 1 2 3; do
  for v in libflashfry_hip.so ab/rowpf.so; do
    FFH_LIBRARY=$PWD/flashfry_amd/lib/$v timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()})" | tee -a gpurun_out/r04/ab8.txt
  done
done
