#!/bin/bash
# round 4, GPU call 9: candidate binning with four record loads in flight per thread; epilogue with the next chunk's target longs requested ahead
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r04/pytest_gpu_9.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_gpu_9.log
tail -3 gpurun_out/r04/pytest_gpu_9.log
for rep in 1 2 3; do
  for v in ab/pre_unroll.so libflashfry_hip.so ab/both.so; do
    FFH_LIBRARY=$PWD/flashfry_amd/lib/$v timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()})" | tee -a gpurun_out/r04/ab9.txt
  done
done
