#!/bin/bash
# round 4, GPU call 7: new tests; persistent blocks per CU of the compare launch
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_comm.py -m gpu -x -q -k "replay or work_queues or not_scanned or copy_transport or one_rank" > gpurun_out/r04/pytest_gpu_7.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_gpu_7.log
tail -4 gpurun_out/r04/pytest_gpu_7.log
for rep in 1 2; do
  for g in 1024 512 768 896 2048 4096; do
    FFH_COMPARE_GRID=$g timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('grid $g', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()})" | tee -a gpurun_out/r04/ab7.txt
  done
done
