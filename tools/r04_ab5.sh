#!/bin/bash
# round 4, GPU call 5: full suite on the tree; which engine copies the hit lists to the host (blit kernel or SDMA) and what it does to the list-delivering discover
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_5.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_gpu_5.log
tail -4 gpurun_out/r04/pytest_gpu_5.log
for rep in 1 2; do
  for e in "X=0" "GPU_BLIT_ENGINE_TYPE=2" "HSA_ENABLE_SDMA=1" "GPU_FORCE_BLIT_COPY_SIZE=0"; do
    env $e timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-skewed --no-c2 --steps 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$e', round(d['ms_per_step'], 3), d.get('discover_product_ms'), 'lists', d.get('discover_with_lists_ms'), d.get('verified'))" | tee -a gpurun_out/r04/ab5.txt
  done
done
