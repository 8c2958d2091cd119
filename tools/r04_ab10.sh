#!/bin/bash
# round 4, GPU call 10: the pair test in two halves (12 group words live at a time) -- at four waves, and at five with KC 192 / stage 64; the geometry alone
mkdir -p gpurun_out/r04
FFH_COMPARE_GRID=1280 FFH_LIBRARY=$PWD/flashfry_amd/lib/ab/w5s.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mismatch or split or enzyme or work_queues or 19mer" > gpurun_out/r04/pytest_gpu_10.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_gpu_10.log
tail -3 gpurun_out/r04/pytest_gpu_10.log
for rep in 1 2; do
  for v in "libflashfry_hip.so 1024" "ab/kc192.so 1024" "ab/w5s.so 1280"; do
    set -- $v
    FFH_COMPARE_GRID=$2 FFH_LIBRARY=$PWD/flashfry_amd/lib/$1 timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1 grid $2', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()})" | tee -a gpurun_out/r04/ab10b.txt
  done
done
