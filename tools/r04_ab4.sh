#!/bin/bash
# round 4, GPU call 4: the suite on the rows epilogue + the captured prepare graph; A/B of the graph; one rank's step at 1/2 .. 1/8
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_gpu_4.log
tail -4 gpurun_out/r04/pytest_gpu_4.log
for rep in 1 2; do
  for g in 0 1; do
    FFH_GRAPH=$g timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('FFH_GRAPH=$g', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'c2', d.get('c2', {}).get('ms_per_step'), d.get('c2', {}).get('ms_per_step_with_lists'))" | tee -a gpurun_out/r04/ab4.txt
  done
done
for f in wave auto; do
  for n in 2 4 8; do
    echo "== epilogue $f shards $n" | tee -a gpurun_out/r04/shard_step4.txt
    if [ $f = wave ]; then export FFH_EPILOGUE=wave; else unset FFH_EPILOGUE; fi
    timeout 300 python tools/shard_step.py --shards $n --rank $((n / 2)) --comm 2>&1 | grep '^{' | tee -a gpurun_out/r04/shard_step4.txt
  done
done
