#!/bin/bash
# heavy work entries at the front of the list
mkdir -p gpurun_out/r03u
O=gpurun_out/r03u
S=$PWD/flashfry_amd/lib/ab
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x > $O/pytest1.log 2>&1; echo "pytest rc=$?" >> $O/pytest1.log; tail -2 $O/pytest1.log
for lib in "" $S/before_lpt.so; do
  echo "== skewed ${lib:-heavy_first}" | tee -a $O/ab.txt
  FFH_LIBRARY=$lib timeout 600 python tools/skewed_ab.py 2>&1 | grep "bounding" | cut -c1-220 | tee -a $O/ab.txt
done
echo "== skewed, wave stats" | tee -a $O/ab.txt
FFH_LIBRARY=$S/stats_q.so timeout 600 python tools/skewed_ab.py 2>&1 | grep "wave stats" | tail -6 | tee -a $O/ab.txt
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'], 'tiles', d['plan']['tiles'], 'c2', round(d['c2']['ms_per_step'], 3))" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run heavy_first X=1
  run before FFH_LIBRARY=$S/before_lpt.so
done
timeout 300 python tools/shard_step.py --shards 8 --rank 4 2>/dev/null | tail -1 | tee -a $O/ab.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
