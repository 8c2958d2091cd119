#!/bin/bash
# queue chunk per image chosen by the host (16 / 4 / none), counters in lines of their own
mkdir -p gpurun_out/r03y
O=gpurun_out/r03y
S=$PWD/flashfry_amd/lib/ab
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x > $O/pytest1.log 2>&1; echo "pytest rc=$?" >> $O/pytest1.log; tail -2 $O/pytest1.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'], 'tiles', d['plan']['tiles'], 'c2', round(d['c2']['ms_per_step'], 3))" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run new X=1
  run head FFH_LIBRARY=$S/head.so
done
for e in "X=1" "FFH_LIBRARY=$S/head.so" "FFH_WORK_QUEUE=4" "FFH_WORK_QUEUE=2"; do
  echo "== skewed $e" | tee -a $O/ab.txt
  env $e timeout 600 python tools/skewed_ab.py 2>&1 | grep "bounding 1" | cut -c1-220 | tee -a $O/ab.txt
done
for e in "X=1" "FFH_LIBRARY=$S/head.so" "FFH_WORK_QUEUE=4"; do
  for n in 8 4; do echo "== shard $n $e" | tee -a $O/ab.txt; env $e timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) 2>/dev/null | tail -1 | cut -c1-260 | tee -a $O/ab.txt; done
done
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
