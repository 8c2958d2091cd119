#!/bin/bash
# adaptive chunk length of the work queues, retirement on the compare kernel's own hit counts
mkdir -p gpurun_out/r03q
O=gpurun_out/r03q
S=$PWD/flashfry_amd/lib/ab
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x > $O/pytest1.log 2>&1; echo "pytest rc=$?" >> $O/pytest1.log; tail -4 $O/pytest1.log
echo "== skewed" | tee -a $O/ab.txt
timeout 600 python tools/skewed_ab.py 2>&1 | grep "bounding" | cut -c1-420 | tee -a $O/ab.txt
echo "== skewed, exact totals" | tee -a $O/ab.txt
FFH_BOUND_TOTALS=exact timeout 600 python tools/skewed_ab.py 2>&1 | grep "bounding 1" | cut -c1-420 | tee -a $O/ab.txt
echo "== skewed, work1k" | tee -a $O/ab.txt
FFH_LIBRARY=$S/work1k.so timeout 600 python tools/skewed_ab.py 2>&1 | grep "bounding 1" | cut -c1-420 | tee -a $O/ab.txt
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
  run queue X=1
  run static FFH_LIBRARY=$S/static_queue.so
done
timeout 300 python tools/shard_step.py --shards 8 --rank 4 2>/dev/null | tail -1 | tee -a $O/ab.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
