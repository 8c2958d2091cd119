#!/bin/bash
mkdir -p gpurun_out/r03h
O=gpurun_out/r03h
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
for v in X=1 FFH_SLAB_PREFIX=per-slab; do
env $v timeout 600 python bench.py --no-traffic --cpu-seconds 0 --steps 5 --warmup 2 --no-verify --no-c2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['skewed']
print('$v', round(d['ms_per_step'], 3), 'skewed', round(s['ms_per_step'], 3), s['breakdown_ms'], 'raw', s['raw_hits'], 'retired', s['retired_guides'], 'unbounded', round(s['unbounded']['ms_per_step'], 3))" | tee -a $O/ab.txt
done
timeout 300 python tools/stress_parity.py 150 2>&1 | tail -4 | tee -a $O/ab.txt
