#!/bin/bash
mkdir -p gpurun_out/r03i
O=gpurun_out/r03i
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
env X=1 timeout 600 python bench.py --no-traffic --cpu-seconds 0 --steps 5 --warmup 2 --no-verify --no-c2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['skewed']
print('six slabs, shared prefix', round(d['ms_per_step'], 3), 'skewed', round(s['ms_per_step'], 3), s['breakdown_ms'], 'raw', s['raw_hits'], 'retired', s['retired_guides'], 'unbounded', round(s['unbounded']['ms_per_step'], 3))" | tee -a $O/ab.txt
bash tools/r03_evidence.sh prof other timeline shard grid bulge c2
