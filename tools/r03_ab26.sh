#!/bin/bash
# device-resident guide set copied by a kernel instead of the runtime's copy
mkdir -p gpurun_out/r03z
O=gpurun_out/r03z
S=$PWD/flashfry_amd/lib/ab
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), round(d['ms_per_step_host_guides'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'c2', round(d['c2']['ms_per_step'], 3))" | tee -a $O/ab.txt
}
for rep in 1 2 3; do
  run kernel_copy X=1
  run head FFH_LIBRARY=$S/head.so
done
timeout 300 python tools/shard_step.py --shards 8 --rank 4 2>/dev/null | tail -1 | cut -c1-200 | tee -a $O/ab.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed" $O/pytest.log | tail -2; tail -1 $O/pytest.log
