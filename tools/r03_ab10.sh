#!/bin/bash
mkdir -p gpurun_out/r03j
O=gpurun_out/r03j
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'])" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run new X=1
  run summary_copy FFH_SUMMARY_COPY=1
done
for n in 8; do timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) 2>/dev/null | tail -1; done | tee -a $O/ab.txt
bash tools/r03_evidence.sh prof other timeline c2 cli ingest
