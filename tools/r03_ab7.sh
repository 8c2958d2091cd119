#!/bin/bash
mkdir -p gpurun_out/r03g
O=gpurun_out/r03g
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
  run nospin FFH_NO_SPIN=1
  run lsd FFH_SORT=lsd
done
env X=1 timeout 600 python bench.py --no-traffic --cpu-seconds 0 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('full', round(d['ms_per_step'], 3), 'host guides', d['ms_per_step_host_guides'], 'lists', d['discover_with_lists_ms'], d['discover_with_lists_no_positions_ms'], 'skewed', round(d['skewed']['ms_per_step'], 3), d['skewed']['breakdown_ms'], 'unbounded', round(d['skewed']['unbounded']['ms_per_step'], 3))" | tee -a $O/ab.txt
bash tools/timeline.sh > $O/timeline.txt 2>&1; grep -E "k_seg|k_sort|k_item|k_guide|span|COPY|publish" $O/timeline.txt
