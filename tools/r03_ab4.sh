#!/bin/bash
# round 3, fourth GPU call: the segment sort (two passes by guide + one wave per guide) against the six-pass LSD sort; full suite
mkdir -p gpurun_out/r03d
O=gpurun_out/r03d
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -14 $O/pytest.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'])" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run segsort X=1
  run lsd FFH_SORT=lsd
done
for v in X=1 FFH_SORT=lsd; do
env $v timeout 600 python bench.py --no-traffic --cpu-seconds 0 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v full', round(d['ms_per_step'], 3), 'lists', d['discover_with_lists_ms'], d['discover_with_lists_no_positions_ms'], 'skewed', round(d['skewed']['ms_per_step'], 3), d['skewed']['breakdown_ms'], 'unbounded', round(d['skewed']['unbounded']['ms_per_step'], 3), d['skewed']['unbounded']['breakdown_ms'])" | tee -a $O/ab.txt
done
bash tools/timeline.sh > $O/timeline.txt 2>&1; tail -70 $O/timeline.txt
