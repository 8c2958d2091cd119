#!/bin/bash
# A/B on one box: tools/ab_env.sh VAR "v1 v2 ..." [bench args]  -- runs bench.py once per value (twice round-robin), prints ms per step and the breakdown
var=$1; vals=$2; shift 2
for rep in 1 2; do
  for v in $vals; do
    env $var=$v timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$var=$v', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()})"
  done
done
