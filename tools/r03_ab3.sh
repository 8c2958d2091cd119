#!/bin/bash
# round 3, third GPU call: the new scale / error-path tests, A/B of the row-packing variants, full default bench line
mkdir -p gpurun_out/r03c
O=gpurun_out/r03c
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'])" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run new X=1
  run nopick FFH_LIBRARY=$PWD/flashfry_amd/lib/ab/nopick.so
  run nopick_even FFH_LIBRARY=$PWD/flashfry_amd/lib/ab/nopick_even.so
  run pick_even FFH_LIBRARY=$PWD/flashfry_amd/lib/ab/pick_even.so
done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
