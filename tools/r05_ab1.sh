#!/bin/bash
# round 5, A/B 1: (a) trip counts of the compare launch at hg38 scale (FFH_TRIP_STATS build: rows, steps, parks, pushes, flushes per
# image) for profiles/r05/compare_attribution.md; (b) buckets per prefix work entry 13 (side_plan's rule) / 14 / 15
mkdir -p gpurun_out/r05
FFH_LIBRARY=$PWD/flashfry_amd/lib/ab/trip.so timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 2 --warmup 1 2> gpurun_out/r05/trip_stats.err > gpurun_out/r05/trip_stats.json
grep "trip stats" gpurun_out/r05/trip_stats.err | tail -1 | tee gpurun_out/r05/trip_stats.txt
for rep in 1 2; do
  for nb in 0 12 14 15; do
    FFH_NB_PREFIX=$nb timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('NB_PREFIX=$nb', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, d['plan'])" | tee -a gpurun_out/r05/ab1.txt
  done
done
