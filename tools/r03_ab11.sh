#!/bin/bash
# span-aware plan for bin shards + persistent, prefetching epilogue: parity first, then A/B on one box
mkdir -p gpurun_out/r03k
O=gpurun_out/r03k
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'], 'c2', round(d['c2']['ms_per_step'], 3) if d.get('c2') else None)" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run persistent X=1
  run one_guide_per_wave FFH_EPILOGUE_BLOCKS=25000
  run one_stream FFH_SIDE_STREAMS=0
  run blocks_2048 FFH_EPILOGUE_BLOCKS=2048
  run blocks_768 FFH_EPILOGUE_BLOCKS=768
done
for n in 8 4 2; do
  timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) 2>/dev/null | tail -1
  timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) --plan-a 10 2>/dev/null | tail -1
  FFH_EPILOGUE_BLOCKS=25000 timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) 2>/dev/null | tail -1
  FFH_SIDE_STREAMS=0 timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) 2>/dev/null | tail -1
done | tee -a $O/ab.txt
timeout 600 python tools/skewed_ab.py 2>&1 | tail -3 | tee -a $O/ab.txt
timeout 600 bash tools/skewed_timeline.sh > $O/skewed_timeline.txt 2>&1; tail -45 $O/skewed_timeline.txt
timeout 300 bash tools/timeline.sh > $O/timeline_step.txt 2>&1; tail -50 $O/timeline_step.txt
