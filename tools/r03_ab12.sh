#!/bin/bash
# span-aware shard plan (fixed condition), wave statistics of the compare launch (uniform, skewed), what bounds the epilogue on a shard
mkdir -p gpurun_out/r03l
O=gpurun_out/r03l
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for n in 8 4 2; do
  timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) 2>/dev/null | tail -1
  timeout 300 python tools/shard_step.py --shards $n --rank $((n/2)) --plan-a 10 2>/dev/null | tail -1
done | tee -a $O/ab.txt
FFH_SUMMARY_COPY=1 timeout 300 python tools/shard_step.py --shards 8 --rank 4 2>/dev/null | tail -1 | tee -a $O/ab.txt
timeout 300 bash tools/skewed_timeline.sh $GRAFT_REPO_ROOT/tools/shard_step.py --shards 8 --rank 4 > $O/shard_timeline.txt 2>&1; tail -40 $O/shard_timeline.txt
export FFH_LIBRARY=$PWD/flashfry_amd/lib/ab/stats.so
timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 3 --warmup 1 2>&1 | grep "wave stats" | tail -2 | tee -a $O/ab.txt
timeout 600 python tools/skewed_ab.py 2>&1 | grep "wave stats\|bounding" | tail -16 | tee -a $O/ab.txt
