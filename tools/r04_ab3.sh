#!/bin/bash
mkdir -p gpurun_out/r04
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
echo "== timeline of one rank's step at 1/8 (working tree)" | tee gpurun_out/r04/timeline_shard8.txt
bash tools/timeline_any.sh python $GRAFT_REPO_ROOT/tools/shard_step.py --shards 8 --rank 4 2>&1 | tee -a gpurun_out/r04/timeline_shard8.txt
echo "== same, through the one-rank communicator" | tee -a gpurun_out/r04/timeline_shard8.txt
bash tools/timeline_any.sh python $GRAFT_REPO_ROOT/tools/shard_step.py --shards 8 --rank 4 --comm 2>&1 | tee -a gpurun_out/r04/timeline_shard8.txt
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in "libflashfry_hip.so 1024" "ab/w4s.so 1024" "ab/w5a.so 1280" "ab/w5b.so 1280" "ab/w5c.so 1280"; do
    set -- $v
    FFH_COMPARE_GRID=$2 FFH_LIBRARY=$PWD/flashfry_amd/lib/$1 timeout 600 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1 grid $2', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()})" | tee -a gpurun_out/r04/ab3.txt
  done
done
