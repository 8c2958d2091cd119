#!/bin/bash
# round 5, A/B 16: the epilogue's summaries crossing the link as whole cache lines.  Variants (tools/build_variant.sh):
#   base  one 88-byte store per wave into the page-locked result (rounds 3-5)
#   e16 / e8  blocks of 16 / 8 waves that store their own 1 408 / 704 bytes (first call)
#   g16   blocks of 4 waves; the last block to finish of every 4 forwards the group's 16 summaries = 11 lines (second call)
# usage: tools/r05_ab16.sh variant ...   (the first named variant also runs the parity files)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05; mkdir -p $O; cd $R
V=${@:-"base g16"}
T=$(echo $V | awk '{print $NF}')
FFH_LIBRARY=$R/flashfry_amd/lib/ab/$T.so timeout 500 python -m pytest tests/test_gpu_comm.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 | sed "s/^/$T /" | tee -a $O/ab16.txt
for rep in 1 2; do for v in $V; do
  FFH_LIBRARY=$R/flashfry_amd/lib/ab/$v.so timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v step', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['breakdown_ms'].items()}, 'c2', round(d['c2']['ms_per_step'],4))" | tee -a $O/ab16.txt
  FFH_LIBRARY=$R/flashfry_amd/lib/ab/$v.so timeout 300 python tools/shard_step.py --shards 8 --rank 4 2>/dev/null | grep '^{' | cut -c1-400 | sed "s/^/$v shard8 /" | tee -a $O/ab16.txt
done; done
# the repeat-structured workload (the large bins of the final ordering ride in the same variant): unbounded, then bounded
for v in $V; do
  FFH_LIBRARY=$R/flashfry_amd/lib/ab/$v.so timeout 400 python tools/skewed_ab.py 2>/dev/null | tail -2 | cut -c1-420 | sed "s/^/$v /" | tee -a $O/ab16.txt
done
