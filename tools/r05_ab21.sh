#!/bin/bash
# round 5, A/B 21: group tests per work entry of a heavy bucket (FFH_MAX_ENTRY_WORK; the product: 2048).  The tail of a slab's compare
# launch is a wave working off an entry whose rows are full of hits (a flush per ~190 records, each with its look-ups and stores):
# smaller entries spread a repeat family over more waves.  Repeat-structured workload + the hg38-scale step (must not lose).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05; mkdir -p $O; cd $R
for v in ${@:-w2048 w1024 w512 w256}; do
  FFH_LIBRARY=$R/flashfry_amd/lib/ab/$v.so timeout 400 python tools/skewed_ab.py 2>/dev/null | tail -1 | cut -c1-150 | sed "s/^/$v /" | tee -a $O/ab21.txt
  FFH_LIBRARY=$R/flashfry_amd/lib/ab/$v.so timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v step', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['breakdown_ms'].items()})" | tee -a $O/ab21.txt
done
