#!/bin/bash
# round 5, A/B 19: the number of slabs of a bounded scan now that a guide is stopped INSIDE the slab in which it reaches the limit
# (round 3 chose six slabs when a guide kept all of its last slab's records).  Variants = kSlabRank rewritten in a private copy of
# ffh_scan.inc (s6 = the product's {0,1,4,12,24,40,64}); the repeat-structured workload, bounded, tools/skewed_ab.py.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05; mkdir -p $O; cd $R
for v in ${@:-s6 s5 s4a s4b s3}; do
  FFH_LIBRARY=$R/flashfry_amd/lib/ab/$v.so timeout 400 python tools/skewed_ab.py 2>/dev/null | tail -1 | cut -c1-330 | sed "s/^/$v /" | tee -a $O/ab19.txt
done
