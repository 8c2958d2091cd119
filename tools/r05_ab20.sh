#!/bin/bash
# round 5, A/B 20: how the compare launches of a bounded scan's slabs deal their work entries.  PMC (tools/pmc_slab_compare.sh): the mean
# wave of a slab's launch lives 38-76 % of the launch -- the rest is waiting for the waves that drew a repeat family's entries.
# m4 / m2 / m1 = queue chunks of 4 (the product) / 2 / 1 entries for medium lists; FFH_WORK_QUEUE=4 forces the medium queue on every
# launch (short lists too), unset = the host's rule (fixed stride below two chunks per wave).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05; mkdir -p $O; cd $R
for v in m4 m2 m1; do for q in "" 4; do
  FFH_WORK_QUEUE=$q FFH_LIBRARY=$R/flashfry_amd/lib/ab/$v.so timeout 400 python tools/skewed_ab.py 2>/dev/null | tail -1 | cut -c1-150 | sed "s/^/$v queue=${q:-auto} /" | tee -a $O/ab20.txt
done; done
