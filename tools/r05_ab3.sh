#!/bin/bash
# round 5, A/B 3: the one-sweep LSD passes (FFH_ONESWEEP=1) against a histogram launch + scan per pass (=0) on the repeat-structured workload
mkdir -p gpurun_out/r05
for rep in 1 2; do for v in 0 1; do
  FFH_ONESWEEP=$v timeout 600 python tools/skewed_ab.py 2>/dev/null | tail -2 | cut -c1-260 | sed "s/^/onesweep=$v /" | tee -a gpurun_out/r05/ab3.txt
done; done
