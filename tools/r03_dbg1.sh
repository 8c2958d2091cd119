#!/bin/bash
A="282008855 133796 229 3 300"
for e in "X=1" "FFH_WORK_QUEUE=0" "FFH_GENERIC_COMPARE=1" "FFH_SLAB_PREFIX=per-slab" "FFH_BOUND_TOTALS=exact" "FFH_SORT=lsd"; do
  env $e timeout 120 python tools/debug_case.py $A 1 2>&1 | grep -v "amdgpu.ids" | tail -8
done
echo "--- unbounded"; timeout 120 python tools/debug_case.py $A 0 2>&1 | grep -v "amdgpu.ids" | tail -3
echo "--- max_ot 2000 bounded"; timeout 120 python tools/debug_case.py 282008855 133796 229 3 2000 1 2>&1 | grep -v "amdgpu.ids" | tail -3
