#!/bin/bash
A="282008855 133796 229 3 300"
timeout 200 python tools/debug_loop.py $A 1 3000 2>&1 | grep -v amdgpu.ids | tail -6
timeout 200 python tools/debug_loop.py $A 1 1500 fresh 2>&1 | grep -v amdgpu.ids | tail -6
timeout 200 python tools/debug_loop.py $A 0 3000 2>&1 | grep -v amdgpu.ids | tail -4
timeout 200 python tools/debug_loop.py 282008855 133796 229 4 300 1 3000 2>&1 | grep -v amdgpu.ids | tail -4
