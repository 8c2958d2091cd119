#!/bin/bash
# tools/kres.sh [kernel-name-regex] [extra hipcc flags]: register / spill / scratch / LDS use of the library's kernels as hipcc compiles
# them for gfx950 (-Rpass-analysis=kernel-resource-usage), one line per kernel.  tests/test_library_cpu.py gates the compare kernel on it.
R=$(cd "$(dirname "$0")/.." && pwd)
RX=${1:-k_compare}
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I/opt/rocm/include ${@:2} -Rpass-analysis=kernel-resource-usage -c -o /tmp/kres.o $R/flashfry_amd/csrc/ffh_api.hip 2> /tmp/kres_all.txt
python3 - "$RX" <<'PY'
import re, sys, subprocess
t = open('/tmp/kres_all.txt').read()
rx = re.compile(sys.argv[1])
for m in re.finditer(r'Function Name: (\S+).*?LDS Size \[bytes/block\]: (\d+)', t, re.S):
    name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip().split('(')[0]
    if not rx.search(name):
        continue
    g = lambda k: re.search(k + r': (\d+)', m.group(0)).group(1)
    print('%-52s SGPRs %3s spilled %3s  VGPRs %3s spilled %3s  scratch %3s B/lane  waves/SIMD %s  LDS %s B/block' % (
        name, g('TotalSGPRs'), g('SGPRs Spill'), g(r'\bVGPRs'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), m.group(2)))
PY
