#!/bin/bash
# tools/kres.sh [kernel-name-regex] [extra hipcc flags]: register / scratch / LDS use of the library's kernels as hipcc compiles them for gfx950
R=$(cd "$(dirname "$0")/.." && pwd)
RX=${1:-9k_compare}
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off ${@:2} -Rpass-analysis=kernel-resource-usage -o /tmp/kres.so $R/flashfry_amd/csrc/ffh_api.hip 2>&1 \
  | grep -A12 "Function Name: .*$RX" | grep -E "Function Name|SGPRs|VGPRs|Scratch|LDS|Occupancy" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//'
