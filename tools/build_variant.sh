#!/bin/bash
# tools/build_variant.sh <git-rev|WORK> <name> [extra hipcc flags, e.g. -DFFH_KT=512]: builds that revision's HIP library as flashfry_amd/lib/ab/<name>.so (for FFH_LIBRARY A/B runs)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/flashfry_amd/lib/ab
T=$(mktemp -d)
if [ "$1" = WORK ]; then mkdir -p $T/flashfry_amd; cp -r $R/flashfry_amd/csrc $T/flashfry_amd/csrc; cp -r $R/include $T/include
else git -C $R archive $1 flashfry_amd/csrc include | tar -x -C $T; fi
S=$T/flashfry_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -I/opt/rocm/include ${@:3} -o $R/flashfry_amd/lib/ab/$2.so $S/ffh_api.hip $S/ffh_dbfile.cpp $S/ffh_dbwrite.cpp -lz -lpthread -ldl
rm -rf $T; ls -la $R/flashfry_amd/lib/ab/$2.so
