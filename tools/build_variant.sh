#!/bin/bash
# tools/build_variant.sh <git-rev|WORK> <name> [KEY=value ...] [extra hipcc flags]: builds that revision's HIP library as
# flashfry_amd/lib/ab/<name>.so (for FFH_LIBRARY A/B runs on one box).  KEY=value rewrites the constexpr of that name in the private
# copy of ffh_compare.hpp's geometry block (FFH_KW=768 FFH_STAGE=192 FFH_WAVES_PER_SIMD=5 ...): the product header has no build switches.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/flashfry_amd/lib/ab
T=$(mktemp -d)
if [ "$1" = WORK ]; then mkdir -p $T/flashfry_amd; cp -r $R/flashfry_amd/csrc $T/flashfry_amd/csrc; cp -r $R/include $T/include
else git -C $R archive $1 flashfry_amd/csrc include | tar -x -C $T; fi
S=$T/flashfry_amd/csrc
flags=()
for a in "${@:3}"; do
  if [[ "$a" =~ ^(FFH_[A-Z_]+)=([0-9]+)$ ]]; then
    if grep -q "^constexpr int ${BASH_REMATCH[1]} = " $S/ffh_compare.hpp; then sed -i "s/^constexpr int ${BASH_REMATCH[1]} = [0-9]*;/constexpr int ${BASH_REMATCH[1]} = ${BASH_REMATCH[2]};/" $S/ffh_compare.hpp
    else flags+=("-D$a"); fi   # (revisions before round 4 took these as macros)
  else flags+=("$a"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -I/opt/rocm/include "${flags[@]}" -o $R/flashfry_amd/lib/ab/$2.so $S/ffh_api.hip $S/ffh_dbfile.cpp $S/ffh_dbwrite.cpp -lz -lpthread -ldl
rm -rf $T; ls -la $R/flashfry_amd/lib/ab/$2.so
