#!/bin/bash
# round 6: tools/probes/hip_uaf_probe under tools/libheapwatch.so, phase masks x the two HIP runtimes of this image (the system's
# /opt/rocm libamdhip64.so.7.2 -- what the C++ CLI and a JVM would load -- and the one PyTorch bundles, which every Python process here uses)
secs=${1:-12}; out=${2:-gpurun_out/r06_evidence/uaf_bisect.txt}
mkdir -p $(dirname $out); : > $out
gcc -O2 -fPIC -shared -Wall -o tools/libheapwatch.so tools/heapwatch.c -ldl -lpthread || exit 1
TORCH_HIP=$(python -c "import os, torch; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))" 2>/dev/null)
for rt in system torch; do
  for mask in ${MASKS:-5F 7F 4F 5B 1F 52 10 57}; do
    pre=$PWD/tools/libheapwatch.so
    [ $rt = torch ] && pre=$pre:$TORCH_HIP
    echo "== runtime $rt mask $mask" | tee -a $out
    HEAPWATCH_LOG= LD_PRELOAD=$pre timeout 120 tools/probes/hip_uaf_probe $secs $mask 2>&1 | sed 's/chunk 0x[0-9a-f]* //' | sort | uniq -c | sort -rn | head -8 | tee -a $out
  done
done
