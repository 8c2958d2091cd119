#!/bin/bash
# round 6: the in-process half of the randomised parity sweep with the process's memory under watch.
#   heapwatch workers  LD_PRELOAD=tools/libheapwatch.so: every freed heap chunk of the process (Python's, the library's, the HIP
#                      runtime's) is poisoned, parked and verified -- a write after free by ANYBODY, host thread or DMA, is reported with
#                      what was written and who had freed the chunk; the checker's database sealed read-only (--seal, ffo_db_seal) so that
#                      a stray CPU store into it faults with a backtrace; half of them also under MALLOC_PERTURB_=165
#   asan workers       the library's HOST side (flashfry_amd/lib/asan/libflashfry_hip.so: hipcc -fsanitize=address -fno-gpu-sanitize, device
#                      code unchanged) and the checker (oracle/asan/libff_oracle.so) under AddressSanitizer in one process
# usage: tools/r06_stress_sanitized.sh <seconds> <heapwatch workers> <asan workers> [out dir]      (GPU box)
secs=${1:-600}; nhw=${2:-8}; nas=${3:-4}; out=${4:-gpurun_out/r06_evidence/stress_sanitized}
mkdir -p $out
root=$(pwd)
export FFH_POOL_DEBUG=1
CLANG_RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
gcc -O2 -fPIC -shared -Wall -o tools/libheapwatch.so tools/heapwatch.c -ldl -lpthread || exit 1
if [ $nas -gt 0 ]; then
  mkdir -p flashfry_amd/lib/asan oracle/asan
  if [ ! -f flashfry_amd/lib/asan/libflashfry_hip.so ] || [ -n "$(find flashfry_amd/csrc -newer flashfry_amd/lib/asan/libflashfry_hip.so -type f | head -1)" ]; then
    (cd flashfry_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address -fno-gpu-sanitize -shared-libasan \
       -fno-omit-frame-pointer -I/opt/rocm/include -o ../lib/asan/libflashfry_hip.so ffh_api.hip ffh_dbfile.cpp ffh_dbwrite.cpp -lz -lpthread -ldl) || exit 1
  fi
  /opt/rocm/lib/llvm/bin/clang -O1 -g -std=gnu99 -fPIC -shared -ffp-contract=off -fsanitize=address -shared-libasan -fno-omit-frame-pointer -o oracle/asan/libff_oracle.so \
     oracle/ff_oracle.c oracle/ff_oracle_score.c oracle/ff_oracle_io.c -lz -lm || exit 1
fi
pids=()
for k in $(seq 1 $nhw); do
  seed=$((${SEED_BASE:-7000} + 100 * k + 1))
  if [ $((k % 2)) = 0 ]; then MP=165; else MP=0; fi
  if [ -n "$HW_FENCE" ] && [ $((k % 2)) = 1 ]; then FZ=$HW_FENCE; else FZ=; fi   # HW_FENCE=905-920: the odd workers run heapwatch's fence mode on that size class
  if [ -n "$AB_DESTROY" ] && [ $((k % 2)) = 1 ]; then SD=1; else SD=0; fi   # AB_DESTROY=1: the odd workers destroy their streams as rounds 1-5 did (FFH_STREAM_DESTROY, ffh_streams.hpp)
  FFH_STREAM_DESTROY=$SD HEAPWATCH_BT=${HW_BT:-900-940} HEAPWATCH_FENCE=$FZ MALLOC_PERTURB_=$MP HEAPWATCH_SEGV=1 HEAPWATCH_LOG=$root/$out/heapwatch_${seed} LD_PRELOAD=$root/tools/libheapwatch.so \
    timeout $((secs + 150)) python tools/stress_parity.py $secs $seed --oracle inproc --quiet --seal > $out/hw_${seed}_mp${MP}_destroy$SD.log 2>&1 &
  pids+=($!)
done
for k in $(seq 1 $nas); do
  seed=$((${SEED_BASE:-7000} + 5000 + 100 * k + 1))
  ASAN_OPTIONS=detect_leaks=0:alloc_dealloc_mismatch=0:protect_shadow_gap=0:allocator_may_return_null=1:halt_on_error=1:abort_on_error=0:detect_odr_violation=0:log_path=$root/$out/asan_${seed} \
    LD_PRELOAD=$CLANG_RT FFH_LIBRARY=$root/flashfry_amd/lib/asan/libflashfry_hip.so FFO_LIBRARY=$root/oracle/asan/libff_oracle.so \
    timeout $((secs + 150)) python tools/stress_parity.py $secs $seed --oracle inproc --quiet > $out/asan_${seed}.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
echo "---- summary (exit $rc) ----" | tee $out/summary.txt
for f in $out/*.log; do echo "$(basename $f): $(grep -c '^MISMATCH' $f) mismatches, $(grep -c 'ORACLE DATABASE CHANGED' $f) database changes, $(grep -c '^HEAPWATCH' $f) heapwatch reports, $(grep -c 'AddressSanitizer' $f) asan reports; $(grep -E '^all [0-9]+ cases agree' $f || tail -4 $f | tr '\n' ' ')"; done | tee -a $out/summary.txt
grep -h -A16 "WRITE AFTER FREE" $out/heapwatch_* 2>/dev/null | head -120 | tee -a $out/summary.txt
cat $out/heapwatch_* 2>/dev/null | grep "WRITE AFTER FREE" | sed 's/pid [0-9]*: //; s/chunk 0x[0-9a-f]* //' | sort | uniq -c | sort -rn | head -40 | tee -a $out/summary.txt
grep -h -A45 "ACCESS AFTER FREE" $out/heapwatch_* 2>/dev/null | head -150 | tee -a $out/summary.txt
echo "fence reports: $(cat $out/heapwatch_* 2>/dev/null | grep -c 'ACCESS AFTER FREE')" | tee -a $out/summary.txt
cat $out/heapwatch_* 2>/dev/null | grep "exit:" | awk '{p += $6; c += $12} END {print "heapwatch: " p " chunks parked, " c " released and checked"}' | tee -a $out/summary.txt
ls $out/asan_*.[0-9]* 2>/dev/null | head | tee -a $out/summary.txt
grep -h -A14 "^MISMATCH\|ORACLE DATABASE CHANGED\|ERROR: AddressSanitizer" $out/*.log $out/asan_*.[0-9]* 2>/dev/null | head -80 | tee -a $out/summary.txt
echo "total: $(cat $out/*.log | grep -E '^all [0-9]+ cases agree' | awk '{s += $2} END {print s + 0}') cases" | tee -a $out/summary.txt
exit 0
