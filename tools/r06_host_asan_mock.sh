#!/bin/bash
# round 6: the library's HOST side under AddressSanitizer and under ThreadSanitizer on a box WITHOUT a GPU -- over tests/mock_hip/libmock_hip.so (device memory = host
# memory, kernels counted and never run): tests/mock_hip/host_logic_main.c = contexts and the stream pool, ffh_ctx_share_db, ffh_pipe_*, the
# sharded discover over the copy transport in both forms of the exchange, ffh_db_write + ffh_db_open through the three loaders (the inflate workers).  (VERDICT r5 item 1a asked for the host side under ASan in the
# in-process sweep; with the GPU pool closed this is the part of it a CPU can do: every host path of round 6's additions, no kernel.)
# usage: tools/r06_host_asan_mock.sh [out file]
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
OUT=${1:-profiles/r06/host_logic_mock.txt}
RTD=$(dirname $(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1))
mkdir -p flashfry_amd/lib/asan
(cd flashfry_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address -fno-gpu-sanitize -shared-libasan \
   -fno-omit-frame-pointer -I/opt/rocm/include -o ../lib/asan/libflashfry_hip.so ffh_api.hip ffh_dbfile.cpp ffh_dbwrite.cpp -lz -lpthread -ldl) || exit 1
gcc -O1 -g -fPIC -shared -Wall -o tests/mock_hip/libmock_hip.so tests/mock_hip/mock_hip.c -lpthread || exit 1
T=$(mktemp -d)
gcc -O1 -g -Wall -o $T/plain tests/mock_hip/host_logic_main.c -Lflashfry_amd/lib -lflashfry_hip -Ltests/mock_hip -lmock_hip -Wl,-rpath,$R/flashfry_amd/lib -Wl,-rpath,$R/tests/mock_hip || exit 1
gcc -O1 -g -Wall -o $T/asan tests/mock_hip/host_logic_main.c -Lflashfry_amd/lib/asan -lflashfry_hip -Ltests/mock_hip -lmock_hip -L$RTD -l:libclang_rt.asan-x86_64.so \
    -Wl,-rpath,$R/flashfry_amd/lib/asan -Wl,-rpath,$R/tests/mock_hip -Wl,-rpath,$RTD || exit 1
# ... and under ThreadSanitizer (the pipe's lanes, the shards' scan threads of ffh_discover_sharded, the loader's and the writer's workers)
TSD=$(dirname $(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1))
mkdir -p flashfry_amd/lib/tsan
(cd flashfry_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=thread -fno-gpu-sanitize -fno-omit-frame-pointer \
   -I/opt/rocm/include -o ../lib/tsan/libflashfry_hip.so ffh_api.hip ffh_dbfile.cpp ffh_dbwrite.cpp -lz -lpthread -ldl) || exit 1
/opt/rocm/lib/llvm/bin/clang -O1 -g -fsanitize=thread -shared-libsan -o $T/tsan tests/mock_hip/host_logic_main.c -Lflashfry_amd/lib/tsan -lflashfry_hip -Ltests/mock_hip -lmock_hip \
    -Wl,-rpath,$R/flashfry_amd/lib/tsan -Wl,-rpath,$R/tests/mock_hip -Wl,-rpath,$TSD || exit 1
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" LD_PRELOAD=$TSD/libclang_rt.tsan-x86_64.so:$R/tests/mock_hip/libmock_hip.so timeout 900 $T/tsan > $T/tsan.log 2>&1
# ... and the drop-in CLI's three subcommands, host layer + library host side under ASan, over the mock runtime (no site is found, no hit: the kernels
# do not run; every line of option handling, FASTA reading, header / BGZF writing and reading, table writing and re-reading does)
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -fsanitize=address -shared-libasan -std=c++17 -ffp-contract=off -o $T/cli flashfry_amd/host/ffhost_core.cpp flashfry_amd/host/ffhost_table.cpp \
    flashfry_amd/host/ffhost_index.cpp flashfry_amd/host/ffhost_cli.cpp -Lflashfry_amd/lib/asan -lflashfry_hip -Wl,-rpath,$R/flashfry_amd/lib/asan -Wl,-rpath,$RTD -lz -lpthread || exit 1
python3 - $T <<'PY'
import sys, numpy as np
rng = np.random.default_rng(3)
seq = "".join(rng.choice(list("ACGT"), size=300000))
open(sys.argv[1] + "/genome.fa", "w").write(">chrA some description\n" + "\n".join(seq[i:i + 60] for i in range(0, len(seq), 60)) + "\n>chrB\n" + seq[:5000] + "\n")
open(sys.argv[1] + "/guides.fa", "w").write(">g1\nGAGTCCGAGCAGAAGAAGAAGGG\n>g2\n" + seq[1000:1023] + "\n")
PY
cli() { ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:detect_odr_violation=0:verify_asan_link_order=0 LD_PRELOAD=$RTD/libclang_rt.asan-x86_64.so:$R/tests/mock_hip/libmock_hip.so timeout 120 $T/cli "$@" > $T/cli.log 2>&1; echo "rc $? $(grep -c AddressSanitizer $T/cli.log) ASan reports; $(tail -1 $T/cli.log)"; }
CLI_INDEX=$(cli index --reference $T/genome.fa --database $T/db --enzyme spcas9ngg --tmpLocation $T)
CLI_DISCOVER=$(cli discover --fasta $T/guides.fa --database $T/db --output $T/out.txt --positionOutput)
CLI_SCORE=$(cli score --input $T/out.txt --output $T/scored.txt --scoringMetrics doench2016cfd,hsu2013,minot,dangerous --database $T/db)
# ... and fault injection: tests/mock_hip/fault_main.c with the n-th HIP call failing, n over the whole scenario, host side under ASan
gcc -O1 -g -Wall -o $T/fault tests/mock_hip/fault_main.c -Lflashfry_amd/lib/asan -lflashfry_hip -Ltests/mock_hip -lmock_hip -L$RTD -l:libclang_rt.asan-x86_64.so \
    -Wl,-rpath,$R/flashfry_amd/lib/asan -Wl,-rpath,$R/tests/mock_hip -Wl,-rpath,$RTD || exit 1
FENV="ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:detect_odr_violation=0:verify_asan_link_order=0 FFH_MOCK_DB=$T/fdb LD_PRELOAD=$RTD/libclang_rt.asan-x86_64.so:$R/tests/mock_hip/libmock_hip.so"
NCALLS=$(env $FENV MOCK_HIP_FAIL_AT=0 $T/fault | awk '{print $2}')
FBAD=0; FERR=0
for n in $(seq 1 $((NCALLS + 8))); do
  o=$(env $FENV MOCK_HIP_FAIL_AT=$n timeout 60 $T/fault 2>&1); rc=$?
  if [ $rc != 0 ] || echo "$o" | grep -q AddressSanitizer; then FBAD=$((FBAD + 1)); echo "fault n=$n rc=$rc: $(echo "$o" | tail -2 | tr '\n' ' ')" >> $T/fault_bad.txt; fi
  echo "$o" | grep -q " errors 0 " || FERR=$((FERR + 1))
done
{
  echo "# tools/r06_host_asan_mock.sh: tests/mock_hip/host_logic_main.c over the mock runtime (no GPU)"
  echo "plain build:                 $(LD_PRELOAD=$R/tests/mock_hip/libmock_hip.so timeout 300 $T/plain 2>&1 | tail -3 | tr '\n' ' ')"
  echo "host side under ASan:        $(ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:detect_odr_violation=0:verify_asan_link_order=0 LD_PRELOAD=$RTD/libclang_rt.asan-x86_64.so:$R/tests/mock_hip/libmock_hip.so timeout 600 $T/asan 2>&1 | tail -3 | tr '\n' ' ')"
  echo "host side under TSan:        $(tail -1 $T/tsan.log); ThreadSanitizer warnings: $(grep -c 'WARNING: ThreadSanitizer' $T/tsan.log)"
  echo "CLI under ASan, index:       $CLI_INDEX"
  echo "CLI under ASan, discover:    $CLI_DISCOVER"
  echo "CLI under ASan, score:       $CLI_SCORE"
  echo "fault injection under ASan:  $NCALLS HIP calls in the scenario, each made to fail in turn: $FERR runs in which a library call then returned an error, $FBAD runs that left an allocation, freed twice, crashed or tripped ASan"
  [ -f $T/fault_bad.txt ] && head -20 $T/fault_bad.txt
  echo "FFH_STREAM_DESTROY=1 (A side): $(FFH_STREAM_DESTROY=1 FFH_NO_SPIN=1 LD_PRELOAD=$R/tests/mock_hip/libmock_hip.so timeout 300 $T/plain 2>&1 | tail -4 | tr '\n' ' ')"
} | tee $OUT
rm -rf $T
