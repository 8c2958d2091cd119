#!/bin/bash
# round 3, second GPU call: full -m gpu suite on the new code, A/B of compare-kernel variants, PMC of the compare kernel
mkdir -p gpurun_out/r03b
O=gpurun_out/r03b
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'])" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run new X=1
  run evenper FFH_LIBRARY=$PWD/flashfry_amd/lib/ab/evenper.so
  run pipe0 FFH_LIBRARY=$PWD/flashfry_amd/lib/ab/pipe0.so
  run pipe3 FFH_LIBRARY=$PWD/flashfry_amd/lib/ab/pipe3.so
done
bash tools/pmc_kernel_sets.sh "k_compare<" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_compare.txt 2>&1
tail -40 $O/pmc_compare.txt
