#!/bin/bash
# round 3, first GPU call: parity of the new compare kernel + same-box A/B against the round-2 library + gather-policy microbenchmark
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-skewed --no-c2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['breakdown_ms'].items()}, 'raw', d['hits']['raw'])" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run r02 FFH_LIBRARY=$PWD/flashfry_amd/lib/ab/r02.so
  run new X=1
  run new_nodirect FFH_NO_DIRECT=1
  run new_generic FFH_GENERIC_COMPARE=1
done
( cd tools/ubench && ./gather_policy ) > $O/gather_policy.txt 2>&1; cat $O/gather_policy.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp_pmc; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/gp_pmc -o pmc -- $GRAFT_REPO_ROOT/tools/ubench/gather_policy > /dev/null 2>&1
python - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r03a/gather_policy_pmc.txt
import csv, glob, collections
for p in glob.glob("/tmp/gp_pmc/**/pmc_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print("%-28s %-12s launches=%d mean_KiB=%.6g  bytes_per_gather(x2 gfx950 corr)=%.1f" % (k, c, len(v), sum(v) / len(v), sum(v) / len(v) * 1024 * 2 / 11600000))
PY
