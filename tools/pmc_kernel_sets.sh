# (every pass runs under `timeout 150`: a counter set the profiler cannot schedule has hung a box for the whole gpurun limit)
# tools/pmc_kernel_sets.sh "<kernel regex>" "<set 1>" "<set 2>" ...: one rocprofv3 --pmc pass of bench.py per counter set, per-kernel means
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
K=$1; shift
i=0
for P in "$@"; do
  i=$((i+1)); rm -rf /tmp/pmcs_$i
  timeout 150 rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "$K" --output-format csv -d /tmp/pmcs_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-traffic --no-verify --no-skewed --no-c2 > /tmp/pmcs_$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("/tmp/pmcs_*/")):
    for p in glob.glob(d + "**/pmc_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            agg[(r["Kernel_Name"].split("(")[0][-36:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print("%-38s %-30s n=%d mean=%.4g max=%.4g" % (k, c, len(v), sum(v) / len(v), max(v)))
PY
