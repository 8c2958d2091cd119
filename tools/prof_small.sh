# kernel-time summary of the chr22-scale configuration (C2: 1 000 guides x 4.5e6 targets)
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_small
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small -o c2 -- python $GRAFT_REPO_ROOT/bench.py --guides ${1:-1000} --targets ${2:-4.5e6} --steps 20 --warmup 2 --cpu-seconds 0 --no-traffic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['breakdown_ms'])"
f=$(find /tmp/prof_small -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    if "ffh::" in r["Name"] and int(r["Calls"]) >= 20:
        per = float(r["TotalDurationNs"]) / 1e3 / 22
        tot += per
        print("%-56s calls/step %5.1f  %7.1f us/step" % (r["Name"][:56], int(r["Calls"]) / 22, per))
print("kernel time per step: %.1f us" % tot)
PY
