#!/bin/bash
# tools/prof_grid_cell.sh <guides> <mismatches>: per-kernel time of ONE cell of tools/timing_grid.py (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_cell
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cell -o cell -- python $GRAFT_REPO_ROOT/tools/timing_grid.py --guides $1 --mismatches $2 --out /tmp/cell.json > /dev/null 2>&1
f=$(find /tmp/prof_cell -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if "ffh::" in r["Name"]:
        print("%-60s calls %5d  avg %8.1f us" % (r["Name"][:60], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
