#!/bin/bash
# tools/timeline_any.sh <command ...>: device timeline (kernels + copies) of the LAST discover step the command makes -- everything after
# the second-to-last k_guide_epilogue launch -- with start offset, duration and the idle gap before each
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_tla
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_tla -o tl -- "$@" > /tmp/prof_tla.log 2>&1
tail -2 /tmp/prof_tla.log | cut -c1-400
k=$(find /tmp/prof_tla -name "*kernel_trace.csv" | head -1)
python - "$k" <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
ev.sort()
ep = [i for i, e in enumerate(ev) if "k_guide_epilogue" in e[2]]
first = ep[-2] + 1 if len(ep) > 1 else 0
t0 = ev[first][0]
prev_end, busy = t0, 0
for s, e, n in ev[first:ep[-1] + 2]:
    print("%9.1f us  +%7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
    busy += e - s
    prev_end = max(prev_end, e)
print("span %.1f us, busy %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
PY
