# device timeline of the last benchmark step: every kernel and copy with its start offset, duration and the idle gap before it
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_tl -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-traffic --no-verify --no-skewed --no-c2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['breakdown_ms'])"
k=$(find /tmp/prof_tl -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/prof_tl -name "*memory_copy_trace.csv" | head -1)
python - "$k" "$m" <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
except Exception as e:
    print("no copy trace", e)
ev.sort()
# the last step = from the last k_guide_keys / first kernel after the previous epilogue to the end
cmp = [i for i, e in enumerate(ev) if "k_compare<" in e[2]][-1]                    # the last step's compare launch
prev_ep = [i for i, e in enumerate(ev) if "k_guide_epilogue" in e[2] and i < cmp][-1]  # the step before it ends with an epilogue + copies
seq = [e for e in ev[prev_ep + 1:]]
t0 = ev[prev_ep][1]
prev_end = t0
busy = 0
for s, e, n in seq:
    gap = s - prev_end
    print("%9.1f us  +%7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, n))
    busy += e - s
    prev_end = max(prev_end, e)
print("span %.1f us, busy %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
PY
