# kernels of the LAST discover call of a script (default: the repeat-structured workload, tools/skewed_ab.py): per-kernel totals and the big launches in order
# usage: tools/skewed_timeline.sh [script.py args...]
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_sk
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sk -o sk -- python ${@:-$GRAFT_REPO_ROOT/tools/skewed_ab.py} 2>/dev/null | tail -3
k=$(find /tmp/prof_sk -name "*kernel_trace.csv" | head -1)
python - "$k" <<'PY'
import csv, sys, collections
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
ev.sort()
ep = [i for i, e in enumerate(ev) if "k_guide_epilogue" in e[2]]
seq = ev[ep[-2] + 1: ep[-1] + 1]     # the last call
t0 = seq[0][0]
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in seq:
    tot[n][0] += 1; tot[n][1] += e - s
    if e - s > 12000:
        print("%9.1f us  +%8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
print("span %.1f us" % ((seq[-1][1] - t0) / 1e3))
for n, (c, d) in sorted(tot.items(), key=lambda x: -x[1][1])[:25]:
    print("%-72s %4d launches %9.1f us" % (n, c, d / 1e3))
PY
