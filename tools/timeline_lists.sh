# device timeline of one list-delivering ffh_discover (hit lists + positions to the host) at the benchmark's sizes
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_tll
cat > /tmp/tll.py <<'PY'
import sys, time, numpy as np, torch
from flashfry_amd import capi, synth
dev = torch.device("cuda:0")
db = synth.make_database(int(3.0e8), seed=synth.DB_SEED, device=dev)
g = synth.make_guides(100000, device=dev).cpu().numpy().view(np.uint64)
kw = dict(hit_scores=False) if len(sys.argv) < 2 else eval(sys.argv[1])
with capi.Context(3) as ctx:
    ctx.load_soa_device(db["targets"].data_ptr(), db["T"], db["positions"].data_ptr(), db["P"])
    for _ in range(4):
        r = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ctx.discover(g, 4, 2000, **kw)
        print("discover ms", (time.perf_counter() - t0) * 1e3, r.n_hits, r.n_positions)
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_tll -o tl -- python /tmp/tll.py "$@" 2>/dev/null | grep "discover ms"
k=$(find /tmp/prof_tll -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/prof_tll -name "*memory_copy_trace.csv" | head -1)
python - "$k" "$m" <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for r in csv.DictReader(open(sys.argv[2])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s" % r.get("Direction", "?")))
ev.sort()
setups = [i for i, e in enumerate(ev) if "k_guide_keys" in e[2]]
first = setups[-2] if len(setups) >= 2 and ev[setups[-1]][0] - ev[setups[-2]][0] < 4000000 else setups[-1]   # (a pipelined call scans twice)
seq = ev[first - 1:]
t0 = seq[0][0]
for s, e, n in seq:
    if e - s > 20000 or "COPY" in n:
        print("%9.1f us  +%8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
print("span %.1f us" % ((max(e for s, e, n in seq) - t0) / 1e3))
PY
