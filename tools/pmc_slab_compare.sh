# issue counters of the SIX compare launches of one bounded scan on the repeat-structured workload, launch by launch (tools/skewed_ab.py's
# last call), beside their durations: where a slab's compare launch spends its time
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf /tmp/pmcc
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-include-regex "k_compare<" --output-format csv -d /tmp/pmcc -o pmc -- python $GRAFT_REPO_ROOT/tools/skewed_ab.py > /tmp/pmcc.log 2>&1
python - <<'PY'
import csv, glob, collections
p = glob.glob("/tmp/pmcc/**/pmc_counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(p)):
    rows.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"].split("(")[0][-24:]})[r["Counter_Name"]] = float(r["Counter_Value"])
t = glob.glob("/tmp/pmcc/**/pmc_kernel_trace.csv", recursive=True)
dur = {}
if t:
    for r in csv.DictReader(open(t[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for d, v in list(rows.items())[-7:]:
    print("%-26s %8.1f us  waves %6d  VALU %.3g SALU %.3g  wave_cycles %.3g  wait_any %.3g  wait_inst %.3g  active_valu %.3g  busy %.3g" % (
        v["k"], dur.get(d, 0), v.get("SQ_WAVES", 0), v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0), v.get("SQ_WAVE_CYCLES", 0), v.get("SQ_WAIT_ANY", 0),
        v.get("SQ_WAIT_INST_ANY", 0), v.get("SQ_ACTIVE_INST_VALU", 0), v.get("SQ_BUSY_CYCLES", 0)))
PY
