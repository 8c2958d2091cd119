#!/bin/bash
# is the epilogue's random 8-byte gather 66 or 132 bytes of HBM traffic per hit?  FETCH_SIZE was calibrated (x 2 on
# gfx950) on a STREAMING kernel (k_image_hist: 8 B per target, coalesced); this measures the same counter -- and the TCC's own request
# counters, which count requests, not KiB -- on a gather of KNOWN footprint: tools/ubench/gather_policy (1.16e7 random 8-byte loads from a
# 2.4 GB table: every load its own 128-byte line, 1.16e7 x 128 B = 1.48 GB if a line is fetched whole, 0.74 GB if as one 64-byte half,
# 0.37 GB if as one 32-byte sector) beside k_fill / k_index, whose traffic is known exactly (streams).  One --pmc set per pass.
# usage (GPU box): tools/pmc_gather_calibration.sh [out dir under gpurun_out]
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${1:-r06_evidence/gather_calibration}
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 -o $R/tools/ubench/gather_policy $R/tools/ubench/gather_policy.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum"; do
  N=$(echo $P | cut -d" " -f1)
  timeout 150 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_$N -o g -- $R/tools/ubench/gather_policy > $OUT/pmc_$N.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, collections, os, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for p in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as f:
    f.write("gather: 1.16e7 random 8-byte loads (table 2.4 GB); k_fill writes 8 B x 3e8 = 2.4 GB; k_index writes 4 B x 1.16e7\n")
    for (k, c), v in sorted(agg.items()):
        line = "%-28s %-24s launches=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v))
        if c == "FETCH_SIZE" and "gather" in k:
            line += "   = %.1f B per load as reported (KiB x 1024 / 1.16e7), %.1f with the streaming calibration x 2" % (sum(v) / len(v) * 1024 / 1.16e7, 2 * sum(v) / len(v) * 1024 / 1.16e7)
        if c.startswith("TCC_EA0_RDREQ") and "gather" in k:
            line += "   = %.2f requests per load" % (sum(v) / len(v) / 1.16e7)
        f.write(line + "\n")
print(open(os.path.join(out, "summary.txt")).read())
PY
