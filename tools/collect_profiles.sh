#!/bin/bash
# Run on the GPU box (through gpurun): collects the rocprofv3 evidence that goes into profiles/<round>/.
# usage: tools/collect_profiles.sh <outdir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-profiles}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. kernel time summary of the default benchmark command (traffic passes and CPU baseline switched off: they are separate runs)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python "$R/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 --no-traffic --no-verify --no-skewed --no-c2 > "$OUT/stats_bench.log" 2>&1
grep -v '^[WEI]2026' "$OUT/stats_bench.log" | tail -1 > "$OUT/stats_bench.json"
# 2. PMC passes on the compare kernel (one --pmc set per run, --kernel-trace only)
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  N=$(echo $P | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "k_compare<|k_image_hist" --output-format csv -d "$OUT/pmc_$N" -o pmc -- python "$R/bench.py" --steps 2 --warmup 1 --cpu-seconds 0 --no-traffic --no-verify --no-skewed --no-c2 > "$OUT/pmc_$N.log" 2>&1
done
# 3. the default benchmark line itself (with its own traffic passes and CPU baseline)
cd "$R" && python bench.py > "$OUT/bench_default.log" 2>&1
tail -1 "$OUT/bench_default.log" > "$OUT/bench_default.json"
python - "$OUT" <<'PY'
import csv, glob, collections, sys, os
out = sys.argv[1]
with open(os.path.join(out, "pmc_summary.txt"), "w") as f:
    for d in sorted(glob.glob(os.path.join(out, "pmc_*/"))):
        p = os.path.join(d, "pmc_counter_collection.csv")
        if not os.path.exists(p):
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            f.write("%-40s %-26s launches=%d mean=%.6g\n" % (k, c, len(v), sum(v) / len(v)))
print(open(os.path.join(out, "pmc_summary.txt")).read())
PY
