# PMC counters of the candidate-binning kernels (k_item_partition, k_item_bin) during bench.py
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"; do
  N=$(echo $P | cut -d" " -f1)
  rm -rf /tmp/pmc_$N
  rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "${KREGEX:-k_item_partition|k_item_bin|k_guide_epilogue}" --output-format csv -d /tmp/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-traffic > /tmp/pmc_$N.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("/tmp/pmc_SQ_*/")):
    for p in glob.glob(d + "**/pmc_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print("%-42s %-24s n=%d mean=%.4g max=%.4g" % (k, c, len(v), sum(v) / len(v), max(v)))
PY
