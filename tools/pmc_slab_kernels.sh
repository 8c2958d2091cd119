# issue / LDS counters of the per-slab passes of a bounded scan (k_slab_totals, k_slab_subhist, k_slab_filter) on the repeat-structured workload
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
RX="k_slab_totals|k_slab_subhist|k_slab_filter"
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES"; do
  N=$(echo $P | cut -d" " -f1)
  rm -rf /tmp/pmcs_$N
  rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "$RX" --output-format csv -d /tmp/pmcs_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-traffic --no-verify --no-c2 > /tmp/pmcs_$N.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("/tmp/pmcs_*/")):
    for p in glob.glob(d + "**/pmc_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print("%-24s %-22s launches=%d sum=%.5g" % (k, c, len(v), sum(v)))
PY
