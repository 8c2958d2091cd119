# extra PMC sets for the compare kernel: instruction fetch, scalar/vector issue cycles, LDS waits, occupancy
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for P in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES SQ_CYCLES" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU" "SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VSKIPPED SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"; do
  N=$(echo $P | cut -d" " -f1)
  rm -rf /tmp/pmcx_$N
  timeout 150 rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "k_compare<" --output-format csv -d /tmp/pmcx_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-traffic > /tmp/pmcx_$N.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("/tmp/pmcx_*/")):
    for p in glob.glob(d + "**/pmc_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(agg.items()):
            print("%-30s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
