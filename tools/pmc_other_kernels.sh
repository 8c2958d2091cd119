# HBM traffic (FETCH_SIZE / WRITE_SIZE, KiB) and issue counters of the kernels around the compare launch during bench.py: candidate
# binning, the radix-sort passes, segments, the epilogue.  One --pmc set per run, --kernel-trace only.
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
RX="k_item_bin_direct|k_guide_by_part|k_msd_hist|k_msd_scatter|k_binsort|k_sort_hist|k_sort_scatter|k_segments|k_segsort|k_guide_epilogue|k_work_count|k_work_fill|k_hit_targets"
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS"; do
  N=$(echo $P | cut -d" " -f1)
  rm -rf /tmp/pmco_$N
  rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "$RX" --output-format csv -d /tmp/pmco_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-traffic --no-verify --no-skewed --no-c2 > /tmp/pmco_$N.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("/tmp/pmco_*/")):
    for p in glob.glob(d + "**/pmc_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print("%-46s %-22s launches=%d mean=%.5g" % (k, c, len(v), sum(v) / len(v)))
PY
