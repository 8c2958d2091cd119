# kernel-time summary of one database open (needs /tmp/ff_ingest/db, written by tools/ingest_scale.py)
[ -f /tmp/ff_ingest/db ] || python tools/ingest_scale.py --targets ${1:-3e8} > /tmp/first.log 2>&1
cat > /tmp/open_once.py <<'PY'
from flashfry_amd import capi
with capi.Context(0) as ctx:
    ctx.open("/tmp/ff_ingest/db")
    print(ctx.load_stats().as_dict())
PY
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for lanes in 64; do
rm -rf /tmp/prof_ingest
FFH_INFLATE_LANES=$lanes rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ingest -- python /tmp/open_once.py 2>&1 | grep open_ms
f=$(find /tmp/prof_ingest -name "*kernel_stats.csv" | head -1)
echo "== lanes $lanes"; head -12 "$f" | cut -c1-40,180-300 | grep -i "inflate\|crc"
done
mkdir -p $GRAFT_REPO_ROOT/gpurun_out && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/ingest_kernel_stats.csv
