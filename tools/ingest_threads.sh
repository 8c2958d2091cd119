# ingest thread sweep on an hg38-scale synthetic database (writes /tmp/ff_ingest/db first)
python tools/ingest_scale.py --targets 3e8 > /tmp/first.log 2>&1; tail -1 /tmp/first.log | cut -c1-400
for n in 8 16 24 32; do echo "== threads $n"; FFH_LOAD_THREADS=$n FFH_VERBOSE=1 python - <<'PY'
import time
from flashfry_amd import capi
for i in range(2):
    with capi.Context(0) as ctx:
        t0=time.perf_counter(); ctx.open("/tmp/ff_ingest/db"); dt=time.perf_counter()-t0
        print(round(dt,3), {k:(round(v,1) if isinstance(v,float) else v) for k,v in ctx.load_stats().as_dict().items()})
PY
done
