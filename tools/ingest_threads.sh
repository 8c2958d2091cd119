# ingest at hg38 scale (writes /tmp/ff_ingest/db first): device inflate (default) vs host-thread inflate
python tools/ingest_scale.py --targets 3e8 > /tmp/first.log 2>&1; tail -1 /tmp/first.log | cut -c1-600
for where in device host; do echo "== FFH_INFLATE=$where"; FFH_INFLATE=$where FFH_VERBOSE=1 python - <<'PY'
import time
from flashfry_amd import capi
for i in range(3):
    with capi.Context(0) as ctx:
        t0=time.perf_counter(); ctx.open("/tmp/ff_ingest/db"); dt=time.perf_counter()-t0
        print(round(dt,3), {k:(round(v,1) if isinstance(v,float) else v) for k,v in ctx.load_stats().as_dict().items()})
PY
done
