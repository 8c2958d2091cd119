"""round 6: flashfry_amd.capi over the mock runtime (tests/mock_hip/libmock_hip.so preloaded): Context.share, Pipe, Comm.set_exchange -- the Python bindings of what was
built while no GPU was available -- called end to end on an empty database (kernels do not run: every guide has zero hits).  Run by tests/test_library_cpu.py."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from flashfry_amd import capi
L = capi.load_library(build=False)
print("devices", L.ffh_device_count())
t = np.zeros(0, dtype=np.uint64); p = np.zeros(0, dtype=np.uint64)
g = (np.arange(40, dtype=np.uint64) * np.uint64(2654435761) << np.uint64(6)) | np.uint64(0x2A) | (np.uint64(1) << np.uint64(48))
with capi.Context(3) as ctx:
    ctx.load_soa(t, p)
    r = ctx.discover(g, 4, 2000, jost=True)
    print("discover", r.n_guides, r.n_hits, len(r.summaries))
    r2 = ctx.discover(g, 4, 40, summaries_only=True)
    other = ctx.share()
    try:
        ctx.load_soa(t, p)
        print("ERROR: owner loaded while shared")
    except capi.FlashFryHipError as e:
        print("owner frozen:", "shared" in str(e))
    r3 = other.discover(g[:10], 3, 60)
    print("shared discover", r3.n_guides, r3.n_hits)
    other.close()
    ctx.load_soa(t, p)
    with ctx.pipe(2) as pipe:
        print("lanes", pipe.lanes)
        ts = [pipe.submit(g[:k + 1], 4, 2000, summaries_only=bool(k % 2), jost=True) for k in range(8)]
        for k, tk in reversed(list(enumerate(ts))):
            rr = pipe.wait(tk)
            assert rr.n_guides == k + 1, (k, rr.n_guides)
        try:
            pipe.wait(ts[0]); print("ERROR: collected twice")
        except capi.FlashFryHipError:
            print("ticket collected once")
    assert ctx.L.ffh_get_bounding(ctx.h) in (0, 1)
ctxs = []
for i in range(3):
    c = capi.Context(3); c.load_soa(t, p); ctxs.append(c)
with capi.Comm.local(ctxs) as comm:
    print("transport", comm.transport, "world", comm.world)
    a = comm.discover(g, 4, 40, jost=True).copy()
    comm.set_exchange("slice")
    b = comm.discover(g, 4, 40, jost=True).copy()
    lists = comm.shard_lists(1)
    comm.set_exchange("gather")
    print("sharded", len(a), a.tobytes() == b.tobytes(), lists.n_guides, comm.timings())
for c in ctxs:
    c.close()
print("python bindings over the mock runtime: ok")
