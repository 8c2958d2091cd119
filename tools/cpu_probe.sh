cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; lscpu | grep -E "Model name|Socket|Thread|Core|NUMA node\(s\)|MHz" | head; 
python - <<'PY'
import time, zlib, os, threading
data = os.urandom(1<<16)
comp = [zlib.compress(bytes((b & 3) for b in os.urandom(65280)), 5) for _ in range(8)]
def work(n):
    for i in range(n):
        for c in comp: zlib.decompress(c)
for nt in (1, 8, 16, 32, 64, 128):
    th=[threading.Thread(target=work,args=(400,)) for _ in range(nt)]
    t0=time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt=time.perf_counter()-t0
    print(nt, "threads: %.2f s, aggregate %.1f GB/s out" % (dt, nt*400*8*65280/dt/1e9))
PY
