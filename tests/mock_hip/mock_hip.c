/* tests/mock_hip/mock_hip.c -- an LD_PRELOAD stand-in of the 35 HIP entry points libflashfry_hip.so imports, for HOST-LOGIC tests on a box
 * without a GPU (round 6; test infrastructure).  "Device" memory is calloc'd host memory, copies are memcpy, kernels DO NOT RUN (a launch is
 * counted and returns), streams / events / graphs are tokens.  What the library's host code does with contexts, shared databases, pipes,
 * streams and allocations can then be exercised end to end -- create, load an empty database, share, scan (every count reads zero), finalize,
 * destroy -- and counted: mock_hip_counts() tells how many streams were created and DESTROYED (the library must never destroy one,
 * csrc/ffh_streams.hpp), how many device allocations are live, how many frees hit something that was not allocated.
 * It proves nothing about kernels or about the real runtime. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int hipError_t;
typedef struct { unsigned x, y, z; } dim3_t;
enum { N_LIVE = 1 << 16 };
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static void *live[N_LIVE];
static struct { size_t n; void *bt[10]; int nbt; } live_info[N_LIVE];   /* MOCK_HIP_TRACE=1: who allocated what is still live (mock_hip_dump) */
static size_t track_size;
static long long c_stream_create, c_stream_destroy, c_malloc, c_free, c_bad_free, c_launch, c_host_malloc, c_host_free, c_event_create, c_event_destroy;
static __thread hipError_t last_error;

/* MOCK_HIP_FAIL_AT=n: the n-th call (counted over the entry points that can fail on a real box: allocations, copies, memsets, launches, waits,
 * stream / event creation) returns an error, once -- tests/mock_hip/fault_main.c walks n over a whole scenario to see what the library's error paths
 * leave behind */
static long long fail_at = -1, calls;
static int fail_now(void) {
    if (fail_at < 0) { const char *e = getenv("MOCK_HIP_FAIL_AT"); fail_at = e ? atoll(e) : 0; }
    const long long k = __sync_add_and_fetch(&calls, 1);
    return fail_at > 0 && k == fail_at;
}
#define MAYBE_FAIL() do { if (fail_now()) { last_error = 999; return 999; } } while (0)
long long mock_hip_calls(void) { return calls; }

static void track(void *p) {
    static int trace = -1;
    if (trace < 0) trace = getenv("MOCK_HIP_TRACE") != NULL;
    pthread_mutex_lock(&mu);
    for (int i = 0; i < N_LIVE; i++) if (!live[i]) { live[i] = p; live_info[i].n = track_size; live_info[i].nbt = trace ? backtrace(live_info[i].bt, 10) : 0; break; }
    pthread_mutex_unlock(&mu);
}
void mock_hip_dump(void) {   /* what is still allocated, with the library frames that allocated it (addr2line -f -e libflashfry_hip.so <offset>) */
    for (int i = 0; i < N_LIVE; i++) if (live[i]) {
        fprintf(stderr, "[mock hip] live: %zu bytes, allocated by", live_info[i].n);
        for (int k = 2; k < live_info[i].nbt; k++) { Dl_info di; if (dladdr(live_info[i].bt[k], &di) && di.dli_fname && strstr(di.dli_fname, "flashfry")) fprintf(stderr, " +0x%lx", (unsigned long)((char *)live_info[i].bt[k] - (char *)di.dli_fbase)); }
        fprintf(stderr, "\n");
    }
}
static int untrack(void *p) { int ok = 0; pthread_mutex_lock(&mu); for (int i = 0; i < N_LIVE; i++) if (live[i] == p) { live[i] = 0; ok = 1; break; } pthread_mutex_unlock(&mu); return ok; }
static long long n_live(void) { long long n = 0; pthread_mutex_lock(&mu); for (int i = 0; i < N_LIVE; i++) n += live[i] != 0; pthread_mutex_unlock(&mu); return n; }

void mock_hip_counts(long long *out /* [10]: streams created, destroyed, device mallocs, frees, bad frees, live allocations (device + host), launches, host mallocs, events created, events destroyed */) {
    out[0] = c_stream_create; out[1] = c_stream_destroy; out[2] = c_malloc; out[3] = c_free; out[4] = c_bad_free; out[5] = n_live(); out[6] = c_launch; out[7] = c_host_malloc;
    out[8] = c_event_create; out[9] = c_event_destroy;
}

hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
hipError_t hipSetDevice(int d) { return d == 0 ? 0 : 101; }
hipError_t hipDeviceSynchronize(void) { MAYBE_FAIL(); return 0; }
const char *hipGetErrorString(hipError_t e) { (void)e; return "mock HIP error"; }
hipError_t hipGetLastError(void) { hipError_t e = last_error; last_error = 0; return e; }

hipError_t hipMalloc(void **p, size_t n) { MAYBE_FAIL(); *p = calloc(1, n ? n : 1); if (!*p) return 2; track_size = n; track(*p); __sync_fetch_and_add(&c_malloc, 1); return 0; }
hipError_t hipFree(void *p) { if (!p) return 0; if (!untrack(p)) { __sync_fetch_and_add(&c_bad_free, 1); return 1; } free(p); __sync_fetch_and_add(&c_free, 1); return 0; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags) { MAYBE_FAIL(); (void)flags; *p = calloc(1, n ? n : 1); if (!*p) return 2; track_size = n; track(*p); __sync_fetch_and_add(&c_host_malloc, 1); return 0; }
hipError_t hipHostFree(void *p) { if (!p) return 0; if (!untrack(p)) { __sync_fetch_and_add(&c_bad_free, 1); return 1; } free(p); __sync_fetch_and_add(&c_host_free, 1); return 0; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned flags) { (void)flags; *d = h; return 0; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, int kind) { MAYBE_FAIL(); (void)kind; if (n) memmove(dst, src, n); return 0; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, int kind, void *st) { MAYBE_FAIL(); (void)kind; (void)st; if (n) memmove(dst, src, n); return 0; }
hipError_t hipMemsetAsync(void *dst, int v, size_t n, void *st) { MAYBE_FAIL(); (void)st; if (n) memset(dst, v, n); return 0; }

hipError_t hipStreamCreateWithFlags(void **s, unsigned flags) { MAYBE_FAIL(); (void)flags; *s = malloc(8); __sync_fetch_and_add(&c_stream_create, 1); return 0; }
hipError_t hipStreamDestroy(void *s) { free(s); __sync_fetch_and_add(&c_stream_destroy, 1); return 0; }
hipError_t hipStreamSynchronize(void *s) { MAYBE_FAIL(); (void)s; return 0; }
hipError_t hipStreamWaitEvent(void *s, void *e, unsigned flags) { (void)s; (void)e; (void)flags; return 0; }
hipError_t hipEventCreate(void **e) { MAYBE_FAIL(); *e = malloc(8); __sync_fetch_and_add(&c_event_create, 1); return 0; }
hipError_t hipEventCreateWithFlags(void **e, unsigned flags) { (void)flags; return hipEventCreate(e); }
hipError_t hipEventDestroy(void *e) { free(e); __sync_fetch_and_add(&c_event_destroy, 1); return 0; }
hipError_t hipEventRecord(void *e, void *s) { MAYBE_FAIL(); (void)e; (void)s; return 0; }
hipError_t hipEventSynchronize(void *e) { MAYBE_FAIL(); (void)e; return 0; }
hipError_t hipEventElapsedTime(float *ms, void *a, void *b) { (void)a; (void)b; *ms = 0.01f; return 0; }

/* no capture, no graphs: the library runs its launches plainly when a capture cannot be begun */
hipError_t hipStreamBeginCapture(void *s, int mode) { (void)s; (void)mode; return 801; }
hipError_t hipStreamEndCapture(void *s, void **g) { (void)s; *g = 0; return 801; }
hipError_t hipGraphInstantiate(void **exec, void *g, void *a, void *b, size_t c) { (void)g; (void)a; (void)b; (void)c; *exec = 0; return 801; }
hipError_t hipGraphLaunch(void *exec, void *s) { (void)exec; (void)s; return 801; }
hipError_t hipGraphDestroy(void *g) { (void)g; return 0; }
hipError_t hipGraphExecDestroy(void *e) { (void)e; return 0; }

/* kernels are registered and "launched", never run -- except k_publish (csrc/ffh_ctx.hpp), the one-wave kernel behind the library's POLLED host
 * wait: it copies the counter block and a sequence number into page-locked memory the host is spinning on, and is done here on the spot, so
 * that the default wait (no FFH_NO_SPIN) can be exercised too */
static const void *fn_publish;
void **__hipRegisterFatBinary(const void *data) { static void *handle; (void)data; return &handle; }
void __hipRegisterFunction(void **modules, const void *host_fn, char *dev_fn, const char *dev_name, unsigned tl, void *tid, void *bid, void *bdim, void *gdim, int *ws) {
    (void)modules; (void)dev_fn; (void)tl; (void)tid; (void)bid; (void)bdim; (void)gdim; (void)ws;
    if (dev_name && strstr(dev_name, "k_publish")) fn_publish = host_fn;
}
void __hipUnregisterFatBinary(void **modules) { (void)modules; }
static __thread struct { dim3_t grid, block; size_t shmem; void *stream; } cfg;
hipError_t __hipPushCallConfiguration(dim3_t grid, dim3_t block, size_t shmem, void *stream) { cfg.grid = grid; cfg.block = block; cfg.shmem = shmem; cfg.stream = stream; return 0; }
hipError_t __hipPopCallConfiguration(dim3_t *grid, dim3_t *block, size_t *shmem, void **stream) { *grid = cfg.grid; *block = cfg.block; *shmem = cfg.shmem; *stream = cfg.stream; return 0; }
hipError_t hipLaunchKernel(const void *fn, dim3_t grid, dim3_t block, void **args, size_t shmem, void *stream) {
    (void)grid; (void)block; (void)shmem; (void)stream;
    MAYBE_FAIL();
    __sync_fetch_and_add(&c_launch, 1);
    if (fn && fn == fn_publish) {   /* k_publish(const u64 *counters, volatile u64 *host, u64 seq, const u32 *word) */
        const unsigned long long *counters = *(const unsigned long long **)args[0];
        volatile unsigned long long *host = *(volatile unsigned long long **)args[1];
        const unsigned long long seq = *(unsigned long long *)args[2];
        const unsigned *word = *(const unsigned **)args[3];
        if (counters) for (int i = 0; i < 16; i++) host[i] = counters[i];
        if (word) host[17] = *word;
        __sync_synchronize();
        host[16] = seq;
    }
    return 0;
}
