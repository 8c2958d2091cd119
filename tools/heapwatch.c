/* tools/heapwatch.c -- LD_PRELOAD heap watcher (round 6; test infrastructure, not part of the product).
 *
 * What it is for: twice in ~500 000 randomised parity cases the CPU checker, living in the library's process, gave an answer that a
 * fresh checker, an isolated checker and the library all contradicted (profiles/r05/stress_sweep_b_inproc_4101.log): a guide index of one
 * of its small malloc'd work arrays had turned into another one.  The checker itself is clean under MemorySanitizer, AddressSanitizer,
 * UBSan and MALLOC_PERTURB_ on the very inputs (tools/oracle_replay.c), so SOMEBODY ELSE wrote into a heap chunk the checker owned: a
 * write through a pointer to memory that had been freed and handed out again -- by the library's host code, the HIP runtime's own
 * threads, or a DMA into a user buffer that was pinned on the fly.  In the JVM embedding the victim would be the JVM's heap.
 *
 * What it does: every free() of the process -- Python's, the library's, libamdhip64's, ROCr's -- fills the chunk with 0xFB and parks it in
 * a quarantine instead of freeing it; when the chunk leaves the quarantine (FIFO, HEAPWATCH_MB megabytes, default 512), and whenever
 * heapwatch_check_all() is called, every byte must still be 0xFB.  A byte that is not is a write after free by somebody, reported with
 * the chunk's size, the offset and value of what was written, and the module + offset of whoever freed the chunk.  Every freed chunk of
 * the process becomes a detector, instead of the few hundred bytes of the checker's arrays.
 *
 *   gcc -O2 -fPIC -shared -o tools/libheapwatch.so tools/heapwatch.c -ldl -lpthread
 *   LD_PRELOAD=tools/libheapwatch.so HEAPWATCH_LOG=heapwatch.log python tools/stress_parity.py ...
 *
 * Exported for ctypes.CDLL(None): heapwatch_check_all() -> number of damaged chunks found so far, heapwatch_errors(), heapwatch_stats(). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <errno.h>
#include <fcntl.h>
#include <malloc.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define POISON 0xFB
#define MAX_CHUNK (4u << 20)          /* larger chunks are mmap'd by glibc and unmapped by free: a late write there faults by itself */
#define RING (1u << 22)

static void *(*real_malloc)(size_t);
static void (*real_free)(void *);
static void *(*real_calloc)(size_t, size_t);
static void *(*real_realloc)(void *, size_t);
static void *(*real_memalign)(size_t, size_t);
static int (*real_posix_memalign)(void **, size_t, size_t);
static void *(*real_aligned_alloc)(size_t, size_t);

static char boot[1 << 16];            /* dlsym calls calloc before the real one is known */
static size_t boot_used;
static int initialising, ready;

typedef struct { void *p; uint32_t size; void *caller; } parked;
static parked *ring;
static size_t head, tail;             /* [tail, head) are parked */
static size_t parked_bytes, limit_bytes = (size_t)512 << 20;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static unsigned long long n_errors, n_parked_total, n_checked;
static int log_fd = 2;

/* ---- fence mode: HEAPWATCH_FENCE=lo-hi (request sizes in bytes) ----------------------------------------------------------------------
 * Allocations of that size class get a page of their own inside one reserved arena; free() makes the page PROT_NONE and never hands it
 * out again.  Whoever touches the object after its free FAULTS, and the handler prints the toucher's own native backtrace, who had
 * allocated and who had freed the object, then opens the page again and lets the access proceed (the run goes on, every later touch of
 * another freed object is reported too).  HEAPWATCH_FENCE_MAX objects (default 30000: one mapping per page, vm.max_map_count). */
#define FENCE_FRAMES 14
static __thread int in_bt;   /* this thread is inside backtrace(): its own allocations and frees pass through unwatched */
typedef struct { uint32_t size; uint8_t state /* 1 live, 2 freed, 3 freed and re-opened after a report */; uint8_t na, nf; void *alloc_bt[FENCE_FRAMES], *free_bt[FENCE_FRAMES]; } fence_rec;
static char *fence_base;
static size_t fence_pages, fence_next, fence_lo, fence_hi;
static fence_rec *fence_tab;
static unsigned long long n_fence_faults;
static int in_fence(const void *p) { return fence_base && (const char *)p >= fence_base && (const char *)p < fence_base + fence_pages * 4096; }
static void print_bt(const char *what, void *const *bt, int n) {
    char line[600];
    for (int i = 0; i < n; i++) {
        Dl_info di;
        if (dladdr(bt[i], &di) && di.dli_fname) snprintf(line, sizeof line, "    %s #%d %s+0x%lx%s%s\n", what, i, di.dli_fname, (unsigned long)((char *)bt[i] - (char *)di.dli_fbase), di.dli_sname ? " near " : "", di.dli_sname ? di.dli_sname : "");
        else snprintf(line, sizeof line, "    %s #%d %p\n", what, i, bt[i]);
        if (write(log_fd, line, strlen(line)) < 0) {}
        if (log_fd != 2 && write(2, line, strlen(line)) < 0) {}
    }
}
static size_t *fence_fifo, fifo_head, fifo_tail;   /* freed pages, oldest first: handed out again once the arena has been used up */
static pthread_mutex_t fence_mu = PTHREAD_MUTEX_INITIALIZER;
static void *fence_alloc(size_t n) {
    if (in_bt) return NULL;
    size_t k = __atomic_fetch_add(&fence_next, 1, __ATOMIC_RELAXED);
    if (k >= fence_pages) {
        pthread_mutex_lock(&fence_mu);
        if (fifo_head - fifo_tail < fence_pages / 2) { pthread_mutex_unlock(&fence_mu); return NULL; }   /* (keep at least half of the arena closed) */
        k = fence_fifo[fifo_tail++ % fence_pages];
        pthread_mutex_unlock(&fence_mu);
    }
    char *pg = fence_base + k * 4096;
    if (mprotect(pg, 4096, PROT_READ | PROT_WRITE)) return NULL;
    fence_rec *r = &fence_tab[k];
    r->size = (uint32_t)n; r->state = 1;
    in_bt = 1;
    r->na = (uint8_t)backtrace(r->alloc_bt, FENCE_FRAMES);
    in_bt = 0;
    return pg;   /* (page start: 4096-aligned serves every alignment the callers ask for) */
}
static void fence_free(void *p) {
    size_t k = (size_t)((char *)p - fence_base) / 4096;
    fence_rec *r = &fence_tab[k];
    in_bt = 1;
    r->nf = (uint8_t)backtrace(r->free_bt, FENCE_FRAMES);
    in_bt = 0;
    r->state = 2;
    mprotect(fence_base + k * 4096, 4096, PROT_NONE);
    pthread_mutex_lock(&fence_mu);
    fence_fifo[fifo_head++ % fence_pages] = k;
    pthread_mutex_unlock(&fence_mu);
}

/* HEAPWATCH_BT=lo-hi (usable sizes): the free() of a chunk of that size class also records its native backtrace, printed if the chunk is
 * found damaged -- the outermost frames name the caller's own call site (addr2line on the -g build of the library) */
#define BT_SLOTS 65536
typedef struct { void *p; int n; void *bt[FENCE_FRAMES]; } bt_rec;
static bt_rec *bt_tab;
static size_t bt_lo = 1, bt_hi = 0;

/* HEAPWATCH_SEGV=1: a fault -- e.g. a store into the checker's sealed database (ffo_db_seal) -- prints where it happened */
static struct sigaction old_segv;
static void on_segv(int sig, siginfo_t *si, void *uc) {
    char msg[256];
    if (si && in_fence(si->si_addr)) {   /* a touch of a fenced object after its free: report, open the page, go on */
        size_t k = (size_t)((char *)si->si_addr - fence_base) / 4096;
        fence_rec *r = &fence_tab[k];
        if (r->state == 2) {
            __atomic_fetch_add(&n_fence_faults, 1, __ATOMIC_RELAXED);
            snprintf(msg, sizeof msg, "[heapwatch] pid %d: ACCESS AFTER FREE (fence): object of %u bytes, offset %zu touched by thread %ld\n", (int)getpid(), r->size,
                     (size_t)((char *)si->si_addr - (fence_base + k * 4096)), (long)syscall(186 /* gettid */));
            if (write(log_fd, msg, strlen(msg)) < 0) {}
            if (log_fd != 2 && write(2, msg, strlen(msg)) < 0) {}
            void *bt[32];
            int n = backtrace(bt, 32);
            print_bt("access", bt, n);
            print_bt("freed-by", r->free_bt, r->nf);
            print_bt("allocated-by", r->alloc_bt, r->na);
            r->state = 3;
            mprotect(fence_base + k * 4096, 4096, PROT_READ | PROT_WRITE);
            return;
        }
        if (r->state == 3) { mprotect(fence_base + k * 4096, 4096, PROT_READ | PROT_WRITE); return; }
    }
    snprintf(msg, sizeof msg, "[heapwatch] pid %d: signal %d at address %p; native backtrace:\n", (int)getpid(), sig, si ? si->si_addr : NULL);
    if (write(log_fd, msg, strlen(msg)) < 0) {}
    void *bt[48];
    int n = backtrace(bt, 48);
    backtrace_symbols_fd(bt, n, log_fd);
    if (log_fd != 2) { if (write(2, msg, strlen(msg)) < 0) {} backtrace_symbols_fd(bt, n, 2); }
    sigaction(SIGSEGV, &old_segv, NULL);   /* back to whoever was there (Python's faulthandler, the default): the fault repeats and ends the process */
    (void)uc;
}

static void init(void) {
    if (ready || initialising) return;
    initialising = 1;
    real_malloc = (void *(*)(size_t))dlsym(RTLD_NEXT, "malloc");
    real_free = (void (*)(void *))dlsym(RTLD_NEXT, "free");
    real_calloc = (void *(*)(size_t, size_t))dlsym(RTLD_NEXT, "calloc");
    real_realloc = (void *(*)(void *, size_t))dlsym(RTLD_NEXT, "realloc");
    real_memalign = (void *(*)(size_t, size_t))dlsym(RTLD_NEXT, "memalign");
    real_posix_memalign = (int (*)(void **, size_t, size_t))dlsym(RTLD_NEXT, "posix_memalign");
    real_aligned_alloc = (void *(*)(size_t, size_t))dlsym(RTLD_NEXT, "aligned_alloc");
    const char *mb = getenv("HEAPWATCH_MB");
    if (mb && atol(mb) > 0) limit_bytes = (size_t)atol(mb) << 20;
    const char *lg = getenv("HEAPWATCH_LOG");
    if (lg && *lg) {
        char path[512];
        snprintf(path, sizeof path, "%s.%d", lg, (int)getpid());
        int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
        if (fd >= 0) log_fd = fd;
    }
    ring = (parked *)real_malloc(sizeof(parked) * RING);
    const char *bz = getenv("HEAPWATCH_BT");
    if (bz && sscanf(bz, "%zu-%zu", &bt_lo, &bt_hi) == 2 && bt_hi >= bt_lo) {
        bt_tab = (bt_rec *)mmap(NULL, sizeof(bt_rec) * BT_SLOTS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (bt_tab == MAP_FAILED) bt_tab = NULL;
        void *warm[2];
        (void)backtrace(warm, 2);
    }
    const char *fz = getenv("HEAPWATCH_FENCE");
    if (fz && sscanf(fz, "%zu-%zu", &fence_lo, &fence_hi) == 2 && fence_hi >= fence_lo && fence_hi <= 4096) {
        const char *fm = getenv("HEAPWATCH_FENCE_MAX");
        fence_pages = fm && atol(fm) > 0 ? (size_t)atol(fm) : 30000;
        fence_tab = (fence_rec *)mmap(NULL, sizeof(fence_rec) * fence_pages, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        fence_base = (char *)mmap(NULL, fence_pages * 4096, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        fence_fifo = (size_t *)mmap(NULL, sizeof(size_t) * fence_pages, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (fence_base == MAP_FAILED || fence_tab == MAP_FAILED || fence_fifo == MAP_FAILED) { fence_base = NULL; fence_pages = 0; }
    }
    const char *sg = getenv("HEAPWATCH_SEGV");
    if ((sg && *sg == '1') || fence_base) {
        void *warm[2];
        (void)backtrace(warm, 2);   /* (loads libgcc now, not inside the handler) */
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = on_segv; sa.sa_flags = SA_SIGINFO;
        sigaction(SIGSEGV, &sa, &old_segv);
    }
    ready = 1;
    initialising = 0;
}

static void say(const char *s) { if (write(log_fd, s, strlen(s)) < 0) {} if (log_fd != 2 && write(2, s, strlen(s)) < 0) {} }

static int damaged(const parked *k) {   /* 0 = every byte is still the poison */
    const unsigned char *b = (const unsigned char *)k->p;
    size_t n = k->size, i = 0;
    const uint64_t want = 0xFBFBFBFBFBFBFBFBull;
    for (; i + 8 <= n; i += 8) { uint64_t v; memcpy(&v, b + i, 8); if (v != want) break; }
    for (; i < n; i++) if (b[i] != POISON) break;
    if (i >= n) return 0;
    size_t last = n;
    while (last > i && b[last - 1] == POISON) last--;
    char msg[1024], hex[400];
    size_t hn = 0;
    for (size_t j = i; j < last && hn + 40 < sizeof hex; j++) {   /* every changed byte with its offset */
        if (b[j] == POISON) continue;
        hn += (size_t)snprintf(hex + hn, sizeof hex - hn, "@%zu:", j);
        for (; j < last && b[j] != POISON && hn + 8 < sizeof hex; j++) hn += (size_t)snprintf(hex + hn, sizeof hex - hn, "%02x", b[j]);
        hn += (size_t)snprintf(hex + hn, sizeof hex - hn, " ");
    }
    Dl_info di;
    const char *mod = "?";
    unsigned long off = 0;
    if (k->caller && dladdr(k->caller, &di) && di.dli_fname) { mod = di.dli_fname; off = (unsigned long)((char *)k->caller - (char *)di.dli_fbase); }
    snprintf(msg, sizeof msg, "[heapwatch] pid %d: WRITE AFTER FREE: chunk %p of %u bytes, bytes [%zu, %zu) changed: %s%s; freed by %s+0x%lx\n", (int)getpid(), k->p, k->size, i, last,
             hex, "", mod, off);
    say(msg);
    if (bt_tab) { bt_rec *r = &bt_tab[((uintptr_t)k->p >> 4) % BT_SLOTS]; if (r->p == k->p) print_bt("freed-by", r->bt, r->n); }
    return 1;
}

static void release_oldest(void) {     /* mu held */
    parked k = ring[tail % RING];
    tail++;
    parked_bytes -= k.size;
    n_checked++;
    if (damaged(&k)) n_errors++;
    real_free(k.p);
}

void free(void *p) {
    if (!p) return;
    if ((char *)p >= boot && (char *)p < boot + sizeof boot) return;
    if (!ready) { init(); if (!ready) return; }
    if (in_fence(p)) { fence_free(p); return; }
    size_t sz = malloc_usable_size(p);
    if (sz == 0 || sz > MAX_CHUNK) { real_free(p); return; }
    memset(p, POISON, sz);
    void *caller = __builtin_return_address(0);
    if (bt_tab && sz >= bt_lo && sz <= bt_hi && !in_bt) {   /* (outside the lock, and not from inside the unwinder's own frees) */
        in_bt = 1;
        bt_rec tmp;
        tmp.p = p; tmp.n = backtrace(tmp.bt, FENCE_FRAMES);
        bt_tab[((uintptr_t)p >> 4) % BT_SLOTS] = tmp;
        in_bt = 0;
    }
    pthread_mutex_lock(&mu);
    ring[head % RING] = (parked){p, (uint32_t)sz, caller};
    head++;
    parked_bytes += sz;
    n_parked_total++;
    while (parked_bytes > limit_bytes || head - tail >= RING - 1) release_oldest();
    pthread_mutex_unlock(&mu);
}

void *malloc(size_t n) {
    if (!ready) {
        init();
        if (!ready) { size_t a = (boot_used + 15) & ~(size_t)15; if (a + n > sizeof boot) return NULL; boot_used = a + n; return boot + a; }
    }
    if (fence_base && n >= fence_lo && n <= fence_hi) { void *q = fence_alloc(n); if (q) return q; }
    return real_malloc(n);
}
void *calloc(size_t a, size_t b) {
    if (!ready) {
        init();
        if (!ready) { size_t n = a * b, o = (boot_used + 15) & ~(size_t)15; if (o + n > sizeof boot) return NULL; boot_used = o + n; memset(boot + o, 0, n); return boot + o; }
    }
    if (fence_base && a * b >= fence_lo && a * b <= fence_hi) { void *q = fence_alloc(a * b); if (q) return q; }   /* (a fresh page is zero) */
    return real_calloc(a, b);
}
void *realloc(void *p, size_t n) {
    if (!ready) init();
    if (p && (char *)p >= boot && (char *)p < boot + sizeof boot) { void *q = real_malloc(n); if (q) memcpy(q, p, n); return q; }
    if (p && in_fence(p)) {
        size_t old = fence_tab[(size_t)((char *)p - fence_base) / 4096].size;
        void *q = malloc(n);
        if (q) { memcpy(q, p, old < n ? old : n); fence_free(p); }
        return q;
    }
    /* shrinking or moving: glibc frees the old chunk (or its tail) itself, unwatched; growth in place is not a free at all */
    return real_realloc(p, n);
}
#define FENCED(a, n) (fence_base && (n) >= fence_lo && (n) <= fence_hi && (a) <= 4096)
void *memalign(size_t a, size_t n) { if (!ready) init(); if (FENCED(a, n)) { void *q = fence_alloc(n); if (q) return q; } return real_memalign(a, n); }
int posix_memalign(void **out, size_t a, size_t n) { if (!ready) init(); if (FENCED(a, n)) { void *q = fence_alloc(n); if (q) { *out = q; return 0; } } return real_posix_memalign(out, a, n); }
void *aligned_alloc(size_t a, size_t n) { if (!ready) init(); if (FENCED(a, n)) { void *q = fence_alloc(n); if (q) return q; } return real_aligned_alloc(a, n); }

size_t malloc_usable_size(void *p) {
    static size_t (*real_mus)(void *);
    if (p && in_fence(p)) return fence_tab[(size_t)((char *)p - fence_base) / 4096].size;
    if (!real_mus) real_mus = (size_t (*)(void *))dlsym(RTLD_NEXT, "malloc_usable_size");
    return real_mus(p);
}

unsigned long long heapwatch_check_all(void) {   /* verify everything parked right now (without releasing it) */
    if (!ready) return 0;
    pthread_mutex_lock(&mu);
    for (size_t i = tail; i < head; i++) {
        parked *k = &ring[i % RING];
        if (damaged(k)) { n_errors++; memset(k->p, POISON, k->size); }   /* re-armed: the same writer shows again */
    }
    unsigned long long e = n_errors;
    pthread_mutex_unlock(&mu);
    return e;
}
unsigned long long heapwatch_errors(void) { return n_errors; }
void heapwatch_stats(unsigned long long *out /* [4]: parked now, bytes parked now, parked ever, released (checked) */) {
    pthread_mutex_lock(&mu);
    out[0] = head - tail; out[1] = parked_bytes; out[2] = n_parked_total; out[3] = n_checked;
    pthread_mutex_unlock(&mu);
}

__attribute__((destructor)) static void fini(void) {
    if (!ready) return;
    unsigned long long e = heapwatch_check_all();
    char msg[256];
    snprintf(msg, sizeof msg, "[heapwatch] pid %d: exit: %llu chunks parked over the run, %llu released and checked, %zu still parked, %llu damaged; fence: %zu objects, %llu accesses after free\n",
             (int)getpid(), n_parked_total, n_checked, head - tail, e, fence_next < fence_pages ? fence_next : fence_pages, n_fence_faults);
    say(msg);
}
