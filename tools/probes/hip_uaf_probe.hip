// tools/probes/hip_uaf_probe.hip -- which HIP call sequence makes the runtime write into a heap chunk it has already freed?  (round 6)
//
// tools/heapwatch.c (LD_PRELOAD) found it under the parity sweep: a 920-byte chunk freed inside libamdhip64 gets, AFTER its free, a
// counter at offset 152 decremented and a 4-byte zero at offset 888 -- once per ~200 library contexts.  Whoever malloc's ~916 bytes next
// (the CPU checker's int[229] work arrays did, twice in 500 000 cases; in the JVM embedding: the JVM) owns the damage.  This probe is pure
// HIP -- no flashfry code -- and replays the library's call pattern phase by phase, asking heapwatch after every phase who was hit.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/hip_uaf_probe tools/probes/hip_uaf_probe.hip -ldl
//   LD_PRELOAD=tools/libheapwatch.so tools/probes/hip_uaf_probe <seconds> <phase mask, hex> [streams per context]
//     mask bits: 1 pageable copies, 2 kernels + events + polled wait, 4 graph capture / instantiate / replay, 8 second stream + event wait + D2H,
//                16 destroy the streams (off: they are kept and reused), 32 hipStreamSynchronize instead of the polled wait, 64 events
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_work(unsigned long long *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 3 + 1; }
__global__ void k_publish(volatile unsigned long long *host, unsigned long long seq) { __threadfence_system(); if (threadIdx.x == 0) host[0] = seq; }

static unsigned long long (*hw_check)(void);
static unsigned long long seen;
static unsigned long long hit[16];
static void check(int phase) {
    if (!hw_check) return;
    unsigned long long e = hw_check();
    if (e != seen) { hit[phase] += e - seen; seen = e; }
}

int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 10.0;
    const unsigned mask = argc > 2 ? (unsigned)strtoul(argv[2], nullptr, 16) : 0x5Fu;
    hw_check = (unsigned long long (*)(void))dlsym(RTLD_DEFAULT, "heapwatch_check_all");
    if (!hw_check) fprintf(stderr, "heapwatch is not preloaded: nothing will be reported\n");
    CK(hipSetDevice(0));
    const int N = 1 << 16;
    std::vector<unsigned long long> host(N, 1);
    hipStream_t keep_st = nullptr, keep_cp = nullptr;
    unsigned long long iters = 0, seq = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        ++iters;
        // ---- phase 0: create (what ffh_create does) ----
        hipStream_t st, cp;
        if ((mask & 16u) || !keep_st) {
            CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            CK(hipStreamCreateWithFlags(&cp, hipStreamNonBlocking));
            keep_st = st; keep_cp = cp;
        } else { st = keep_st; cp = keep_cp; }
        hipEvent_t ev[8] = {}, cev = nullptr;
        if (mask & 64u) { for (auto &e : ev) CK(hipEventCreate(&e)); CK(hipEventCreateWithFlags(&cev, hipEventDisableTiming)); }
        unsigned long long *d = nullptr, *h_pub = nullptr, *d_pub = nullptr, *pinned = nullptr;
        CK(hipMalloc((void **)&d, N * 8));
        CK(hipHostMalloc((void **)&h_pub, 256, hipHostMallocMapped));
        memset(h_pub, 0, 256);
        CK(hipHostGetDevicePointer((void **)&d_pub, h_pub, 0));
        CK(hipHostMalloc((void **)&pinned, N * 8, hipHostMallocDefault));
        check(0);
        auto wait = [&]() {
            if (mask & 32u) { CK(hipStreamSynchronize(st)); return; }
            ++seq;
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, (volatile unsigned long long *)d_pub, seq);
            volatile unsigned long long *h = h_pub;
            while (h[0] != seq) __builtin_ia32_pause();
        };
        // ---- phase 1: pageable copies ----
        if (mask & 1u) {
            CK(hipMemcpyAsync(d, host.data(), N * 8, hipMemcpyDefault, st));
            unsigned long long back = 0;
            CK(hipMemcpyAsync(&back, d + 5, 8, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            check(1);
        }
        // ---- phase 2: kernels + events + the polled wait ----
        if (mask & 2u) {
            for (int r = 0; r < 3; ++r) {
                if (mask & 64u) CK(hipEventRecord(ev[0], st));
                for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(k_work, dim3(N / 256), dim3(256), 0, st, d, N);
                if (mask & 64u) CK(hipEventRecord(ev[1], st));
                wait();
                if (mask & 64u) { float ms; CK(hipEventSynchronize(ev[1])); CK(hipEventElapsedTime(&ms, ev[0], ev[1])); }
            }
            check(2);
        }
        // ---- phase 3: capture, instantiate, replay ----
        hipGraphExec_t exec = nullptr;
        if (mask & 4u) {
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(k_work, dim3(N / 256), dim3(256), 0, st, d, N);
            hipGraph_t g = nullptr;
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
            CK(hipGraphDestroy(g));
            for (int r = 0; r < 3; ++r) { CK(hipGraphLaunch(exec, st)); wait(); }
            check(3);
        }
        // ---- phase 4: the copy stream ----
        if (mask & 8u) {
            if (mask & 64u) { CK(hipEventRecord(cev, st)); CK(hipStreamWaitEvent(cp, cev, 0)); }
            CK(hipMemcpyAsync(pinned, d, N * 8, hipMemcpyDeviceToHost, cp));
            hipLaunchKernelGGL(k_work, dim3(N / 256), dim3(256), 0, st, d, N);
            CK(hipStreamSynchronize(st));
            CK(hipStreamSynchronize(cp));
            check(4);
        }
        // ---- phase 5: destroy (the order of ffh_destroy) ----
        CK(hipStreamSynchronize(st));
        if (exec) CK(hipGraphExecDestroy(exec));
        check(5);
        CK(hipFree(d));
        CK(hipHostFree(h_pub));
        CK(hipHostFree(pinned));
        check(6);
        if (mask & 64u) { for (auto &e : ev) CK(hipEventDestroy(e)); CK(hipEventDestroy(cev)); }
        check(7);
        if (mask & 16u) { CK(hipStreamSynchronize(cp)); CK(hipStreamDestroy(cp)); CK(hipStreamDestroy(st)); }
        check(8);
        // a little heap traffic of the caller's own between two contexts (the checker's arrays)
        for (int k = 0; k < 4; ++k) { int *a = (int *)malloc(916); memset(a, 0x11, 916); free(a); }
        check(9);
    }
    check(10);
    static const char *name[] = {"create", "pageable copies", "kernels+events+wait", "graph", "copy stream", "graph exec destroy", "hipFree/hipHostFree", "event destroy", "stream destroy",
                                 "between contexts", "end"};
    printf("mask 0x%02x: %llu iterations in %.0f s, damaged chunks first seen after phase:", mask, iters, secs);
    unsigned long long tot = 0;
    for (int i = 0; i <= 10; ++i) if (hit[i]) { printf(" [%s: %llu]", name[i], hit[i]); tot += hit[i]; }
    printf(" total %llu\n", tot);
    return 0;
}
