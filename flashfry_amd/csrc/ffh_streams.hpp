// ffh_streams.hpp -- the library never destroys a HIP stream (round 6).
//
// hipStreamDestroy -> amd::HostQueue::terminate() (ROCclr, commandqueue.cpp) deletes the queue's roc::VirtualGPU as soon as the queue's last
// command is complete ON THE DEVICE, while the runtime's own signal-handler thread can still be inside that command's completion
// callback: the callback then decrements a counter and clears a word inside the VirtualGPU object that has just been freed.  Found with
// tools/heapwatch.c under the randomised parity sweep (profiles/r06/uaf_analysis.md): a 920-byte heap chunk freed at
// libamdhip64.so+0x3aab4d (the `delete virtualDevice_` of terminate()) gets, after its free, offset 152 decremented and four zero bytes at
// offset 888 -- about once per 200 contexts created and destroyed on a busy box.  Whoever malloc's ~916 bytes next owns the damage: twice in
// ~500 000 in-process parity cases that was an int[229] work array of the CPU checker (a guide index turned into its neighbour:
// profiles/r05/stress_sweep_b_inproc_4101.log); in the JVM embedding it would be the JVM's heap.
//
// So: streams come from a process-wide pool per device and go back to it; a stream that exists is never terminated.  A long-lived
// process that creates and destroys contexts reuses the same few streams; nothing is destroyed at process exit either (the runtime
// tears itself down).  Header-only, shared by the two translation units of the library (the inline function's static is one object).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace ffh {

struct StreamPool {
    std::mutex m;
    std::map<int, std::vector<hipStream_t>> idle;   // per device
    size_t created = 0;
    bool destroy = false;   // FFH_STREAM_DESTROY=1 (test hook, tools/r06_stress_sanitized.sh): release() calls hipStreamDestroy as rounds 1-5 did -- the A side of the A/B
    // the device must be current; a non-blocking stream
    hipError_t acquire(int device, hipStream_t *out) {
        {
            std::lock_guard<std::mutex> g(m);
            auto &v = idle[device];
            if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
        }
        const hipError_t e = hipStreamCreateWithFlags(out, hipStreamNonBlocking);
        if (e == hipSuccess) { std::lock_guard<std::mutex> g(m); ++created; }
        return e;
    }
    // the device must be current; everything issued on the stream is waited for, then the stream is idle
    void release(int device, hipStream_t s) {
        if (!s) return;
        (void)hipStreamSynchronize(s);
        if (destroy) { (void)hipStreamDestroy(s); return; }
        std::lock_guard<std::mutex> g(m);
        try { idle[device].push_back(s); } catch (...) {}   // (called from ffh_destroy: must not throw; a stream that cannot be listed stays alive unlisted -- never destroyed)
    }
};
inline StreamPool &stream_pool() {
    static StreamPool *p = [] {   // (never deleted: its streams outlive every static destructor)
        StreamPool *q = new StreamPool();
        const char *e = std::getenv("FFH_STREAM_DESTROY");
        q->destroy = e && e[0] == '1';
        return q;
    }();
    return *p;
}

}  // namespace ffh
