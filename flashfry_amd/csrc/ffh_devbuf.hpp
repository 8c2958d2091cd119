// ffh_devbuf.hpp -- DevBuf<T>: a device allocation that grows on demand and frees itself, or an ALIAS of another context's allocation
// (ffh_ctx_share_db, round 6) that frees nothing.  In a header of its own so that tests/devbuf_emul_main.cpp can run its ownership rules on
// the CPU against counting stand-ins of hipMalloc / hipFree.  Part of the ONE translation unit ffh_api.hip.
#pragma once
#include <chrono>
namespace {

// host wall time this thread has spent inside hipMalloc / hipFree (milliseconds): ffh_load_stats.alloc_ms reads it around a load -- a multi-GB
// allocation after another context's buffers have just been freed is where a load's time can go without any kernel running
static thread_local double t_alloc_ms = 0.0;
struct AllocTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~AllocTimer() { t_alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// A captured launch sequence (PrepGraph) holds raw pointers: it is only replayed while every buffer it refers to is where it was
// (prep_signature: address and capacity of each, per context).  While a sequence is being captured on this thread an allocation is
// refused (hipMalloc is not capturable): the caller then runs the sequence uncaptured.
static thread_local bool t_capturing = false;

template <typename T>
struct DevBuf {  // device allocation that grows on demand and frees itself (on the device that is current: the entry points set it)
    T *p = nullptr;
    size_t cap = 0;
    bool borrowed = false;   // an alias of another context's allocation (ffh_ctx_share_db): never freed here, replaced by an allocation of its own when it has to grow
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap), borrowed(o.borrowed) { o.p = nullptr; o.cap = 0; o.borrowed = false; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; cap = o.cap; borrowed = o.borrowed; o.p = nullptr; o.cap = 0; o.borrowed = false; }
        return *this;
    }
    ~DevBuf() { release(); }
    hipError_t reserve(size_t n) {  // contents are NOT preserved
        if (n <= cap) return hipSuccess;
        if (t_capturing) return hipErrorStreamCaptureUnsupported;
        AllocTimer timed;
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr; cap = 0; borrowed = false;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) return e;
        cap = want;
        return hipSuccess;
    }
    void release() { if (p && !borrowed) { AllocTimer timed; (void)hipFree(p); } p = nullptr; cap = 0; borrowed = false; }
    void alias(const DevBuf &o) { release(); p = o.p; cap = o.cap; borrowed = o.p != nullptr; }
};

}  // namespace
