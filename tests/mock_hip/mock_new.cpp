// tests/mock_hip/mock_new.cpp -- LD_PRELOAD beside libmock_hip.so: the n-th `operator new` after mock_new_arm(n) throws std::bad_alloc, once
// (round 6).  tests/mock_hip/fault_main.c arms it from MOCK_NEW_FAIL_AT and walks n: whatever allocation of the library's host code fails, the
// entry point must RETURN an error (FFH_CATCH, csrc/ffh_abi_guard.hpp) -- an exception that crossed the C ABI would end the process, and in the
// JNI embedding the JVM.
#include <atomic>
#include <cstdlib>
#include <new>
#include <exception>
#include <execinfo.h>
#include <unistd.h>

static std::atomic<long long> g_count{0}, g_fail_at{0};
static void *g_throw_bt[24];
static int g_throw_n;
static void on_terminate() {   // where the exception that nobody caught was thrown from (the frames of the library name the unguarded path)
    const char msg[] = "[mock new] std::terminate: the injected std::bad_alloc was thrown from\n";
    if (write(2, msg, sizeof msg - 1) < 0) {}
    backtrace_symbols_fd(g_throw_bt, g_throw_n, 2);
    _exit(134);
}
extern "C" void mock_new_arm(long long n) { g_count = 0; g_fail_at = n; std::set_terminate(on_terminate); }
extern "C" long long mock_new_count(void) { return g_count.load(); }
static void *take(std::size_t n) {
    const long long at = g_fail_at.load();
    if (++g_count == at && at > 0) { g_throw_n = backtrace(g_throw_bt, 24); throw std::bad_alloc(); }
    void *p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new(std::size_t n) { return take(n); }
void *operator new[](std::size_t n) { return take(n); }
void *operator new(std::size_t n, const std::nothrow_t &) noexcept { try { return take(n); } catch (...) { return nullptr; } }
void *operator new[](std::size_t n, const std::nothrow_t &) noexcept { try { return take(n); } catch (...) { return nullptr; } }
void operator delete(void *p) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }
