// tests/pipe_emul_main.cpp -- ffh_pipe_* (flashfry_amd/csrc/ffh_pipe.inc: lanes, FIFO, tickets) run on the CPU from the library's own source,
// against stand-ins of the few things it needs of a context (round 6; built with -fsanitize=thread by tests/test_library_cpu.py).  A stand-in
// "discover" sleeps a little and returns a result that encodes its input, so that the test can tell every ticket got ITS batch's result:
// several producer threads, results collected out of order, errors of single batches, tickets collected twice, a pipe destroyed with work
// still queued.  What this cannot cover: the scans themselves (tests/test_zz_r6_pipe.py, -m gpu).
#include <stdint.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

enum { FFH_OK = 0, FFH_E_ARG = -1, FFH_E_STATE = -2, FFH_E_NOMEM = -3, FFH_E_HIP = -4 };
struct Image { int width = 11; };
struct ffh_ctx { std::string err; Image img[2]; int id = 0; };
struct ffh_result { uint64_t sum = 0; uint32_t n = 0; int lane = 0; };
static std::atomic<int> g_live_results{0}, g_contexts{0}, g_running{0}, g_max_running{0};
static int ffh_ctx_share_db(ffh_ctx *owner, ffh_ctx **out) { *out = new ffh_ctx(); (*out)->id = ++g_contexts; (void)owner; return FFH_OK; }
static void ffh_destroy(ffh_ctx *c) { --g_contexts; delete c; }
static void ffh_result_free(ffh_result *r) { if (r) { --g_live_results; delete r; } }
static int ffh_discover(ffh_ctx *ctx, const uint64_t *g, uint32_t n, int mm, int ot, unsigned flags, ffh_result **out) {
    const int now = ++g_running;
    int seen = g_max_running.load();
    while (now > seen && !g_max_running.compare_exchange_weak(seen, now)) {}
    std::this_thread::sleep_for(std::chrono::microseconds(200 + 37 * (n % 13)));
    --g_running;
    if (mm == 99) { ctx->err = "batch refused"; return FFH_E_STATE; }   // (the test's failing batches)
    ffh_result *r = new ffh_result();
    ++g_live_results;
    for (uint32_t i = 0; i < n; ++i) r->sum += g[i] * 3 + (uint64_t)ot;
    r->n = n; r->lane = ctx->id; (void)flags;
    *out = r;
    return FFH_OK;
}

#include "../flashfry_amd/csrc/ffh_abi_guard.hpp"
#include "../flashfry_amd/csrc/ffh_pipe.inc"

int main() {
    int bad = 0;
    ffh_ctx owner;
    ffh_pipe *pipe = nullptr;
    if (ffh_pipe_create(&owner, 0, &pipe) != FFH_E_ARG || ffh_pipe_create(&owner, 9, &pipe) != FFH_E_ARG) { printf("lane count not checked\n"); ++bad; }
    if (ffh_pipe_create(&owner, 3, &pipe) != FFH_OK || ffh_pipe_lanes(pipe) != 3 || g_contexts != 2) { printf("create failed\n"); return 1; }
    struct Sub { uint64_t ticket; uint64_t want; uint32_t n; bool fails; };
    std::vector<std::vector<Sub>> subs(4);
    std::vector<std::thread> producers;
    for (int t = 0; t < 4; ++t)
        producers.emplace_back([&, t] {
            for (int k = 0; k < 150; ++k) {
                const uint32_t n = 1 + (uint32_t)((t * 31 + k * 7) % 50);
                std::vector<uint64_t> g(n);
                uint64_t want = 0;
                for (uint32_t i = 0; i < n; ++i) { g[i] = (uint64_t)t << 32 | (uint64_t)k << 8 | i; want += g[i] * 3 + (uint64_t)(k + 1); }
                const bool fails = k % 41 == 40;
                uint64_t ticket = 0;
                if (ffh_pipe_submit(pipe, g.data(), n, fails ? 99 : 4, k + 1, 0, &ticket) != FFH_OK || !ticket) { printf("submit failed\n"); ++bad; }
                for (auto &x : g) x = ~0ull;   // the guides were copied: the caller's buffer is its own again
                subs[t].push_back(Sub{ticket, want, n, fails});
            }
        });
    for (auto &p : producers) p.join();
    std::vector<std::thread> consumers;
    std::atomic<int> wrong{0};
    for (int t = 0; t < 4; ++t)
        consumers.emplace_back([&, t] {
            for (size_t k = subs[t].size(); k-- > 0;) {   // out of order: last submitted first
                const Sub &s = subs[t][k];
                ffh_result *r = nullptr;
                const int rc = ffh_pipe_wait(pipe, s.ticket, &r);
                if (s.fails) { if (rc != FFH_E_STATE || r) ++wrong; continue; }
                if (rc != FFH_OK || !r || r->sum != s.want || r->n != s.n) ++wrong;
                ffh_result_free(r);
            }
        });
    for (auto &c : consumers) c.join();
    if (wrong) { printf("%d tickets with a wrong result\n", wrong.load()); ++bad; }
    ffh_result *r = nullptr;
    if (ffh_pipe_wait(pipe, subs[0][0].ticket, &r) != FFH_E_ARG) { printf("a ticket could be collected twice\n"); ++bad; }
    if (g_max_running < 2 || g_max_running > 3) { printf("%d calls ran at once with 3 lanes\n", g_max_running.load()); ++bad; }
    // destroyed with work queued and results uncollected: everything is run, freed, the sharing contexts destroyed
    for (int k = 0; k < 40; ++k) { uint64_t g = k, ticket; ffh_pipe_submit(pipe, &g, 1, 4, 1, 0, &ticket); }
    ffh_pipe_destroy(pipe);
    if (g_live_results != 0 || g_contexts != 0) { printf("%d results, %d contexts left behind\n", g_live_results.load(), g_contexts.load()); ++bad; }
    printf("%s\n", bad ? "PIPE DIFFERENCES" : "600 batches through 3 lanes from 4 producers: every ticket its own result, errors per batch, nothing left behind");
    return bad ? 1 : 0;
}
