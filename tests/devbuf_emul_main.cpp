// tests/devbuf_emul_main.cpp -- the ownership rules of DevBuf<T> (flashfry_amd/csrc/ffh_devbuf.hpp) on the CPU, against counting stand-ins of
// hipMalloc / hipFree (round 6: ffh_ctx_share_db makes a context whose database buffers are ALIASES of another context's).  Every allocation is
// freed exactly once, by its owner; an alias never frees; an alias that has to grow gets an allocation of its own and leaves the owner's alone;
// moves and swaps (select_images swaps image pairs whose buffers may be borrowed) carry the flag.
#include <stdint.h>
#include <stdlib.h>

#include <cstdio>
#include <set>
#include <utility>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorStreamCaptureUnsupported = 900 };
static std::set<void *> g_live;
static int g_double_free = 0, g_mallocs = 0;
static hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); g_live.insert(*p); ++g_mallocs; return hipSuccess; }
static hipError_t hipFree(void *p) { if (!g_live.erase(p)) ++g_double_free; else free(p); return hipSuccess; }

#include "../flashfry_amd/csrc/ffh_devbuf.hpp"

int main() {
    int bad = 0;
    auto expect = [&](bool ok, const char *what) { if (!ok) { printf("FAILED: %s\n", what); ++bad; } };
    {
        DevBuf<uint64_t> owner;
        expect(owner.reserve(1000) == hipSuccess && owner.p && !owner.borrowed && g_live.size() == 1, "owner allocates");
        uint64_t *op = owner.p;
        {
            DevBuf<uint64_t> a;
            a.alias(owner);
            expect(a.p == op && a.cap == owner.cap && a.borrowed, "alias points at the owner's memory");
            expect(a.reserve(500) == hipSuccess && a.p == op && a.borrowed, "an alias that is large enough stays an alias");
            DevBuf<uint64_t> b(std::move(a));
            expect(b.p == op && b.borrowed && !a.p && !a.borrowed, "move construction carries the flag");
            DevBuf<uint64_t> c;
            c.reserve(10);
            uint64_t *cp = c.p;
            std::swap(b, c);   // (what select_images does with image pairs)
            expect(b.p == cp && !b.borrowed && c.p == op && c.borrowed, "swap of an owned and a borrowed buffer");
            expect(c.reserve(owner.cap + 1) == hipSuccess && c.p != op && !c.borrowed && g_live.count(op) == 1, "an alias that has to grow gets memory of its own; the owner's stays");
            DevBuf<uint64_t> d;
            d.alias(owner);
            d.release();
            expect(!d.p && g_live.count(op) == 1, "releasing an alias frees nothing");
            d.alias(owner);
            d.alias(owner);
            DevBuf<uint64_t> e;
            e.alias(DevBuf<uint64_t>());
            expect(!e.p && !e.borrowed, "an alias of nothing is nothing");
        }
        expect(g_live.count(op) == 1 && g_live.size() == 1, "the aliases are gone, the owner's allocation lives, theirs are freed");
        t_capturing = true;
        expect(owner.reserve(owner.cap + 5) == hipErrorStreamCaptureUnsupported && owner.p == op, "no allocation while a sequence is captured");
        t_capturing = false;
    }
    expect(g_live.empty() && g_double_free == 0, "everything freed exactly once");
    printf("%s (%d allocations)\n", bad ? "DEVBUF DIFFERENCES" : "DevBuf: owners free once, aliases never", g_mallocs);
    return bad ? 1 : 0;
}
