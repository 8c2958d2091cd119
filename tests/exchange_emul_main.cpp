// tests/exchange_emul_main.cpp -- the kernels of the shards' exchange (flashfry_amd/csrc/ffh_exchange_kernels.hpp) run ON THE CPU, thread after
// thread, from the SAME source the GPU build compiles: the exchange by guide slices (round 6: pack -> all-to-all -> fold per slice -> priors and
// flag back -> assemble -> all-gather of the folded slices) must leave on every shard the prior, the flag word and the reduced records the
// all-gather form leaves (k_exchange_reduce), for random records, world sizes that do and do not divide the guide count, fewer guides than
// shards, failing shards, first and second round.  What this cannot cover: the transports (RCCL, device copies) above the kernels.
//   g++ -O1 -std=c++17 -o exchange_emul tests/exchange_emul_main.cpp && ./exchange_emul
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define __global__
struct Dim3 { unsigned x = 0; };
static Dim3 blockIdx, threadIdx, blockDim;
static uint32_t atomicOr(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p |= v; return o; }
static uint32_t atomicAdd(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p += v; return o; }
using std::max;
using std::min;
namespace ffh {
struct GuideSummary {  // mirrors ffh_guide_summary (flashfry_amd/csrc/ffh_kernels.hpp)
    uint32_t n_hits, ot_count, overflow, hist[5], closest, closest_count, in_genome, n_scored;
    double cfd_max, cfd_sum, hsu_sum, jost_max, jost_sum;
};
}
#include "../flashfry_amd/csrc/ffh_exchange_kernels.hpp"
using namespace ffh;
static_assert(sizeof(GuideSummary) == 88, "the record the exchange moves");

template <typename F, typename... A>
static void launch(uint64_t threads_needed, unsigned block, F f, A... a) {
    blockDim.x = block;
    const unsigned blocks = (unsigned)std::max<uint64_t>(1, (threads_needed + block - 1) / block);
    for (unsigned b = 0; b < blocks; ++b)
        for (unsigned t = 0; t < block; ++t) { blockIdx.x = b; threadIdx.x = t; f(a...); }
}

static int run_case(uint32_t W, uint32_t G, uint32_t clamp, int fail_shard, uint32_t fail_word, unsigned seed) {
    std::mt19937_64 rng(seed);
    auto rnd = [&](uint32_t n) { return (uint32_t)(rng() % n); };
    // every shard's records [G + 1] (record G = status)
    std::vector<std::vector<GuideSummary>> summ(W, std::vector<GuideSummary>(G + 1));
    for (uint32_t r = 0; r < W; ++r) {
        for (uint32_t g = 0; g < G; ++g) {
            GuideSummary &v = summ[r][g];
            std::memset(&v, 0, sizeof v);
            const uint32_t kind = rnd(4);
            v.ot_count = kind == 0 ? 0u : kind == 1 ? rnd(clamp + 1) : kind == 2 ? rnd(3 * clamp + 2) : rnd(5);
            v.n_hits = v.ot_count ? 1 + rnd(v.ot_count) : 0;
            v.overflow = v.ot_count >= clamp;
            for (int k = 0; k < 5; ++k) v.hist[k] = rnd(7);
            v.closest = rnd(3) ? rnd(5) : 0xFFFFFFFFu; v.closest_count = v.closest == 0xFFFFFFFFu ? 0 : 1 + rnd(9);
            v.in_genome = rnd(3); v.n_scored = rnd(40);
            v.cfd_max = rnd(1000) / 1000.0; v.jost_max = rnd(1000) / 997.0;
            v.cfd_sum = rnd(100000) / 977.0; v.hsu_sum = rnd(100000) / 31.0; v.jost_sum = rnd(100000) / 7919.0;
        }
        std::memset(&summ[r][G], (int)r == fail_shard ? (int)(fail_word & 0xFF) : 0, sizeof(GuideSummary));
    }
    int bad = 0;
    for (int adjusted = 0; adjusted < 2; ++adjusted) {
        // ---- the all-gather form: every rank holds all[world][G + 1] ----
        std::vector<GuideSummary> all((size_t)W * (G + 1));
        for (uint32_t r = 0; r < W; ++r) std::copy(summ[r].begin(), summ[r].end(), all.begin() + (size_t)r * (G + 1));
        std::vector<std::vector<uint32_t>> prior_ref(W, std::vector<uint32_t>(G + 1, 0xABABABABu));
        std::vector<GuideSummary> red_ref(G + 1);
        uint32_t flag_ref = 0;
        for (uint32_t me = 0; me < W; ++me) {
            uint32_t flag[2] = {0, 0};
            launch((uint64_t)G + 1, 256, k_exchange_reduce, (const GuideSummary *)all.data(), G, W, clamp, adjusted, me, prior_ref[me].data(), red_ref.data(), flag);
            if (me == 0) flag_ref = flag[0];
            else if (flag[0] != flag_ref) { printf("  all-gather form: ranks disagree on the flag\n"); ++bad; }
        }
        // ---- the slice form ----
        const uint32_t sl = (G + W - 1) / W;
        if (sl == 0) continue;
        std::vector<std::vector<GuideSummary>> send(W, std::vector<GuideSummary>((size_t)W * (sl + 1))), recv(W, std::vector<GuideSummary>((size_t)W * (sl + 1)));
        for (uint32_t i = 0; i < W; ++i) launch((uint64_t)W * (sl + 1), 256, k_slice_pack, (const GuideSummary *)summ[i].data(), G, sl, W, send[i].data());
        for (uint32_t i = 0; i < W; ++i)          // the all-to-all: shard i's block j -> shard j's block i
            for (uint32_t j = 0; j < W; ++j) std::copy(send[i].begin() + (size_t)j * (sl + 1), send[i].begin() + (size_t)(j + 1) * (sl + 1), recv[j].begin() + (size_t)i * (sl + 1));
        std::vector<std::vector<uint32_t>> prior_all(W, std::vector<uint32_t>((size_t)W * (sl + 1), 0xCDCDCDCDu)), prior_in(W, std::vector<uint32_t>((size_t)W * (sl + 1), 0xEFEFEFEFu));
        std::vector<std::vector<GuideSummary>> red_slice(W, std::vector<GuideSummary>(sl + 1));
        std::vector<std::vector<uint32_t>> flag(W, std::vector<uint32_t>(2, 0));
        for (uint32_t j = 0; j < W; ++j) {
            const uint64_t s0 = (uint64_t)j * sl;
            const uint32_t n_slice = s0 >= G ? 0u : (uint32_t)std::min<uint64_t>(sl, G - s0);
            launch((uint64_t)sl + 1, 256, k_exchange_reduce_slice, (const GuideSummary *)recv[j].data(), n_slice, sl, W, clamp, adjusted, prior_all[j].data(), red_slice[j].data(), flag[j].data());
            if (!adjusted) launch(1024, 1024, k_slice_flag, (const uint32_t *)flag[j].data(), sl, W, prior_all[j].data());
        }
        std::vector<GuideSummary> red((size_t)W * sl);
        for (uint32_t j = 0; j < W; ++j) std::copy(red_slice[j].begin(), red_slice[j].begin() + sl, red.begin() + (size_t)j * sl);   // the all-gather of the folded slices
        if (std::memcmp(red.data(), red_ref.data(), (size_t)G * sizeof(GuideSummary))) { printf("  reduced records differ (adjusted %d)\n", adjusted); ++bad; }
        if (!adjusted) {
            for (uint32_t j = 0; j < W; ++j)      // the all-to-all back: rank j's row r -> shard r's block j
                for (uint32_t r = 0; r < W; ++r) std::copy(prior_all[j].begin() + (size_t)r * (sl + 1), prior_all[j].begin() + (size_t)(r + 1) * (sl + 1), prior_in[r].begin() + (size_t)j * (sl + 1));
            for (uint32_t r = 0; r < W; ++r) {
                std::vector<uint32_t> prior(G + 1, 0x12121212u);
                uint32_t f[2] = {0x77777777u, 0};
                launch((uint64_t)G + 1, 256, k_slice_assemble, (const uint32_t *)prior_in[r].data(), G, sl, W, 1, prior.data(), f);
                if (std::memcmp(prior.data(), prior_ref[r].data(), (size_t)G * 4)) { printf("  prior of shard %u differs\n", r); ++bad; }
                if (f[0] != flag_ref) { printf("  flag of shard %u: %08x, all-gather form %08x\n", r, f[0], flag_ref); ++bad; }
            }
        }
    }
    return bad;
}

int main() {
    int bad = 0, cases = 0;
    const uint32_t worlds[] = {2, 3, 5, 8}, guides[] = {1, 2, 5, 7, 8, 64, 299, 300, 1000, 4097};
    for (uint32_t W : worlds)
        for (uint32_t G : guides)
            for (uint32_t clamp : {1u, 25u, 2000u})
                for (int fail : {-1, 0, (int)W - 1}) {
                    const int b = run_case(W, G, clamp, fail, fail == 0 ? 0xFEu : 0xFFu, 1000u * W + 7u * G + clamp + (unsigned)(fail + 1));
                    if (b) printf("world %u guides %u clamp %u failing shard %d: %d differences\n", W, G, clamp, fail, b);
                    bad += b; ++cases;
                }
    printf("%d cases: %s\n", cases, bad ? "DIFFERENCES" : "the exchange by guide slices leaves what the all-gather form leaves");
    return bad ? 1 : 0;
}
