// ffh_dbwrite.cpp -- ffh_db_write: structure-of-arrays (targets[] with their occurrence counts, positions[]) -> FlashFry's
// on-disk database: BGZF body written bin by bin + text "<db>.header".  Replaces DatabaseWriter.writeToBinnedFile
// (reference/binary/DatabaseWriter.scala:58-111), BlockManager.createLinearBlock / createIndexedBlock
// (blocks/BlockManager.scala:362-442), BinaryHeader.writeHeader (binary/BinaryHeader.scala:69-97) and htsjdk's
// BlockCompressedOutputStream (BGZF, SAM spec 4.1: gzip members with a 'BC' extra subfield, 0xff00 payload bytes each).
//
// Host C++ only.  The three stages are parallel over independent units: bins (payload assembly), BGZF members (deflate);
// only the final concatenation, which fixes every member's file offset and hence every bin's virtual pointer, is serial.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/flashfry_hip.h"

namespace ffh {
void set_global_error(const std::string &m);  // ffh_api.hip: what ffh_last_error(NULL) returns
unsigned usable_cpus();                       // ffh_dbfile.cpp
}

namespace {

struct Pack { int scan, pam; bool five_prime; };
const Pack kPack[7] = {{0, 0, false}, {24, 4, true}, {23, 3, false}, {23, 3, false}, {23, 3, false}, {22, 3, false}, {22, 3, false}};  // StandardScanParameters.scala:84-215

// payload bytes per BGZF member: bgzip / htslib's 0xff00.  htsjdk (the reference's writer) cuts its members at a slightly larger payload
// (its DEFAULT_UNCOMPRESSED_BLOCK_SIZE is 64 KiB minus the gzip overheads; htsjdk is not in the reference tree), so member boundaries
// and with them the virtual pointers in <db>.header differ from a reference-written file of the same content.  Both are valid BGZF and
// the pointers are self-consistent, so each side reads the other's files; byte parity of the container is NOT claimed.
constexpr size_t kMember = 0xff00;
constexpr int kMaxLinear = 500;     // BlockManager maxTargetsPerLinearBin, DatabaseWriter.scala:84-85
constexpr int kSub = 256, kLookup = 4;

unsigned worker_count(size_t units) {
    return (unsigned)std::max<size_t>(1, std::min<size_t>(ffh::usable_cpus(), units));
}

template <typename F>
void parallel_units(size_t n, size_t grain, F fn) {  // fn(begin, end) over dynamic chunks of `grain` units
    std::atomic<size_t> next(0);
    std::atomic<int> threw(0);   // (an exception inside a worker -- std::bad_alloc of a unit's scratch -- ends the process if it leaves the thread: it is carried to the caller)
    auto work = [&]() {
        try {
            for (;;) {
                const size_t a = next.fetch_add(grain);
                if (a >= n || threw) break;
                fn(a, std::min(n, a + grain));
            }
        } catch (...) { threw = 1; }
    };
    const unsigned nt = worker_count((n + grain - 1) / grain);
    std::vector<std::thread> pool;
    try { for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work); } catch (...) { threw = 1; }   // (the threads that did start are joined below)
    work();
    for (auto &t : pool) t.join();
    if (threw) throw std::bad_alloc();
}

std::string bin_name(int width, uint32_t idx) {  // utils/BaseCombinationGenerator.scala:33-69
    std::string s((size_t)width, 'A');
    for (int i = 0; i < width; ++i) s[(size_t)i] = "ACGT"[(idx >> (2 * (width - 1 - i))) & 3];
    return s;
}

int fail(const std::string &m, int code) {
    ffh::set_global_error(m);
    return code;
}

}  // namespace

extern "C" int ffh_db_write(const char *db_path, int enzyme_index, int bin_width, const char *const *contigs, uint32_t n_contigs, const uint64_t *targets,
                            uint64_t n_targets, const uint64_t *positions, uint64_t n_positions) try {
    if (!db_path || (n_targets && !targets) || (n_positions && !positions) || (n_contigs && !contigs)) return fail("null argument", FFH_E_ARG);
    if (enzyme_index < 1 || enzyme_index > 6) return fail("Unable to find the correct parameter pack for enzyme: " + std::to_string(enzyme_index), FFH_E_ARG);
    if (bin_width < 1 || bin_width > 12) return fail("binSize must be within 1..12", FFH_E_ARG);
    const Pack pk = kPack[enzyme_index];
    // bin of a target = the bin_width bases after the 5' PAM, if any (crispr/BinWriter.scala:58-64)
    const int shift = pk.five_prime ? 2 * (pk.scan - (bin_width + pk.pam)) : 2 * (pk.scan - bin_width);
    const int sshift = shift - 2 * kLookup;
    const uint32_t n_bins = 1u << (2 * bin_width);
    const uint64_t T = n_targets;

    // ---- position offsets + bin membership (stable counting sort by bin; the input order inside a bin is kept) ----
    std::vector<uint64_t> pos_off(T + 1, 0);
    std::vector<uint32_t> bin_cnt(n_bins + 1, 0);
    for (uint64_t i = 0; i < T; ++i) {
        const int c = (int)(int16_t)(targets[i] >> 48);
        if (c <= 0) return fail("Encoded position count should be greater than zero", FFH_E_FORMAT);
        pos_off[i + 1] = pos_off[i] + (uint64_t)c;
        ++bin_cnt[(uint32_t)((targets[i] >> shift) & (n_bins - 1)) + 1];
    }
    if (pos_off[T] != n_positions) return fail("positions array length does not equal the sum of the target counts", FFH_E_FORMAT);
    std::vector<uint64_t> bin_first(n_bins + 1, 0);
    for (uint32_t b = 0; b < n_bins; ++b) bin_first[b + 1] = bin_first[b] + bin_cnt[b + 1];
    bool in_bin_order = true;
    for (uint64_t i = 1; i < T && in_bin_order; ++i)
        in_bin_order = ((targets[i - 1] >> shift) & (n_bins - 1)) <= ((targets[i] >> shift) & (n_bins - 1));
    std::vector<uint64_t> order;  // only when the input is not already grouped by bin (5' PAM enzymes)
    if (!in_bin_order) {
        order.resize(T);
        std::vector<uint64_t> fill(bin_first.begin(), bin_first.end() - 1);
        for (uint64_t i = 0; i < T; ++i) order[fill[(uint32_t)((targets[i] >> shift) & (n_bins - 1))]++] = i;
    }
    auto member_of = [&](uint64_t k) { return in_bin_order ? k : order[k]; };

    // ---- layout of the uncompressed stream: [type][table?][target, positions...]* per bin -------------------------
    std::vector<uint64_t> bin_long(n_bins + 1, 0);  // offset of every bin in longs
    std::vector<uint8_t> indexed(n_bins, 0);
    for (uint32_t b = 0; b < n_bins; ++b) {
        const uint64_t nt = bin_first[b + 1] - bin_first[b];
        uint64_t np = 0;
        if (in_bin_order) np = pos_off[bin_first[b + 1]] - pos_off[bin_first[b]];
        else for (uint64_t k = bin_first[b]; k < bin_first[b + 1]; ++k) np += pos_off[order[k] + 1] - pos_off[order[k]];
        indexed[b] = nt > (uint64_t)kMaxLinear && !pk.five_prime;  // no indexed blocks for 5' PAM enzymes (DatabaseWriter.scala:84-85)
        bin_long[b + 1] = bin_long[b] + 1 + (indexed[b] ? kSub : 0) + nt + np;
    }
    const uint64_t total_bytes = bin_long[n_bins] * 8;
    std::vector<int64_t> stream;
    try { stream.resize(bin_long[n_bins]); } catch (const std::bad_alloc &) { return fail("out of host memory for the database body", FFH_E_NOMEM); }
    parallel_units(n_bins, 16, [&](size_t b0, size_t b1) {
        for (size_t b = b0; b < b1; ++b) {
            int64_t *out = stream.data() + bin_long[b];
            *out++ = indexed[b] ? 2 : 1;
            int64_t *table = out;
            if (indexed[b]) {  // createIndexedBlock :362-413: entry = (first long of the sub-bin << 32) | longs, (-1, 0) when empty
                for (int s = 0; s < kSub; ++s) table[s] = (int64_t)(((uint64_t)(int64_t)-1 << 32) | 0u);
                out += kSub;
            }
            const int64_t *payload = out;
            for (uint64_t k = bin_first[b]; k < bin_first[b + 1]; ++k) {
                const uint64_t i = member_of(k);
                const uint64_t cnt = pos_off[i + 1] - pos_off[i];
                if (indexed[b]) {
                    const int sb = (int)((targets[i] >> sshift) & (kSub - 1));
                    const int64_t e = table[sb];
                    const int first = (int)(e >> 32), size = (int)(uint32_t)e;
                    const int here = (int)(out - payload);
                    table[sb] = (int64_t)(((uint64_t)(int64_t)(first < 0 ? here : first) << 32) | (uint32_t)(size + 1 + (int)cnt));
                }
                *out++ = (int64_t)targets[i];
                std::memcpy(out, positions + pos_off[i], cnt * 8);  // native (little-endian) order, Utils.scala:154-160
                out += cnt;
            }
        }
    });

    // ---- BGZF: every 0xff00 bytes of the stream become one gzip member; deflate in parallel --------------------------
    const size_t n_members = (size_t)((total_bytes + kMember - 1) / kMember);
    std::vector<std::vector<uint8_t>> comp(n_members);
    std::atomic<int> zfail(0);
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(stream.data());
    // htsjdk's default compression level is 5 (Defaults.COMPRESSION_LEVEL); FFH_DEFLATE_LEVEL = 1 .. 9 trades file size for indexing time
    // (the writer is deflate-bound: 3.1 Gb genome, 16 threads: 11.5 s at level 5).  Any level gives a database every reader accepts.
    int level = 5;
    if (const char *e = std::getenv("FFH_DEFLATE_LEVEL")) { const int v = std::atoi(e); if (v >= 1 && v <= 9) level = v; }
    parallel_units(n_members, 8, [&](size_t m0, size_t m1) {
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { zfail = 1; return; }
        std::vector<uint8_t> tmp(kMember + 1024);
        for (size_t m = m0; m < m1; ++m) {
            const size_t off = m * kMember, len = (size_t)std::min<uint64_t>(kMember, total_bytes - off);
            deflateReset(&zs);
            zs.next_in = const_cast<Bytef *>(bytes + off); zs.avail_in = (uInt)len;
            zs.next_out = tmp.data() + 18; zs.avail_out = (uInt)(tmp.size() - 26);
            if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { zfail = 1; break; }
            const size_t clen = zs.total_out, total = 18 + clen + 8;
            static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            std::memcpy(tmp.data(), hdr, 16);
            tmp[16] = (uint8_t)((total - 1) & 0xff); tmp[17] = (uint8_t)((total - 1) >> 8);
            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), bytes + off, (uInt)len), isz = (uint32_t)len;
            std::memcpy(tmp.data() + 18 + clen, &crc, 4);
            std::memcpy(tmp.data() + 22 + clen, &isz, 4);
            comp[m].assign(tmp.begin(), tmp.begin() + (long)total);
        }
        deflateEnd(&zs);
    });
    if (zfail) return fail("deflate failed", FFH_E_IO);

    // ---- concatenate: member file offsets -> virtual pointers (BlockCompressedOutputStream.getPosition) --------------
    FILE *f = std::fopen(db_path, "wb");
    if (!f) return fail(std::string("cannot create ") + db_path, FFH_E_IO);
    std::vector<uint64_t> coff(n_members + 1, 0);
    for (size_t m = 0; m < n_members; ++m) {
        coff[m + 1] = coff[m] + comp[m].size();
        if (std::fwrite(comp[m].data(), 1, comp[m].size(), f) != comp[m].size()) { std::fclose(f); return fail("short write", FFH_E_IO); }
        std::vector<uint8_t>().swap(comp[m]);
    }
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (std::fwrite(eof, 1, 28, f) != 28 || std::fclose(f) != 0) return fail("short write", FFH_E_IO);
    FILE *h = std::fopen((std::string(db_path) + ".header").c_str(), "w");
    if (!h) return fail(std::string("cannot create ") + db_path + ".header", FFH_E_IO);
    std::fprintf(h, "%lld\n1\n%d\n%u\n", 0x1234ABCDE123890LL, enzyme_index, n_bins);  // BinaryHeader.writeHeader :69-97
    for (uint32_t b = 0; b < n_bins; ++b) {
        // the stream position when the bin starts: a full buffer has already been flushed, so (next member, 0) -- never (member, 0xff00)
        const uint64_t byte = bin_long[b] * 8, m = byte / kMember, within = byte % kMember;
        const uint64_t vptr = (coff[m] << 16) | within;
        std::fprintf(h, "%s=%llu,%llu,%u\n", bin_name(bin_width, b).c_str(), (unsigned long long)vptr, (unsigned long long)((bin_long[b + 1] - bin_long[b]) * 8),
                     (unsigned)(bin_first[b + 1] - bin_first[b]));
    }
    for (uint32_t c = 0; c < n_contigs; ++c) std::fprintf(h, "%s=%u\n", contigs[c], c + 1);
    if (std::fclose(h) != 0) return fail("short write", FFH_E_IO);
    return FFH_OK;
} catch (...) {   // (no C++ exception crosses the C ABI: csrc/ffh_abi_guard.hpp; an open FILE is left to the process)
    try { return fail("out of host memory writing the database", FFH_E_NOMEM); } catch (...) { return FFH_E_NOMEM; }
}
