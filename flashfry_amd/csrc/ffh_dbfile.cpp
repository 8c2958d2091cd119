// ffh_dbfile.cpp -- see ffh_dbfile.hpp.  Host side of the ingest: header parse, BGZF member directory, parallel
// inflate into page-locked buffers with the copies to the device overlapped (zlib + std::thread + HIP streams).
#include "ffh_dbfile.hpp"
#include "ffh_streams.hpp"

#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

namespace ffh {

static const long long kMagic = 0x1234ABCDE123890LL;  // reference/binary/BinaryConstants.scala:26

static std::string bin_name(int width, uint32_t idx) {  // utils/BaseCombinationGenerator.scala:33-69
    std::string s((size_t)width, 'A');
    for (int i = 0; i < width; ++i) s[(size_t)i] = "ACGT"[(idx >> (2 * (width - 1 - i))) & 3];
    return s;
}

std::string read_db_header(const std::string &path, DbHeader &h) {  // BinaryHeader.readHeader :115-160
    std::ifstream f(path);
    if (!f) return "cannot open database header " + path;
    std::string line;
    auto next = [&](std::string &l) { if (!std::getline(f, l)) return false; while (!l.empty() && (l.back() == '\r' || l.back() == '\n')) l.pop_back(); return true; };
    if (!next(line) || std::atoll(line.c_str()) != kMagic) return "Binary file " + path + " doesn't have the magic number expected at the top of the file";
    if (!next(line) || std::atoll(line.c_str()) != 1) return "Binary file " + path + " doesn't have the correct version, expecting 1";
    if (!next(line)) return "truncated header " + path;
    h.enzyme_index = std::atoi(line.c_str());
    if (h.enzyme_index < 1 || h.enzyme_index > 6) return "Unable to find the correct parameter pack for enzyme: " + line;
    if (!next(line)) return "truncated header " + path;
    const long long nb = std::atoll(line.c_str());
    if (nb < 1 || nb > (1LL << 26)) return "bad bin count in " + path;
    h.bin_width = (int)(std::log((double)nb) / std::log(4.0));  // :132
    h.n_bins = 1u << (2 * h.bin_width);
    h.virtual_offset.assign(h.n_bins, 0);
    h.uncompressed_bytes.assign(h.n_bins, 0);
    h.n_targets.assign(h.n_bins, 0);
    for (uint32_t b = 0; b < h.n_bins; ++b) {
        if (!next(line)) return "Missing line for bin " + bin_name(h.bin_width, b);
        const size_t eq = line.find('=');
        unsigned long long v = 0, u = 0;
        unsigned nt = 0;
        if (eq == std::string::npos || std::sscanf(line.c_str() + eq + 1, "%llu,%llu,%u", &v, &u, &nt) != 3)
            return "Missing line for bin " + bin_name(h.bin_width, b) + " from line: " + line;
        if (line.substr(0, eq) != bin_name(h.bin_width, b))
            return "Failed to verify bin name, expected: " + bin_name(h.bin_width, b) + " isn't what we got " + line.substr(0, eq);  // :147
        h.virtual_offset[b] = v; h.uncompressed_bytes[b] = u; h.n_targets[b] = nt;
    }
    while (next(line)) {  // contig ids are re-assigned in file order (:152-155)
        if (line.empty()) continue;
        h.contigs.push_back(line.substr(0, line.find('=')));
    }
    return "";
}

BodyFile::~BodyFile() {
    if (data && data != (const uint8_t *)MAP_FAILED) munmap((void *)data, size);
    if (fd >= 0) close(fd);
}

std::string open_body(const std::string &path, BodyFile &bf) {
    bf.fd = open(path.c_str(), O_RDONLY);
    if (bf.fd < 0) return "cannot open database " + path;
    struct stat sb;
    if (fstat(bf.fd, &sb) != 0) return "cannot stat database " + path;
    bf.size = (size_t)sb.st_size;
    if (bf.size) {
        void *m = mmap(nullptr, bf.size, PROT_READ, MAP_PRIVATE, bf.fd, 0);
        if (m == MAP_FAILED) { bf.data = nullptr; return "cannot map database " + path; }
        bf.data = (const uint8_t *)m;
        (void)madvise(m, bf.size, MADV_WILLNEED);
    }
    // walk the members: each carries its own compressed and uncompressed size (SAM spec 4.1: gzip member, 'BC' extra subfield)
    for (size_t off = 0; off + 18 <= bf.size;) {
        const uint8_t *p = bf.data + off;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return "bad BGZF member in " + path;
        const int xlen = p[10] | (p[11] << 8);
        int bsize = -1;
        for (int x = 0; x + 4 <= xlen && off + 12 + (size_t)x + 6 <= bf.size;) {
            const uint8_t *sf = p + 12 + x;
            const int slen = sf[2] | (sf[3] << 8);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2) bsize = (sf[4] | (sf[5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || off + (size_t)bsize > bf.size) return "BGZF member without BC subfield in " + path;
        Member m;
        m.cdata_off = off + 12 + (size_t)xlen; m.cdata_len = (size_t)bsize - 12 - (size_t)xlen - 8;
        std::memcpy(&m.crc, p + bsize - 8, 4);
        std::memcpy(&m.isize, p + bsize - 4, 4);
        if (m.isize > 65536) return "BGZF member larger than 64 KiB in " + path;
        m.coff = off;
        m.uoff = bf.utotal;
        bf.utotal += m.isize;
        bf.members.push_back(m);
        off += (size_t)bsize;
    }
    return "";
}

std::string locate_bins(const BodyFile &bf, const DbHeader &h, uint32_t b0, uint32_t b1, uint64_t &need_lo, uint64_t &need_hi,
                        std::vector<uint64_t> &bin_off, std::vector<uint64_t> &bin_len) {
    const std::vector<Member> &ms = bf.members;
    auto linear = [&](uint64_t vptr, uint64_t &out) -> bool {  // BlockCompressedInputStream.seek(virtual pointer)
        const uint64_t c = vptr >> 16, within = vptr & 0xffff;
        size_t lo = 0, hi = ms.size();
        while (lo < hi) { size_t mid = (lo + hi) / 2; if (ms[mid].coff < c) lo = mid + 1; else hi = mid; }
        if (lo == ms.size() || ms[lo].coff != c) return false;
        out = ms[lo].uoff + within;
        return true;
    };
    if (b1 == 0 || b1 > h.n_bins) b1 = h.n_bins;
    if (b0 > b1) return "bad bin range";
    bin_off.assign(b1 - b0, 0);
    bin_len.assign(b1 - b0, 0);
    need_lo = UINT64_MAX; need_hi = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        if (h.uncompressed_bytes[b] % 8) return "bin size is not a multiple of 8";
        uint64_t lin = 0;
        if (!linear(h.virtual_offset[b], lin)) return "bin pointer does not address a BGZF member";
        if (lin + h.uncompressed_bytes[b] > bf.utotal) return "bin runs past the end of the database";
        bin_off[b - b0] = lin;
        bin_len[b - b0] = h.uncompressed_bytes[b] / 8;
        need_lo = std::min(need_lo, lin);
        need_hi = std::max(need_hi, lin + h.uncompressed_bytes[b]);
    }
    if (b1 == b0) { need_lo = need_hi = 0; return ""; }
    for (auto &o : bin_off) {
        if ((o - need_lo) % 8) return "bin payloads are not 8-byte aligned to each other";
        o = (o - need_lo) / 8;
    }
    return "";
}

// CPUs this process may actually use: hardware threads, the affinity mask and the cgroup CPU quota (a container with
// `cpu.max = 1600000 100000` owns 16 CPUs however many the machine has; more inflate threads than that only add set-up)
unsigned usable_cpus() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    auto quota = [&](const char *path, const char *period_path) {
        FILE *f = std::fopen(path, "r");
        if (!f) return;
        char a[64] = {0};
        long long q = -1, per = 100000;
        if (period_path) {  // cgroup v1: two files
            if (std::fscanf(f, "%lld", &q) != 1) q = -1;
            if (FILE *g = std::fopen(period_path, "r")) { if (std::fscanf(g, "%lld", &per) != 1) per = 100000; std::fclose(g); }
        } else if (std::fscanf(f, "%63s %lld", a, &per) >= 1 && std::strcmp(a, "max") != 0) {
            q = std::atoll(a);
        }
        std::fclose(f);
        if (q > 0 && per > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (q + per - 1) / per));
    };
    quota("/sys/fs/cgroup/cpu.max", nullptr);
    quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    if (const char *e = std::getenv("FFH_LOAD_THREADS")) { const long v = std::atol(e); if (v > 0) n = (unsigned)v; }
    return std::min(n, 128u);
}

// Members are independent deflate streams: every host thread inflates groups of consecutive members into its own slice
// of one page-locked arena and queues the copy to the device on its own stream, so inflate, PCIe and the next inflate
// overlap.  All HIP objects are created once by the calling thread (driver calls serialise; per-thread creation cost more
// than the inflate at 128 threads).
void member_range(const BodyFile &bf, uint64_t need_lo, uint64_t need_hi, size_t &m0, size_t &m1) {  // members overlapping [need_lo, need_hi)
    const std::vector<Member> &ms = bf.members;
    size_t lo = 0, hi = ms.size();
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (ms[mid].uoff + ms[mid].isize <= need_lo) lo = mid + 1; else hi = mid; }
    m0 = lo;
    lo = m0; hi = ms.size();
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (ms[mid].uoff < need_hi) lo = mid + 1; else hi = mid; }
    m1 = lo;
}

std::string inflate_to_device(const BodyFile &bf, uint64_t need_lo, uint64_t need_hi, uint8_t *d_raw, int device, IngestStats &stats, bool raw_copy, bool force_pipeline) {
    const std::vector<Member> &ms = bf.members;
    stats.threads = 0; stats.compressed_bytes = 0; stats.raw_bytes = need_hi - need_lo;
    if (need_hi <= need_lo) return "";
    size_t m0 = 0, m1 = ms.size();
    member_range(bf, need_lo, need_hi, m0, m1);
    if (m1 <= m0) return "";
    const auto count_compressed = [&]() { for (size_t i = m0; i < m1; ++i) stats.compressed_bytes += ms[i].cdata_len + 26; };   // (every path below accounts once)
    const size_t file_lo = ms[m0].coff;  // raw_copy: the members' file bytes go to d_raw + (file offset - file_lo), still compressed
    if (!raw_copy && need_hi - need_lo <= kSmallBodyBytes && !force_pipeline) {   // (compared here: the INFLATED size of what is needed)
        // a small body (a chr22-scale database: 57 MB -> 100 MB): the host threads inflate into one pageable buffer and ONE copy takes it
        // over.  The page-locked arena, the 16 streams and their teardown cost 0.15 s there, the device's inflate kernel 0.1 s whatever
        // the size (a member is decoded by one lane); this is 0.03 s (profiles/r05/ab_log.txt 12).
        const uint64_t u0 = ms[m0].uoff, u1 = ms[m1 - 1].uoff + ms[m1 - 1].isize;
        std::unique_ptr<uint8_t[]> buf(new (std::nothrow) uint8_t[(size_t)(u1 - u0)]);   // (not value-initialised: first touched by the thread that fills it)
        if (!buf) return "out of host memory inflating the database body";
        count_compressed();
        const unsigned nthreads = (unsigned)std::max<size_t>(1, std::min<size_t>(usable_cpus(), (m1 - m0 + 15) / 16));
        stats.threads = nthreads;
        std::atomic<size_t> next(m0);
        std::atomic<int> bad(0);
        auto worker = [&]() {
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
            while (!bad) {
                const size_t a = next.fetch_add(8);
                if (a >= m1) break;
                for (size_t i = a; i < std::min(m1, a + 8); ++i) {
                    const Member &m = ms[i];
                    if (m.isize == 0) continue;
                    inflateReset(&zs);
                    zs.next_in = const_cast<Bytef *>(bf.data + m.cdata_off); zs.avail_in = (uInt)m.cdata_len;
                    zs.next_out = buf.get() + (m.uoff - u0); zs.avail_out = (uInt)m.isize;
                    const int rc = inflate(&zs, Z_FINISH);
                    if (rc != Z_STREAM_END || zs.total_out != m.isize || (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf.get() + (m.uoff - u0), m.isize) != m.crc) bad = 1;
                }
            }
            inflateEnd(&zs);
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
        if (bad) return "BGZF inflate / crc failure in the database body";
        hipError_t he = hipSetDevice(device);
        if (he == hipSuccess) he = hipMemcpy(d_raw, buf.get() + (need_lo - u0), (size_t)(need_hi - need_lo), hipMemcpyHostToDevice);
        return he == hipSuccess ? "" : std::string("copy to the device failed: ") + hipGetErrorString(he);
    }
    if (raw_copy) {
        // the device inflates: a small body goes over in ONE copy straight from the file mapping (compared here: the COMPRESSED span
        // [file_lo, file_hi) against kSmallBodyBytes) -- the page-locked arena, the 16 streams and the loader threads of the pipeline below
        // pay from a few hundred MB on (3.2 GB of hg38 in 0.1 s)
        const size_t file_hi = ms[m1 - 1].cdata_off + ms[m1 - 1].cdata_len + 8;
        if (file_hi - file_lo <= kSmallBodyBytes && !force_pipeline) {
            count_compressed();
            stats.threads = 1;
            hipError_t he = hipSetDevice(device);
            if (he == hipSuccess) he = hipMemcpy(d_raw, bf.data + file_lo, file_hi - file_lo, hipMemcpyHostToDevice);
            return he == hipSuccess ? "" : std::string("copy to the device failed: ") + hipGetErrorString(he);
        }
    }
    constexpr size_t kGroup = 32;                 // members per chunk: <= 2 MiB of payload
    constexpr size_t kChunkBytes = kGroup * 65536;
    const size_t nchunks = (m1 - m0 + kGroup - 1) / kGroup;
    const unsigned nthreads = (unsigned)std::max<size_t>(1, std::min<size_t>(usable_cpus(), nchunks));
    stats.threads = nthreads;
    count_compressed();

    struct Lane { hipStream_t st = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; uint8_t *buf[2] = {nullptr, nullptr}; };
    std::vector<Lane> lanes(nthreads);
    uint8_t *arena = nullptr;
    auto release = [&]() {
        for (auto &l : lanes) { for (auto e : l.ev) if (e) (void)hipEventDestroy(e); ffh::stream_pool().release(device, l.st); }
        if (arena) (void)hipHostFree(arena);
    };
    hipError_t he = hipSetDevice(device);
    if (he == hipSuccess) he = hipHostMalloc((void **)&arena, (size_t)nthreads * 2 * kChunkBytes, hipHostMallocDefault);
    for (unsigned t = 0; t < nthreads && he == hipSuccess; ++t) {
        he = ffh::stream_pool().acquire(device, &lanes[t].st);   // (pooled, never destroyed: ffh_streams.hpp)
        for (int i = 0; i < 2 && he == hipSuccess; ++i) {
            lanes[t].buf[i] = arena + ((size_t)t * 2 + (size_t)i) * kChunkBytes;
            he = hipEventCreateWithFlags(&lanes[t].ev[i], hipEventDisableTiming);
        }
    }
    if (he != hipSuccess) { release(); return std::string("loader set-up failed: ") + hipGetErrorString(he); }

    std::atomic<size_t> next_chunk(0);
    std::atomic<int> failed(0);
    std::mutex err_mu;
    std::string err;
    auto fail = [&](const std::string &m) { std::lock_guard<std::mutex> g(err_mu); if (err.empty()) err = m; failed = 1; };
    const bool verbose = std::getenv("FFH_VERBOSE") != nullptr;
    std::atomic<long long> us_inflate(0), us_wait(0);
    auto now_us = []() { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const long long t_start = now_us();
    auto worker = [&](unsigned lane_id) {
        Lane &ln = lanes[lane_id];
        if (hipSetDevice(device) != hipSuccess) { fail("hipSetDevice failed in a loader thread"); return; }
        bool used[2] = {false, false};
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { fail("zlib set-up failed"); return; }
        int slot = 0;
        while (!failed) {
            const size_t c = next_chunk.fetch_add(1);
            if (c >= nchunks) break;
            const long long w1 = now_us();
            if (used[slot] && hipEventSynchronize(ln.ev[slot]) != hipSuccess) { fail("copy to the device failed"); break; }
            const long long w2 = now_us();
            us_wait += w2 - w1;
            const size_t a = m0 + c * kGroup, b = std::min(m1, a + kGroup);
            const uint64_t u0 = ms[a].uoff, u1 = ms[b - 1].uoff + ms[b - 1].isize;
            uint8_t *buf = ln.buf[slot];
            if (raw_copy) {  // no inflate here: the device decodes (ffh_inflate.hpp); the host only moves the file into page-locked memory
                const size_t f0 = ms[a].coff, f1 = ms[b - 1].cdata_off + ms[b - 1].cdata_len + 8;
                std::memcpy(buf, bf.data + f0, f1 - f0);
                us_inflate += now_us() - w2;
                if (hipMemcpyAsync(d_raw + (f0 - file_lo), buf, f1 - f0, hipMemcpyHostToDevice, ln.st) != hipSuccess ||
                    hipEventRecord(ln.ev[slot], ln.st) != hipSuccess) { fail("copy to the device failed"); break; }
                used[slot] = true;
                slot ^= 1;
                continue;
            }
            for (size_t i = a; i < b && !failed; ++i) {
                const Member &m = ms[i];
                if (m.isize == 0) continue;
                inflateReset(&zs);
                zs.next_in = const_cast<Bytef *>(bf.data + m.cdata_off); zs.avail_in = (uInt)m.cdata_len;
                zs.next_out = buf + (m.uoff - u0); zs.avail_out = (uInt)m.isize;
                const int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.total_out != m.isize || (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf + (m.uoff - u0), m.isize) != m.crc)
                    fail("BGZF inflate / crc failure in the database body");
            }
            if (failed) break;
            us_inflate += now_us() - w2;
            const uint64_t s = std::max(u0, need_lo), e = std::min(u1, need_hi);
            if (e > s) {
                if (hipMemcpyAsync(d_raw + (s - need_lo), buf + (s - u0), (size_t)(e - s), hipMemcpyHostToDevice, ln.st) != hipSuccess ||
                    hipEventRecord(ln.ev[slot], ln.st) != hipSuccess) { fail("copy to the device failed"); break; }
                used[slot] = true;
                slot ^= 1;
            }
        }
        if (hipStreamSynchronize(ln.st) != hipSuccess) fail("copy to the device failed");
        inflateEnd(&zs);
    };
    const long long t_ready = now_us();
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto &t : pool) t.join();
    const long long t_done = now_us();
    release();
    if (verbose)
        std::fprintf(stderr, "[ffh ingest] %u threads, %zu chunks: set-up %.1f ms, workers %.1f ms (per-thread mean: inflate %.1f ms, waiting for copies %.1f ms), teardown %.1f ms\n",
                     nthreads, nchunks, (t_ready - t_start) / 1e3, (t_done - t_ready) / 1e3, us_inflate / 1e3 / nthreads, us_wait / 1e3 / nthreads, (now_us() - t_done) / 1e3);
    return failed ? err : "";
}

}  // namespace ffh

extern "C" int ffh_host_threads(void) { return (int)ffh::usable_cpus(); }
