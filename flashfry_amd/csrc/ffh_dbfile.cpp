// ffh_dbfile.cpp -- see ffh_dbfile.hpp.  Host C++ only (zlib + std::thread).
#include "ffh_dbfile.hpp"

#include <zlib.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
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

struct Member { size_t coff, clen_total, cdata_off, cdata_len; uint32_t isize, crc; uint64_t uoff; };

std::string read_db_bins(const std::string &path, const DbHeader &h, uint32_t b0, uint32_t b1, std::vector<int64_t> &longs,
                         std::vector<uint64_t> &offs) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return "cannot open database " + path;
    std::fseek(f, 0, SEEK_END);
    const long fsz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> raw((size_t)fsz);
    if (fsz && std::fread(raw.data(), 1, (size_t)fsz, f) != (size_t)fsz) { std::fclose(f); return "short read on " + path; }
    std::fclose(f);
    // pass 1: walk the members (each carries its own compressed and uncompressed size)
    std::vector<Member> ms;
    uint64_t utotal = 0;
    for (size_t off = 0; off + 18 <= (size_t)fsz;) {
        const uint8_t *p = raw.data() + off;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return "bad BGZF member in " + path;
        const int xlen = p[10] | (p[11] << 8);
        int bsize = -1;
        for (int x = 0; x + 4 <= xlen;) {
            const uint8_t *s = p + 12 + x;
            const int slen = s[2] | (s[3] << 8);
            if (s[0] == 'B' && s[1] == 'C' && slen == 2) bsize = (s[4] | (s[5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 0 || off + (size_t)bsize > (size_t)fsz) return "BGZF member without BC subfield in " + path;
        Member m;
        m.coff = off; m.clen_total = (size_t)bsize; m.cdata_off = off + 12 + (size_t)xlen; m.cdata_len = (size_t)bsize - 12 - (size_t)xlen - 8;
        std::memcpy(&m.crc, p + bsize - 8, 4);
        std::memcpy(&m.isize, p + bsize - 4, 4);
        m.uoff = utotal;
        utotal += m.isize;
        ms.push_back(m);
        off += (size_t)bsize;
    }
    // which uncompressed range do we need?
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
    std::vector<uint64_t> lin(b1 - b0);
    uint64_t need_lo = UINT64_MAX, need_hi = 0, total_longs = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        if (h.uncompressed_bytes[b] % 8) return "bin size is not a multiple of 8";
        if (!linear(h.virtual_offset[b], lin[b - b0])) return "bin pointer does not address a BGZF member";
        if (lin[b - b0] + h.uncompressed_bytes[b] > utotal) return "bin runs past the end of the database";
        need_lo = std::min(need_lo, lin[b - b0]);
        need_hi = std::max(need_hi, lin[b - b0] + h.uncompressed_bytes[b]);
        total_longs += h.uncompressed_bytes[b] / 8;
    }
    if (b1 == b0) { longs.clear(); offs.assign(1, 0); return ""; }
    // pass 2: inflate the members that overlap [need_lo, need_hi) in parallel
    std::vector<uint8_t> data((size_t)(need_hi - need_lo));
    std::atomic<size_t> next_member(0);
    std::atomic<int> failed(0);
    auto worker = [&]() {
        std::vector<uint8_t> tmp(65536 + 64);
        for (;;) {
            const size_t i = next_member.fetch_add(1);
            if (i >= ms.size()) break;
            const Member &m = ms[i];
            if (m.isize == 0 || m.uoff + m.isize <= need_lo || m.uoff >= need_hi) continue;
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            inflateInit2(&zs, -15);
            zs.next_in = raw.data() + m.cdata_off; zs.avail_in = (uInt)m.cdata_len;
            zs.next_out = tmp.data(); zs.avail_out = (uInt)tmp.size();
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.total_out != m.isize || (uint32_t)crc32(crc32(0L, Z_NULL, 0), tmp.data(), m.isize) != m.crc) { failed = 1; continue; }
            const uint64_t s = std::max<uint64_t>(m.uoff, need_lo), e = std::min<uint64_t>(m.uoff + m.isize, need_hi);
            std::memcpy(data.data() + (s - need_lo), tmp.data() + (s - m.uoff), (size_t)(e - s));
        }
    };
    unsigned nthreads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (failed) return "BGZF inflate / crc failure in " + path;
    // bytes -> longs in native (little-endian) order, utils/Utils.scala:167-186
    longs.resize((size_t)total_longs);
    offs.assign((size_t)(b1 - b0) + 1, 0);
    uint64_t w = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        offs[b - b0] = w;
        std::memcpy(longs.data() + w, data.data() + (lin[b - b0] - need_lo), (size_t)h.uncompressed_bytes[b]);
        w += h.uncompressed_bytes[b] / 8;
    }
    offs[b1 - b0] = w;
    return "";
}

std::string decode_blocks(const int64_t *longs, const uint64_t *offs, uint32_t n_bins, std::vector<uint64_t> &targets,
                          std::vector<uint64_t> &positions) {
    const int kSub = 256;  // 4^4 sub-bins, BlockManager.scala:40-49
    for (uint32_t b = 0; b < n_bins; ++b) {
        const int64_t *blk = longs + offs[b];
        uint64_t n = offs[b + 1] - offs[b];
        if (n == 0) return "empty block for bin " + std::to_string(b);  // BlockManager.scala:72 would fail
        const int64_t type = blk[0];
        ++blk; --n;
        if (type == 2) {  // indexed: validate the table the way compareIndexedBlock walks it (:160-170), then the payload is contiguous
            if (n < (uint64_t)kSub) return "indexed block shorter than its lookup table";
            long long last_pos = 0, last_size = 0;
            uint64_t covered = 0;
            for (int i = 0; i < kSub; ++i) {
                const int pos = (int)(blk[i] >> 32);
                const int size = (int)((int64_t)((uint64_t)blk[i] << 32) >> 32);
                if (last_pos != 0 && pos >= 0 && pos != last_pos + last_size) return "indexed block: sub-bin table is not contiguous";
                last_pos = pos > 0 ? pos : 0;
                last_size = size;
                if (pos >= 0 && size > 0) {
                    if ((uint64_t)pos + (uint64_t)size > n - kSub) return "indexed block: sub-bin slice out of range";
                    covered += (uint64_t)size;
                }
            }
            if (covered != n - kSub) return "indexed block: sub-bin sizes do not cover the payload";
            blk += kSub; n -= kSub;
        } else if (type != 1) {
            return "Invalid bin type, unknown value: " + std::to_string((long long)type);  // :85-87
        }
        for (uint64_t off = 0; off < n;) {  // compareLinearBlock :225-252
            const uint64_t t = (uint64_t)blk[off];
            const int count = (int)(int16_t)(t >> 48);
            if (count <= 0) return "Encoded position count should be greater than zero";
            if (n < off + (uint64_t)count + 1) return "Failed to correctly parse block, the number of position entries exceeds the buffer size";
            targets.push_back(t);
            for (int k = 0; k < count; ++k) positions.push_back((uint64_t)blk[off + 1 + k]);
            off += (uint64_t)count + 1;
        }
    }
    return "";
}

}  // namespace ffh
