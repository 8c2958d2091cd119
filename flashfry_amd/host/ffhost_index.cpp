// ffhost_index.cpp -- `index`: build the on-disk off-target database from a reference FASTA, in the reference's format
// (text <db>.header, BinaryHeader.scala:69-97, + BGZF body written bin by bin, DatabaseWriter.scala:58-111).
// CPU code: this is the "next" row of the scope table, needed so that a database can exist on a
// box without a JVM.  The GPU takes over again at `discover`.
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>

#include "ffhost.hpp"

namespace ffhost {

namespace {

// BGZF writer (htsjdk BlockCompressedOutputStream; SAM spec 4.1): gzip members with a BC extra field, <= 64 KiB each
struct BgzfWriter {
    static constexpr int kBlock = 0xff00;
    FILE *f = nullptr;
    std::vector<uint8_t> buf;
    uint64_t blockAddress = 0;
    explicit BgzfWriter(const std::string &path) : f(std::fopen(path.c_str(), "wb")) {
        if (!f) throw Error("cannot create " + path);
        buf.reserve(kBlock);
    }
    uint64_t position() const { return (blockAddress << 16) | (uint64_t)buf.size(); }  // getPosition: virtual file pointer
    void flushBlock() {
        if (buf.empty()) return;
        std::vector<uint8_t> out((size_t)kBlock + 1024);
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, 5, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw Error("deflateInit2 failed");
        zs.next_in = buf.data(); zs.avail_in = (uInt)buf.size();
        zs.next_out = out.data() + 18; zs.avail_out = (uInt)(out.size() - 26);
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); throw Error("deflate failed"); }
        const size_t clen = zs.total_out, total = 18 + clen + 8;
        deflateEnd(&zs);
        static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        std::memcpy(out.data(), hdr, 16);
        out[16] = (uint8_t)((total - 1) & 0xff); out[17] = (uint8_t)((total - 1) >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf.data(), (uInt)buf.size()), isz = (uint32_t)buf.size();
        std::memcpy(out.data() + 18 + clen, &crc, 4);
        std::memcpy(out.data() + 22 + clen, &isz, 4);
        if (std::fwrite(out.data(), 1, total, f) != total) throw Error("short write");
        blockAddress += total;
        buf.clear();
    }
    void write(const uint8_t *p, size_t n) {
        while (n) {
            const size_t c = std::min(n, (size_t)kBlock - buf.size());
            buf.insert(buf.end(), p, p + c);
            p += c; n -= c;
            if (buf.size() == (size_t)kBlock) flushBlock();
        }
    }
    void close() {
        flushBlock();
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        std::fwrite(eof, 1, 28, f);
        std::fclose(f);
        f = nullptr;
    }
};

struct TargetPos {  // reference/binary/BlockReader.scala:138-159
    uint64_t target;  // count in the top 16 bits
    uint32_t firstPosition, nPositions;
};

std::string binName(int width, uint32_t idx) {  // utils/BaseCombinationGenerator.scala:33-69
    std::string s((size_t)width, 'A');
    for (int i = 0; i < width; ++i) s[(size_t)i] = "ACGT"[(idx >> (2 * (width - 1 - i))) & 3];
    return s;
}

}  // namespace

void buildOffTargetDatabase(const std::string &reference, const std::string &output, const ParameterPack &pack, int binSize) {
    if (binSize < 1 || binSize > 12) throw Error("binSize must be within 1..12");
    BitEncoding enc(pack);
    BitPosition posEnc;
    std::fprintf(stderr, "Discovering target sites in the input genome file...\n");
    const std::vector<CRISPRSite> sites = findTargetSites(reference, pack, 0, &posEnc);  // BuildOffTargetDatabase.scala:68
    // sort by sequence (CRISPRSite.compare = bases), stable in discovery order; BlockReader.loadBlock :87-135
    std::vector<uint32_t> order(sites.size());
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sites[a].bases < sites[b].bases; });
    std::vector<TargetPos> targets;
    std::vector<uint64_t> positions;
    positions.reserve(sites.size());
    for (size_t i = 0; i < order.size();) {
        size_t j = i;
        while (j < order.size() && sites[order[j]].bases == sites[order[i]].bases) ++j;
        const size_t cnt = std::min<size_t>(j - i, 32767);  // count and position list capped at Short.MaxValue (:147-153)
        TargetPos t;
        t.target = enc.bitEncodeString(sites[order[i]].bases, (int)cnt);
        t.firstPosition = (uint32_t)positions.size();
        t.nPositions = (uint32_t)cnt;
        for (size_t k = 0; k < cnt; ++k) {
            const CRISPRSite &s = sites[order[i + k]];
            positions.push_back(posEnc.encode(s.contig, (uint32_t)s.position, (int)s.bases.size(), s.forwardStrand));
        }
        targets.push_back(t);
        i = j;
    }
    // bin of a target = the binSize bases after the 5' PAM, if any (crispr/BinWriter.scala:58-64)
    const int shift = pack.fivePrimePam ? 2 * (pack.totalScanLength - (binSize + pack.pamLength)) : 2 * (pack.totalScanLength - binSize);
    const uint32_t nBins = 1u << (2 * binSize);
    std::vector<std::vector<uint32_t>> perBin(nBins);
    for (uint32_t i = 0; i < targets.size(); ++i) perBin[(uint32_t)((targets[i].target >> shift) & (nBins - 1))].push_back(i);
    std::fprintf(stderr, "Creating the final binary database file...\n");
    BgzfWriter w(output);
    struct Off { uint64_t vpos, bytes; uint32_t n; };
    std::vector<Off> offs(nBins);
    std::vector<int64_t> block;
    for (uint32_t b = 0; b < nBins; ++b) {  // DatabaseWriter.scala:76-97
        const auto &idx = perBin[b];
        block.clear();
        const bool indexed = idx.size() > 500 && !pack.fivePrimePam;  // maxTargetsPerLinearBin = 500; no indexed blocks for Cpf1 (:84-85)
        if (!indexed) {  // BlockManager.createLinearBlock :424-442
            block.push_back(1);
            for (uint32_t i : idx) {
                block.push_back((int64_t)targets[i].target);
                for (uint32_t k = 0; k < targets[i].nPositions; ++k) block.push_back((int64_t)positions[targets[i].firstPosition + k]);
            }
        } else {  // BlockManager.createIndexedBlock :362-413 (4-base sub-bins)
            const int lookup = 4, nsub = 256, sshift = shift - 2 * lookup;
            std::vector<int> first(nsub, -1), size(nsub, 0);
            std::vector<int64_t> payload;
            int cur = 0;
            for (uint32_t i : idx) {
                const int sb = (int)((targets[i].target >> sshift) & (nsub - 1));
                if (first[sb] >= cur || first[sb] < 0) first[sb] = cur;
                cur += 1 + (int)targets[i].nPositions;
                size[sb] += 1 + (int)targets[i].nPositions;
                payload.push_back((int64_t)targets[i].target);
                for (uint32_t k = 0; k < targets[i].nPositions; ++k) payload.push_back((int64_t)positions[targets[i].firstPosition + k]);
            }
            block.push_back(2);
            for (int s = 0; s < nsub; ++s) block.push_back((int64_t)(((uint64_t)(int64_t)first[s] << 32) | (uint64_t)(int64_t)size[s]));
            block.insert(block.end(), payload.begin(), payload.end());
        }
        offs[b] = {w.position(), (uint64_t)block.size() * 8, (uint32_t)idx.size()};
        w.write(reinterpret_cast<const uint8_t *>(block.data()), block.size() * 8);  // native (little-endian) order, Utils.scala:154-160
    }
    w.close();
    FILE *h = std::fopen((output + ".header").c_str(), "w");
    if (!h) throw Error("cannot create " + output + ".header");
    std::fprintf(h, "%lld\n1\n%d\n%u\n", 0x1234ABCDE123890LL, pack.index, nBins);  // BinaryHeader.writeHeader :69-97
    for (uint32_t b = 0; b < nBins; ++b)
        std::fprintf(h, "%s=%llu,%llu,%u\n", binName(binSize, b).c_str(), (unsigned long long)offs[b].vpos, (unsigned long long)offs[b].bytes, offs[b].n);
    for (size_t c = 0; c < posEnc.contigs().size(); ++c) std::fprintf(h, "%s=%zu\n", posEnc.contigs()[c].c_str(), c + 1);
    std::fclose(h);
    std::fprintf(stderr, "Wrote %zu unique targets (%zu sites) into %u bins\n", targets.size(), sites.size(), nBins);
}

}  // namespace ffhost
