// ffhost_table.cpp -- the discover/score table (targetio/TabDelimitedHandler.scala) and the GPU traverser.
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <exception>
#include <mutex>
#include <thread>

#include "ffhost.hpp"

namespace ffhost {

// fn(begin, end) over [0, n) in dynamic chunks on the CPUs this process may use (ffh_host_threads: affinity + cgroup quota)
static void parallelFor(size_t n, size_t grain, const std::function<void(size_t, size_t)> &fn) {
    const size_t chunks = (n + grain - 1) / grain;
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, ffh_host_threads()), chunks));
    std::atomic<size_t> next(0);
    std::exception_ptr failure;
    std::mutex mu;
    auto work = [&]() {
        try {
            for (;;) {
                const size_t a = next.fetch_add(grain);
                if (a >= n) break;
                fn(a, std::min(n, a + grain));
            }
        } catch (...) {
            std::lock_guard<std::mutex> g(mu);
            if (!failure) failure = std::current_exception();
            next = n;
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (failure) std::rethrow_exception(failure);
}

// ---- output sink: plain file, or gzip when the name ends in .gz (TabDelimitedHandler.scala:112-116) ---------
struct TabDelimitedOutput::Sink {
    FILE *f = nullptr;
    gzFile z = nullptr;
    void put(const std::string &s) {
        if (z) gzwrite(z, s.data(), (unsigned)s.size());
        else std::fwrite(s.data(), 1, s.size(), f);
    }
};

static const char *kDefaultColumns[7] = {"contig", "start", "stop", "target", "context", "overflow", "orientation"};  // :79

TabDelimitedOutput::TabDelimitedOutput(const std::string &outputFile, const BitEncoding &bitEncoding, const BitPosition &bitPosition,
                                       const std::vector<Metric> &scoringModels, bool writeOTs_, bool writePositions_, bool numericOutput)
    : out(new Sink), enc(bitEncoding), pos(bitPosition), models(scoringModels), writeOTs(writeOTs_), writePositions(writePositions_), numeric(numericOutput) {
    const bool gz = outputFile.size() > 3 && outputFile.compare(outputFile.size() - 3, 3, ".gz") == 0;
    if (gz) out->z = gzopen(outputFile.c_str(), "wb");
    else out->f = std::fopen(outputFile.c_str(), "w");
    if (!out->z && !out->f) { delete out; out = nullptr; throw Error("cannot create " + outputFile); }
    std::string h;
    for (int i = 0; i < 7; ++i) h += (i ? "\t" : "") + std::string(kDefaultColumns[i]);
    for (Metric m : models)
        for (const auto &c : metricHeaderColumns(m)) h += "\t" + c;
    h += writeOTs ? "\totCount\toffTargets\n" : "\totCount\n";  // :122-125
    out->put(h);
}

TabDelimitedOutput::~TabDelimitedOutput() { close(); }

void TabDelimitedOutput::close() {
    if (!out) return;
    if (out->z) gzclose(out->z);
    if (out->f) std::fclose(out->f);
    delete out;
    out = nullptr;
}

// CRISPRHit.toOutput, crispr/CRISPRHit.scala:54-88
static void appendHit(std::string &o, const CRISPRHit &hit, const CRISPRSiteOT &g, const BitEncoding &enc, const BitPosition &pos, bool outputPositions) {
    const StringCount sc = enc.bitDecodeString(hit.sequence);
    o += sc.str;
    o += '_';
    o += std::to_string(sc.count);
    o += '_';
    o += std::to_string(enc.mismatches(g.longEncoding, hit.sequence));
    if (!outputPositions) return;
    if (hit.validOffTargetCoordinates && !hit.coordinates.empty()) {
        o += '<';
        for (size_t k = 0; k < hit.coordinates.size(); ++k) {
            const PositionInformation p = pos.decode(hit.coordinates[k]);
            if (k) o += '|';
            o += p.contig;
            o += ':';
            o += std::to_string(p.start);
            o += '^';
            o += p.forwardStrand ? 'F' : 'R';
        }
        o += '>';
    }
    if (hit.hasCfd) o += "{Doench2016CFDScore=" + javaDoubleToString(hit.cfd) + "}";  // toOutputScores :93-104
}

void TabDelimitedOutput::write(const CRISPRSiteOT &g) {
    std::string o;
    format(g, o);
    out->put(o);
}

// rows are independent: format slabs of guides on all usable CPUs, write the slabs in order
void TabDelimitedOutput::writeAll(const std::vector<CRISPRSiteOT> &guides) {
    const size_t slab = 512, wave = 64 * slab;
    std::vector<std::string> text;
    for (size_t base = 0; base < guides.size(); base += wave) {
        const size_t n = std::min(wave, guides.size() - base);
        text.assign((n + slab - 1) / slab, std::string());
        parallelFor(n, slab, [&](size_t a, size_t b) {
            std::string &o = text[a / slab];
            for (size_t i = a; i < b; ++i) format(guides[base + i], o);
        });
        for (const auto &t : text) out->put(t);
    }
}

void TabDelimitedOutput::format(const CRISPRSiteOT &g, std::string &o) const {  // :131-153
    const ParameterPack &p = enc.mParameterPack;
    o.reserve(o.size() + 256 + g.offTargets.size() * 48);
    o += g.target.contig + "\t" + std::to_string(g.target.position) + "\t" + std::to_string(g.target.position + (int)g.target.bases.size()) + "\t" + g.target.bases + "\t";
    o += (g.target.hasContext ? g.target.sequenceContext : std::string("NONE")) + "\t";
    o += ((g.full() || g.inheritedOverflow) ? "OVERFLOW" : "OK");
    o += "\t";
    o += (g.target.forwardStrand ? "FWD" : "RVS");
    o += "\t";
    for (Metric m : models)
        for (const auto &c : metricColumns(m, g, p, numeric)) o += c + "\t";
    long total = 0;
    for (const auto &h : g.offTargets) total += (long)h.nCoordinates;
    o += std::to_string(total);
    if (writeOTs) {
        o += '\t';
        for (size_t i = 0; i < g.offTargets.size(); ++i) {
            if (i) o += ',';
            appendHit(o, g.offTargets[i], g, enc, pos, writePositions);
        }
    }
    o += '\n';
}

// ---- TabDelimitedInput :169-335 ---------------------------------------------------------------------------------
static std::vector<std::string> splitJava(const std::string &s, char sep) {  // String.split drops trailing empty strings
    std::vector<std::string> v;
    size_t a = 0;
    for (;;) {
        const size_t b = s.find(sep, a);
        if (b == std::string::npos) { v.push_back(s.substr(a)); break; }
        v.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    while (!v.empty() && v.back().empty()) v.pop_back();
    return v;
}

static bool readLine(gzFile f, std::string &line) {
    line.clear();
    char buf[1 << 16];
    bool any = false;
    while (gzgets(f, buf, sizeof buf)) {
        any = true;
        line += buf;
        if (!line.empty() && line.back() == '\n') break;
    }
    while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
    return any;
}

static void addOffTarget(CRISPRSiteOT &ot, const std::string &token, int maximumMismatches, const BitPosition &bitPosition, const BitEncoding &bitEncoding) {  // :277-334
    const size_t u1 = token.find('_'), u2 = u1 == std::string::npos ? u1 : token.find('_', u1 + 1);
    if (u2 == std::string::npos) throw Error("Unable to parse off-target token: " + token.substr(0, 60));
    const std::string seq = token.substr(0, u1);
    const int count = std::atoi(token.c_str() + u1 + 1);
    const int mm = std::atoi(token.c_str() + u2 + 1);
    if (mm > maximumMismatches) return;  // :293
    if (count > 32767) throw Error("The count was too large to encode in a Scala Short value");
    CRISPRHit hit;
    hit.sequence = bitEncoding.bitEncodeString(seq, count);
    const size_t lt = token.find('<', u2);
    if (lt != std::string::npos) {  // :296-309
        size_t gt = token.find('>', lt);
        if (gt == std::string::npos) gt = token.size();
        for (const auto &pe : splitJava(token.substr(lt + 1, gt - lt - 1), '|')) {
            const size_t colon = pe.find(':'), caret = pe.find('^');
            if (colon == std::string::npos || caret == std::string::npos) throw Error("Unable to parse position: " + pe);
            hit.coordinates.push_back(bitPosition.encode(pe.substr(0, colon), (uint32_t)std::atol(pe.c_str() + colon + 1), (int)seq.size(), pe.substr(caret + 1) == "F"));
        }
        hit.nCoordinates = (uint32_t)hit.coordinates.size();
    } else {  // zero-filled coordinates, :310-316
        hit.nCoordinates = (uint32_t)count;  // the reference allocates `count` zeros here; only their number is ever used
        hit.validOffTargetCoordinates = false;
    }
    if (!ot.full()) {  // :305-306 -> CRISPRSiteOT.addOT
        ot.currentTotal += (long)hit.nCoordinates;
        ot.offTargets.push_back(std::move(hit));
    }
}

std::vector<CRISPRSiteOT> readTabDelimited(const std::string &inputFile, const BitEncoding &bitEncoding, const BitPosition &bitPosition, int maximumMismatches,
                                           bool filterOutOverflowedGuides) {
    gzFile f = gzopen(inputFile.c_str(), "rb");
    if (!f) throw Error("cannot open " + inputFile);
    std::string line;
    if (!readLine(f, line)) { gzclose(f); throw Error("Header line not long enough for file: " + inputFile); }
    const std::vector<std::string> header = splitJava(line, '\t');
    if (header.size() < 8) { gzclose(f); throw Error("Header line not long enough for file: " + inputFile); }
    for (int i = 0; i < 7; ++i)
        if (header[(size_t)i] != kDefaultColumns[i]) { gzclose(f); throw Error("Mismatched line doesn't contain the standard header tokens: " + inputFile); }
    const size_t nh = header.size();
    const bool withOTs = header[nh - 2] == "otCount" && header[nh - 1] == "offTargets";
    if (!withOTs && header[nh - 1] != "otCount") { gzclose(f); throw Error("Unable to parse out the final columns in the header"); }
    const size_t nAnnot = nh - 7 - (withOTs ? 2 : 1);
    std::vector<std::string> lines;
    while (readLine(f, line))
        if (!line.empty()) lines.push_back(std::move(line));
    gzclose(f);
    // rows are independent (extractCRISPRSiteOT :225-268): parse them on all usable CPUs, keep the file order
    std::vector<CRISPRSiteOT> parsed(lines.size());
    std::vector<uint8_t> keep(lines.size(), 0);
    parallelFor(lines.size(), 256, [&](size_t a0, size_t b0) {
        for (size_t li = a0; li < b0; ++li) {
            const std::vector<std::string> sp = splitJava(lines[li], '\t');
            if (sp.size() < 8 + nAnnot) throw Error("Unable to parse line: " + lines[li].substr(0, 100));
            CRISPRSiteOT &ot = parsed[li];
            ot.target.contig = sp[0];
            ot.target.position = std::atoi(sp[1].c_str());
            ot.target.bases = sp[3];
            ot.target.hasContext = sp[4] != "NONE";
            if (ot.target.hasContext) ot.target.sequenceContext = sp[4];
            ot.target.forwardStrand = sp[6] == "FWD";
            const bool isOverflowed = sp[5] != "OK";
            const int otCount = std::atoi(sp[7 + nAnnot].c_str());
            ot.overflow = isOverflowed ? otCount : otCount + 1;  // :241-245
            ot.inheritedOverflow = isOverflowed;
            ot.longEncoding = bitEncoding.bitEncodeString(sp[3]);
            if (withOTs && sp.size() == nh)
                for (const auto &tok : splitJava(sp.back(), ',')) addOffTarget(ot, tok, maximumMismatches, bitPosition, bitEncoding);
            keep[li] = !filterOutOverflowedGuides || (!ot.inheritedOverflow && !ot.full());  // :259-262
            std::string().swap(lines[li]);
        }
    });
    std::vector<CRISPRSiteOT> guides;
    for (size_t li = 0; li < parsed.size(); ++li)
        if (keep[li]) guides.push_back(std::move(parsed[li]));
    return guides;
}

// ---- header info via the C ABI -------------------------------------------------------------------------------------------
static std::string abiError(ffh_ctx *c) { return ffh_last_error(c); }

HeaderInfo readHeaderInfo(const std::string &databasePath) {
    ffh_ctx *c = ffh_create(0, 0);
    if (!c) throw Error(abiError(nullptr));
    HeaderInfo h;
    if (ffh_db_open_header(c, databasePath.c_str())) { const std::string e = abiError(c); ffh_destroy(c); throw Error(e); }
    ffh_db_info info;
    ffh_db_info_get(c, &info);
    h.enzymeIndex = info.enzyme_index;
    for (uint32_t i = 1;; ++i) {
        const char *n = ffh_db_contig(c, i);
        if (!n) break;
        h.contigs.push_back(n);
    }
    for (uint32_t b = 0; b < info.n_bins; ++b) h.binBytes.push_back(ffh_db_bin_bytes(c, b));
    ffh_destroy(c);
    return h;
}

// ---- GpuTraverser: the replacement of SeekTraverser / LinearTraverser.scan ------------------------------------------------
ScanStats GpuTraverser::scan(const std::string &binaryFile, std::vector<CRISPRSiteOT> &guides, int maxMismatch, int maximumOffTargets,
                             const std::vector<int> &devices, bool wantPositions, const HeaderInfo *header) {
    using clk = std::chrono::steady_clock;
    ScanStats st;
    const size_t ng = guides.size(), nd = std::max<size_t>(devices.size(), 1);
    st.gpus = (int)nd;
    std::vector<uint64_t> longs(ng);
    for (size_t i = 0; i < ng; ++i) longs[i] = guides[i].longEncoding;
    // contiguous bin ranges balanced by payload bytes
    // (the CLI has read the header already: a second read costs a context -- streams, events, device memory -- created and destroyed for it)
    const HeaderInfo own = header ? HeaderInfo() : readHeaderInfo(binaryFile);
    const HeaderInfo &hdr = header ? *header : own;
    const size_t nbins = hdr.binBytes.size();
    std::vector<uint32_t> cut(nd + 1, 0);
    {
        double total = 0, run = 0;
        for (auto b : hdr.binBytes) total += (double)b;
        size_t r = 1;
        for (size_t b = 0; b < nbins && r < nd; ++b) {
            run += (double)hdr.binBytes[b];
            while (r < nd && run >= total * (double)r / (double)nd) cut[r++] = (uint32_t)(b + 1);
        }
        for (; r < nd; ++r) cut[r] = (uint32_t)nbins;
        cut[nd] = (uint32_t)nbins;
    }
    std::vector<ffh_ctx *> ctx(nd, nullptr);
    std::vector<std::string> errs(nd);
    std::vector<ffh_result *> res(nd, nullptr);
    ffh_comm *comm = nullptr;   // the shards' exchange runs inside the library (RCCL over xGMI between distinct devices; ffh_comm_*)
    std::vector<ffh_guide_summary> reduced(ng);
    auto cleanup = [&]() { for (auto r : res) if (r) ffh_result_free(r); if (comm) ffh_comm_destroy(comm); for (auto c : ctx) if (c) ffh_destroy(c); };
    auto parallel = [&](const std::function<void(size_t)> &fn) {
        std::vector<std::thread> th;
        for (size_t d = 1; d < nd; ++d) th.emplace_back(fn, d);
        fn(0);
        for (auto &t : th) t.join();
        for (const auto &e : errs) if (!e.empty()) { cleanup(); throw Error(e); }
    };
    auto tc = clk::now();
    parallel([&](size_t d) {
        ctx[d] = ffh_create(devices.empty() ? 0 : devices[d], 0);
        if (!ctx[d]) errs[d] = abiError(nullptr);
    });
    auto t0 = clk::now();
    parallel([&](size_t d) {
        if (ffh_db_open(ctx[d], binaryFile.c_str(), cut[d], cut[d + 1])) errs[d] = abiError(ctx[d]);
    });
    auto t1 = clk::now();
    ffh_db_load_stats(ctx[0], &st.load);
    // ONE library call: every shard scans all guides (bounded by maximumOffTargets: a guide that reaches it inside a shard is not
    // scanned against the rest of that shard, as the reference stops feeding such a guide, ResultsAggregator.scala:61-69), then the
    // shards exchange their per-guide totals (ordered cut-off across shards) and aggregates -- the collectives are the library's
    const int maxOT = std::max(maximumOffTargets, 0);
    if (ffh_comm_create_local(ctx.data(), (int)nd, &comm)) { const std::string e = ffh_comm_last_error(nullptr); cleanup(); throw Error(e); }
    if (ffh_discover_sharded(comm, longs.data(), (uint32_t)ng, maxMismatch, maxOT, 0u, reduced.data())) { const std::string e = ffh_comm_last_error(comm); cleanup(); throw Error(e); }
    st.transport = ffh_comm_transport(comm);
    auto t2 = clk::now();
    parallel([&](size_t d) {
        // every shard's retained hits under the cut-off continued from the shards before it; without --positionOutput the table
        // prints sequence_count_mismatches only and the position arrays stay on the device
        if (ffh_comm_shard_lists(comm, (int)d, FFH_FINALIZE_NO_HIT_SCORES | (wantPositions ? 0u : FFH_FINALIZE_NO_POSITIONS), &res[d])) errs[d] = ffh_comm_last_error(comm);
    });
    auto t3 = clk::now();
    // deliver the hits in database order = shard order (what aggregator.updateOT would have received)
    parallelFor(ng, 256, [&](size_t g0, size_t g1) {
    for (size_t g = g0; g < g1; ++g) {
        CRISPRSiteOT &ot = guides[g];
        ot.overflow = maximumOffTargets;
        size_t nHits = 0;
        for (size_t d = 0; d < nd; ++d) nHits += (size_t)(ffh_result_guide_offsets(res[d])[g + 1] - ffh_result_guide_offsets(res[d])[g]);
        ot.offTargets.reserve(nHits);
        for (size_t d = 0; d < nd; ++d) {
            const ffh_result *r = res[d];
            const uint64_t *go = ffh_result_guide_offsets(r), *ht = ffh_result_hit_targets(r), *pp = ffh_result_positions(r);
            const uint64_t *po = wantPositions ? ffh_result_pos_offsets(r) : nullptr;
            // discover attaches no per-hit scores (they come from `score`, which calls ffh_score_lists): FFH_FINALIZE_NO_HIT_SCORES
            for (uint64_t h = go[g]; h < go[g + 1]; ++h) {
                CRISPRHit hit;
                hit.sequence = ht[h];
                hit.nCoordinates = (uint32_t)(ht[h] >> 48);   // the occurrence count rides in bits 63:48 (BitEncoding.scala:46-67)
                if (wantPositions) hit.coordinates.assign(pp + po[h], pp + po[h + 1]);
                hit.hasCfd = false;
                ot.currentTotal += (long)hit.nCoordinates;
                ot.offTargets.push_back(std::move(hit));
            }
        }
        const ffh_guide_summary &sum = reduced[g];   // reduced over the shards by the library (f64 sums added in shard = database order)
        ot.summary = sum;
    }
    });
    st.deliverMs = std::chrono::duration<double, std::milli>(clk::now() - t3).count();
    for (size_t d = 0; d < nd; ++d) {
        ffh_timings tm;
        ffh_get_timings(ctx[d], &tm);
        st.executedComparisons += tm.pairs_prefix + tm.pairs_suffix;
        ffh_db_info info;
        ffh_db_info_get(ctx[d], &info);
        st.targets += info.n_targets;
        st.positions += info.n_positions;
    }
    st.createMs = std::chrono::duration<double, std::milli>(t0 - tc).count();
    st.loadMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
    st.scanMs = std::chrono::duration<double, std::milli>(t2 - t1).count();
    st.finalizeMs = std::chrono::duration<double, std::milli>(t3 - t2).count();
    cleanup();
    return st;
}

}  // namespace ffhost
