// ffhost_cli.cpp -- `flashfry-hip index | discover | score`: the reference's CLI surface for the accelerated path
// (Main.scala:51-57; options accept both -x and --x spellings like the picocli annotations of the reference).
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>

#include "ffhost.hpp"

namespace ffhost {

// ---- tiny option parser ----------------------------------------------------------------------------------------
struct Options {
    std::map<std::string, std::string> kv;
    bool has(const std::string &k) const { return kv.count(k) != 0; }
    std::string str(const std::string &k, const std::string &d = "") const { auto it = kv.find(k); return it == kv.end() ? d : it->second; }
    int num(const std::string &k, int d) const { return has(k) ? std::atoi(kv.at(k).c_str()) : d; }
    double real(const std::string &k, double d) const { return has(k) ? std::atof(kv.at(k).c_str()) : d; }
};

static Options parse(int argc, char **argv, const std::vector<std::string> &flags, const std::vector<std::string> &valued) {
    Options o;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        while (!a.empty() && a[0] == '-') a.erase(0, 1);
        std::string val;
        const size_t eq = a.find('=');
        if (eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); }
        if (std::find(flags.begin(), flags.end(), a) != flags.end()) { o.kv[a] = "1"; continue; }
        if (std::find(valued.begin(), valued.end(), a) != valued.end()) {
            if (eq == std::string::npos) {
                if (i + 1 >= argc) throw Error("Missing value for option --" + a);
                val = argv[++i];
            }
            o.kv[a] = val;
            continue;
        }
        throw Error("Unknown option: " + std::string(argv[i]));
    }
    return o;
}

static void require(const Options &o, const std::vector<std::string> &keys) {
    for (const auto &k : keys)
        if (!o.has(k)) throw Error("Missing required option: --" + k);
}

// --gpus N = devices 0..N-1; --devices 0,1,1 names them explicitly (a device may appear twice: two bin shards on it)
static std::vector<int> deviceList(const Options &o) {
    std::vector<int> d;
    if (o.has("devices")) {
        const std::string s = o.str("devices");
        for (size_t a = 0; a <= s.size();) {
            size_t b = s.find(',', a);
            if (b == std::string::npos) b = s.size();
            if (b > a) d.push_back(std::atoi(s.substr(a, b - a).c_str()));
            a = b + 1;
        }
    }
    if (d.empty()) {
        const int n = std::max(1, o.num("gpus", 1));
        for (int i = 0; i < n; ++i) d.push_back(i);
    }
    return d;
}

// ---- index: modules/BuildOffTargetDatabase.scala:57-89 --------------------------------------------------------------
int runIndex(int argc, char **argv) {
    const Options o = parse(argc, argv, {}, {"reference", "database", "tmpLocation", "enzyme", "binSize", "devices", "gpus"});
    require(o, {"reference", "database"});  // --tmpLocation is accepted and ignored: the sites are sorted in device memory
    const ParameterPack &pack = ParameterPack::nameToParameterPack(o.str("enzyme", "spCas9ngg"));
    buildOffTargetDatabase(o.str("reference"), o.str("database"), pack, o.num("binSize", 7), deviceList(o)[0]);
    return 0;
}

// ---- discover: modules/OffTargetDiscovery.scala:79-153 --------------------------------------------------------------
int runDiscover(int argc, char **argv) {
    const Options o = parse(argc, argv, {"positionOutput", "forceLinear"},
                            {"fasta", "database", "output", "maxMismatch", "flankingSequence", "maximumOffTargets", "minGC", "maxGC", "gpus", "devices"});
    require(o, {"fasta", "database", "output"});
    const int maxMismatch = o.num("maxMismatch", 4), flank = o.num("flankingSequence", 6), maxOT = o.num("maximumOffTargets", 2000);
    const double minGC = o.real("minGC", 0.0), maxGC = o.real("maxGC", 1.0);
    if (!(minGC >= 0 && minGC <= 1.0) || !(maxGC >= 0 && maxGC <= 1.0)) throw Error("minGC / maxGC must be within [0, 1]");  // :81-82
    const std::string db = o.str("database");
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    // the process's first HIP call: runtime, device and code objects come up here, ~0.1 s on an MI355X box whatever the work is (timed on its own:
    // everything else the host does before the scan -- header, guide FASTA, site finding, encoding -- is ~10 ms for 1 000 guides)
    if (ffh_device_count() < 1) throw Error("no HIP device available (flashfry_hip has no CPU fallback)");
    const auto t0b = clk::now();
    std::fprintf(stderr, "Reading the header....\n");
    const HeaderInfo hdr = readHeaderInfo(db);  // :89
    const ParameterPack &pack = ParameterPack::indexToParameterPack(hdr.enzymeIndex);
    BitEncoding bitCoder(pack);
    BitPosition posCoder;
    for (const auto &c : hdr.contigs) posCoder.addReference(c);
    const std::vector<CRISPRSite> sites = findTargetSites(o.str("fasta"), pack, flank);  // :93
    std::fprintf(stderr, "Setting up the guide recording for our %zu candidate guides....\n", sites.size());
    std::vector<CRISPRSiteOT> guides;
    for (const auto &s : sites) {  // filter_by_GC :96 + new CRISPRSiteOT :100-102
        const double gc = gcContent(s.bases);
        if (!(gc >= minGC && gc <= maxGC)) continue;
        CRISPRSiteOT g;
        g.target = s;
        g.longEncoding = bitCoder.bitEncodeString(s.bases, 1);
        g.overflow = maxOT;
        guides.push_back(std::move(g));
    }
    std::fprintf(stderr, "Filtered GC guide count %zu\n", guides.size());
    // ResultsAggregator sorts the guides by start (ResultsAggregator.scala:35); ties keep input order here
    std::stable_sort(guides.begin(), guides.end(), [](const CRISPRSiteOT &a, const CRISPRSiteOT &b) { return a.target.position < b.target.position; });
    std::fprintf(stderr, "scanning against the known targets from the genome with %zu guides\n", guides.size());
    const bool positions = o.has("positionOutput");
    const auto t1 = clk::now();
    const ScanStats st = GpuTraverser::scan(db, guides, maxMismatch, maxOT, deviceList(o), positions, &hdr);  // replaces :120-131
    std::fprintf(stderr, "Performed a total of %llu guide to target comparisons (%llu targets resident on %d GPU(s); load %.1f ms, scan %.1f ms, finalize %.1f ms)\n",
                 (unsigned long long)st.executedComparisons, (unsigned long long)st.targets, st.gpus, st.loadMs, st.scanMs, st.finalizeMs);
    std::fprintf(stderr, "Database load: device set-up %.1f ms; header + member directory %.1f ms, inflate + copy %.1f ms (%u threads, %.1f MB -> %.1f MB), "
                         "block decode %.1f ms, scan images %.1f ms\n",
                 st.createMs, st.load.open_ms, st.load.inflate_ms, st.load.threads, st.load.compressed_bytes / 1e6, st.load.raw_bytes / 1e6, st.load.decode_ms,
                 st.load.prepare_ms);
    std::fprintf(stderr, "Writing final output for %zu guides\n", guides.size());
    const auto t2 = clk::now();
    TabDelimitedOutput out(o.str("output"), bitCoder, posCoder, {}, true, positions);  // :141-146
    for (auto &g : guides)
        for (auto &h : g.offTargets) h.hasCfd = false;  // discover writes no per-hit scores (scoring models = [])
    out.writeAll(guides);
    out.close();
    std::fprintf(stderr, "Host stages: HIP start-up %.1f ms, guide discovery %.1f ms, traverser %.1f ms (of which hit delivery %.1f ms), table output %.1f ms\n", ms(t0, t0b),
                 ms(t0b, t1), ms(t1, t2), st.deliverMs, ms(t2, clk::now()));
    return 0;
}

// ---- score: modules/ScoreResults.scala:90-154 -----------------------------------------------------------------------
int runScore(int argc, char **argv) {
    const Options o = parse(argc, argv, {"includeOTs", "numericOutput", "countOnTargetInScore"},
                            {"input", "output", "scoringMetrics", "maxMismatch", "database", "inputAnnotationBed", "shortestGuideEnergy", "transformPositions",
                             "maxReciprocalMismatch"});
    require(o, {"input", "output", "scoringMetrics", "database"});
    const HeaderInfo hdr = readHeaderInfo(o.str("database"));  // :91 (the database body is never opened)
    const ParameterPack &pack = ParameterPack::indexToParameterPack(hdr.enzymeIndex);
    BitEncoding bitEnc(pack);
    BitPosition posEnc;
    for (const auto &c : hdr.contigs) posEnc.addReference(c);
    const int maxMismatch = o.has("maxMismatch") ? o.num("maxMismatch", 0) : 0x7FFFFFFF;
    std::fprintf(stderr, "Loading CRISPR objects (filtering out overflow guides).. \n");
    std::vector<CRISPRSiteOT> guides = readTabDelimited(o.str("input"), bitEnc, posEnc, maxMismatch, true);  // :95
    std::vector<Metric> models;
    {
        std::string m = o.str("scoringMetrics");
        size_t a = 0;
        while (a <= m.size()) {
            size_t b = m.find(',', a);
            if (b == std::string::npos) b = m.size();
            const std::string name = m.substr(a, b - a);
            if (!name.empty()) {
                const Metric mt = metricByName(name);
                if (metricValidOverEnzyme(mt, pack)) models.push_back(mt);  // :111-118
                else std::fprintf(stderr, "DROPPING SCORING METHOD: %s; it's not valid over enzyme parameter pack: %s\n", name.c_str(), pack.name);
            }
            a = b + 1;
        }
    }
    // every hit-list model is computed by the same device epilogue as discover (no CPU scoring path)
    std::vector<uint64_t> longs(guides.size()), offsets(guides.size() + 1, 0), hitTargets;
    for (size_t g = 0; g < guides.size(); ++g) {
        longs[g] = guides[g].longEncoding;
        for (const auto &h : guides[g].offTargets) hitTargets.push_back(h.sequence);
        offsets[g + 1] = hitTargets.size();
    }
    ffh_ctx *ctx = ffh_create(0, hdr.enzymeIndex);
    if (!ctx) throw Error(ffh_last_error(nullptr));
    ffh_result *res = nullptr;
    if (ffh_score_lists(ctx, longs.data(), (uint32_t)guides.size(), offsets.data(), hitTargets.data(), &res)) {
        const std::string e = ffh_last_error(ctx);
        ffh_destroy(ctx);
        throw Error(e);
    }
    const bool wantCfd = std::find(models.begin(), models.end(), Metric::Doench2016CFD) != models.end();
    for (size_t g = 0; g < guides.size(); ++g) {
        guides[g].summary = ffh_result_summaries(res)[g];
        for (size_t k = 0; k < guides[g].offTargets.size(); ++k) {
            const double c = ffh_result_hit_cfd(res)[offsets[g] + k];
            guides[g].offTargets[k].hasCfd = wantCfd && c == c;  // Doench2016CFDScore attaches pam*cfd to every scored hit (:72)
            guides[g].offTargets[k].cfd = c;
        }
    }
    ffh_result_free(res);
    if (std::find(models.begin(), models.end(), Metric::Reciprocal) != models.end() && !guides.empty()) {
        // ReciprocalOffTargets.scoreGuides :54-62 is a guides x guides mismatch scan: the same device path with the guide
        // list resident as the "database" (file order = database order, the file index rides along as the position)
        const int maxRecip = o.num("maxReciprocalMismatch", 1);  // ScoreResults.scala:85-87
        std::vector<uint64_t> asTargets(guides.size()), index(guides.size());
        for (size_t g = 0; g < guides.size(); ++g) { asTargets[g] = (longs[g] & 0xFFFFFFFFFFFFULL) | (1ULL << 48); index[g] = g; }
        ffh_result *rr = nullptr;
        if (ffh_db_load_soa(ctx, asTargets.data(), asTargets.size(), index.data(), index.size(), 0) ||
            ffh_discover(ctx, longs.data(), (uint32_t)guides.size(), std::max(maxRecip, 0), 0x7FFFFFFF, 0, &rr)) {
            const std::string e = ffh_last_error(ctx);
            ffh_destroy(ctx);
            throw Error(e);
        }
        const uint64_t *go = ffh_result_guide_offsets(rr), *po = ffh_result_pos_offsets(rr), *pp = ffh_result_positions(rr);
        const uint8_t *mm = ffh_result_hit_mismatches(rr);
        for (size_t g = 0; g < guides.size(); ++g)
            for (uint64_t h = go[g]; h < go[g + 1]; ++h)
                if (mm[h] != 0) guides[g].reciprocal.push_back(guides[(size_t)pp[po[h]]].target.bases);  // :57-58
        ffh_result_free(rr);
    }
    ffh_destroy(ctx);
    std::stable_sort(guides.begin(), guides.end(), [](const CRISPRSiteOT &a, const CRISPRSiteOT &b) { return a.target.position < b.target.position; });  // :137
    TabDelimitedOutput out(o.str("output"), bitEnc, posEnc, models, o.has("includeOTs"), true, o.has("numericOutput"));  // :142-147
    out.writeAll(guides);
    out.close();
    return 0;
}

// ---- bulge: BASELINE.json config C5 -- Cas12a (TTTV) off-targets with <= N mismatches AND <= 1 one-base bulge --------------------
// No counterpart in the reference (it has no gap search): the alignment rules are DESIGN.md section 8 / csrc/ffh_bulge.hpp.  The table is
// this repository's own: one row per (guide, off-target), guides in FASTA order, off-targets in database order.
int runBulge(int argc, char **argv) {
    const Options o = parse(argc, argv, {"tttv"}, {"fasta", "database", "output", "maxMismatch", "maxBulge", "flankingSequence", "gpus", "devices"});
    require(o, {"fasta", "database", "output"});
    const int maxMismatch = o.num("maxMismatch", 3), maxBulge = o.num("maxBulge", 1);
    const std::string db = o.str("database");
    const HeaderInfo hdr = readHeaderInfo(db);
    if (hdr.enzymeIndex != 1) throw Error("the bulge search is specified for Cas12a / Cpf1 databases (enzyme cpf1) only");
    const ParameterPack &pack = ParameterPack::indexToParameterPack(hdr.enzymeIndex);
    BitEncoding bitCoder(pack);
    const std::vector<CRISPRSite> sites = findTargetSites(o.str("fasta"), pack, o.num("flankingSequence", 6));
    std::vector<uint64_t> longs;
    for (const auto &s : sites) longs.push_back(bitCoder.bitEncodeString(s.bases, 1));
    std::fprintf(stderr, "bulge search for %zu guides, <= %d mismatches, <= %d bulge%s\n", sites.size(), maxMismatch, maxBulge, o.has("tttv") ? ", TTTV sites only" : "");
    const std::vector<int> devs = deviceList(o);
    // bin shards on several GPUs concatenate in shard order = database order (no cut-off, no scores: nothing to exchange)
    std::vector<ffh_ctx *> ctx(devs.size(), nullptr);
    std::vector<ffh_bulge_result *> res(devs.size(), nullptr);
    const uint32_t nBins = (uint32_t)hdr.binBytes.size();
    FILE *out = std::fopen(o.str("output").c_str(), "w");
    if (!out) throw Error("cannot write " + o.str("output"));
    std::string err;
    for (size_t d = 0; d < devs.size() && err.empty(); ++d) {
        ctx[d] = ffh_create(devs[d], 0);
        if (!ctx[d]) { err = ffh_last_error(nullptr); break; }
        const uint32_t b0 = (uint32_t)((uint64_t)nBins * d / devs.size()), b1 = (uint32_t)((uint64_t)nBins * (d + 1) / devs.size());
        if (ffh_db_open(ctx[d], db.c_str(), b0, b1) ||
            ffh_discover_bulge(ctx[d], longs.data(), (uint32_t)longs.size(), maxMismatch, maxBulge, o.has("tttv") ? FFH_BULGE_PAM_TTTV : 0u, &res[d]))
            err = ffh_last_error(ctx[d]);
    }
    if (err.empty()) {
        std::fprintf(out, "guide\tguideSequence\toffTarget\tcount\tmismatches\tbulgeType\tbulgePosition\n");
        static const char *kind[3] = {"none", "RNA", "DNA"};
        for (size_t g = 0; g < sites.size(); ++g)
            for (size_t d = 0; d < devs.size(); ++d) {
                const uint64_t *go = ffh_bulge_result_guide_offsets(res[d]), *ht = ffh_bulge_result_hit_targets(res[d]);
                const uint8_t *mm = ffh_bulge_result_hit_mismatches(res[d]), *ty = ffh_bulge_result_hit_bulge_type(res[d]), *ps = ffh_bulge_result_hit_bulge_position(res[d]);
                for (uint64_t h = go[g]; h < go[g + 1]; ++h) {
                    const StringCount sc = bitCoder.bitDecodeString(ht[h]);
                    std::fprintf(out, "%s:%d\t%s\t%s\t%d\t%u\t%s\t%u\n", sites[g].contig.c_str(), sites[g].position, sites[g].bases.c_str(), sc.str.c_str(), sc.count,
                                 (unsigned)mm[h], kind[ty[h] < 3 ? ty[h] : 0], (unsigned)ps[h]);
                }
            }
    }
    std::fclose(out);
    for (size_t d = 0; d < devs.size(); ++d) { if (res[d]) ffh_bulge_result_free(res[d]); if (ctx[d]) ffh_destroy(ctx[d]); }
    if (!err.empty()) throw Error(err);
    return 0;
}

}  // namespace ffhost

static void usage() {
    std::fprintf(stderr,
                 "flashfry-hip <index|discover|score|bulge> [options]   (MI355X build of FlashFry's discover/score path)\n"
                 "  index    --reference FILE --database FILE [--enzyme spcas9ngg] [--binSize 7] [--tmpLocation DIR]\n"
                 "  discover --database FILE --fasta FILE --output FILE [--positionOutput] [--maxMismatch 4] [--flankingSequence 6]\n"
                 "           [--maximumOffTargets 2000] [--minGC 0] [--maxGC 1] [--forceLinear] [--gpus N]\n"
                 "  score    --input FILE --output FILE --database FILE\n"
                 "           --scoringMetrics hsu2013,doench2016cfd,minot,dangerous,jostandsantos,reciprocalofftargets\n"
                 "           [--maxMismatch N] [--includeOTs] [--numericOutput] [--maxReciprocalMismatch 1]\n"
                 "  bulge    --database FILE(cpf1) --fasta FILE --output FILE [--maxMismatch 3] [--maxBulge 1] [--tttv] [--gpus N]\n");
}

int main(int argc, char **argv) {
    if (argc < 2) { usage(); return 2; }
    const std::string cmd = argv[1];
    const auto t0 = std::chrono::steady_clock::now();
    try {
        int rc;
        if (cmd == "index") rc = ffhost::runIndex(argc - 2, argv + 2);
        else if (cmd == "discover") rc = ffhost::runDiscover(argc - 2, argv + 2);
        else if (cmd == "score") rc = ffhost::runScore(argc - 2, argv + 2);
        else if (cmd == "bulge") rc = ffhost::runBulge(argc - 2, argv + 2);
        else { usage(); return 2; }
        std::fprintf(stderr, "Total runtime %.2f seconds\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());  // Main.scala:63
        return rc;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "flashfry-hip %s: %s\n", cmd.c_str(), e.what());
        return 1;
    }
}
