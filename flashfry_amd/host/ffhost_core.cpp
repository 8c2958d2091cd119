// ffhost_core.cpp -- enzyme packs, bit codecs, guide discovery, Java double formatting, score columns.
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ffhost.hpp"

namespace ffhost {

// ---- standards/StandardScanParameters.scala:90-215 ---------------------------------------------------------
static const ParameterPack kPacks[6] = {
    {1, "CPF1", 24, 4, true, 0x00FFFFFFFFFFULL, 4, 24, false},
    {2, "SPCAS9", 23, 3, false, 0x3FFFFFFFFFC0ULL, 0, 20, true},
    {3, "SPCAS9NGG", 23, 3, false, 0x3FFFFFFFFFC0ULL, 0, 20, true},
    {4, "SPCAS9NAG", 23, 3, false, 0x3FFFFFFFFFC0ULL, 0, 20, true},
    {5, "SPCAS919", 22, 3, false, 0x0FFFFFFFFFC0ULL, 0, 19, false},
    {6, "SPCAS9NGG19", 22, 3, false, 0x0FFFFFFFFFC0ULL, 0, 19, false},
};

const ParameterPack &ParameterPack::indexToParameterPack(int index) {  // :61-69
    if (index < 1 || index > 6) throw Error("Unable to find the correct parameter pack for enzyme: " + std::to_string(index));
    return kPacks[index - 1];
}

const ParameterPack &ParameterPack::nameToParameterPack(const std::string &name) {  // :51-59
    std::string up = name;
    for (auto &c : up) c = (char)std::toupper((unsigned char)c);
    for (const auto &p : kPacks)
        if (up == p.name) return p;
    throw Error("Unable to find the correct parameter pack for enzyme: " + name);
}

static inline bool acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
static inline bool allAcgt(const char *s, int n) {
    for (int i = 0; i < n; ++i)
        if (!acgt(s[i])) return false;
    return true;
}

// fwdRegex / revRegex of each pack (:104-106, 126-128, 148-150, 170-172, 192-194, 209-211); the lookahead makes
// every position a candidate, so the regex is evaluated as a predicate at one position
bool ParameterPack::fwdMatch(const char *s, size_t rem) const {
    const int L = totalScanLength;
    if (rem < (size_t)L) return false;
    switch (index) {
        case 1: return s[0] == 'T' && s[1] == 'T' && s[2] == 'T' && allAcgt(s + 3, 21);
        case 2: case 5: return allAcgt(s, L - 2) && (s[L - 2] == 'A' || s[L - 2] == 'G') && s[L - 1] == 'G';
        case 3: case 6: return allAcgt(s, L - 2) && s[L - 2] == 'G' && s[L - 1] == 'G';
        case 4: return allAcgt(s, L - 2) && s[L - 2] == 'A' && s[L - 1] == 'G';
    }
    return false;
}
bool ParameterPack::revMatch(const char *s, size_t rem) const {
    const int L = totalScanLength;
    if (rem < (size_t)L) return false;
    switch (index) {
        case 1: return allAcgt(s, 21) && s[21] == 'A' && s[22] == 'A' && s[23] == 'A';
        case 2: case 5: return s[0] == 'C' && (s[1] == 'C' || s[1] == 'T') && allAcgt(s + 2, L - 2);
        case 3: case 6: return s[0] == 'C' && s[1] == 'C' && allAcgt(s + 2, L - 2);
        case 4: return s[0] == 'C' && s[1] == 'T' && allAcgt(s + 2, L - 2);
    }
    return false;
}

// ---- bitcoding/BitEncoding.scala ------------------------------------------------------------------------------
uint64_t BitEncoding::bitEncodeString(const std::string &str, int count) const {
    if (str.size() > 24) throw Error("String " + str + " is too long to be encoded (" + std::to_string(str.size()) + " > 24)");
    if (count < 1) throw Error("String count " + str + " - " + std::to_string(count) + " has a count <= 0");
    uint64_t enc = 0;
    for (char ch : str) {
        enc <<= 2;
        switch (std::toupper((unsigned char)ch)) {
            case 'A': break;
            case 'C': enc |= 1; break;
            case 'G': enc |= 2; break;
            case 'T': enc |= 3; break;
            default: throw Error(std::string("Unable to encode character ") + ch);
        }
    }
    return enc | ((uint64_t)(int64_t)count << 48);
}

StringCount BitEncoding::bitDecodeString(uint64_t enc, int actualSize) const {
    if (actualSize < 0) actualSize = mParameterPack.totalScanLength;
    std::string s((size_t)actualSize, 'A');
    for (int i = 0; i < actualSize; ++i) s[(size_t)(actualSize - 1 - i)] = "ACGT"[(enc >> (2 * i)) & 3];
    return {s, getCount(enc)};
}

int BitEncoding::mismatches(uint64_t e1, uint64_t e2, uint64_t additionalMask) const {
    const uint64_t first = (e1 ^ e2) & additionalMask & mParameterPack.comparisonBitEncoding;
    return __builtin_popcountll((first & upperBits) | ((first << 1) & upperBits));
}

// ---- bitcoding/BitPosition.scala ---------------------------------------------------------------------------------
void BitPosition::addReference(const std::string &refName) {
    indexToContig.push_back(refName);
    contigMap[refName] = (int)indexToContig.size();
    if (indexToContig.size() + 1 >= (0x000FFFFF00000000ULL >> 32)) throw Error("Contig count exceeds the current capacity of 1048575");
}
uint64_t BitPosition::encode(const std::string &refName, uint32_t position, int targetLength, bool forwardStrand) const {
    auto it = contigMap.find(refName);
    if (it == contigMap.end()) throw Error("Unknown contig: " + refName);
    if (targetLength >= 256) throw Error("Target length is too large, should be less than 128: " + std::to_string(targetLength));
    return ((uint64_t)it->second << 32) | (uint64_t)position | (forwardStrand ? 0ULL : (1ULL << 60)) | ((uint64_t)targetLength << 52);
}
PositionInformation BitPosition::decode(uint64_t e) const {
    const int id = (int)((e & 0x000FFFFF00000000ULL) >> 32);
    PositionInformation p;
    p.contig = (id >= 1 && (size_t)id <= indexToContig.size()) ? indexToContig[(size_t)id - 1] : std::string("?");
    p.start = (uint32_t)(e & 0xFFFFFFFFULL);
    p.length = (int)((e & 0x0FF0000000000000ULL) >> 52);
    p.forwardStrand = ((e & 0xF000000000000000ULL) >> 60) == 0;
    return p;
}

// ---- FASTA + SimpleSiteFinder (reference/ReferenceEncoder.scala:46-175) ---------------------------------------------
double gcContent(const std::string &s) {
    int gc = 0;
    for (char c : s) { c = (char)std::toupper((unsigned char)c); gc += (c == 'C' || c == 'G'); }
    return (double)gc / (double)s.size();
}

static std::string reverseComp(const char *s, size_t n) {  // utils/Utils.scala:81-88
    std::string o(n, 'N');
    for (size_t i = 0; i < n; ++i) {
        const char c = s[n - 1 - i];
        o[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
    }
    return o;
}

static void scanContig(const std::string &contig, const std::string &seq, const ParameterPack &p, int flank, std::vector<CRISPRSite> &out) {
    const int L = p.totalScanLength;
    const size_t n = seq.size();
    for (int pass = 0; pass < 2; ++pass)  // all forward matches of the contig first, then the reverse ones (:121-163)
        for (size_t i = 0; i + (size_t)L <= n; ++i) {
            if (!(pass == 0 ? p.fwdMatch(seq.data() + i, n - i) : p.revMatch(seq.data() + i, n - i))) continue;
            CRISPRSite s;
            s.contig = contig;
            s.forwardStrand = pass == 0;
            s.position = (int)i;
            const size_t cs = i >= (size_t)flank ? i - (size_t)flank : 0, ce = std::min(n, i + (size_t)L + (size_t)flank);
            s.hasContext = (ce - cs) == (size_t)(L + 2 * flank);  // context only when both flanks are complete (:131-134)
            if (pass == 0) {
                s.bases = seq.substr(i, (size_t)L);
                if (s.hasContext) s.sequenceContext = seq.substr(cs, ce - cs);
            } else {
                s.bases = reverseComp(seq.data() + i, (size_t)L);
                if (s.hasContext) s.sequenceContext = reverseComp(seq.data() + cs, ce - cs);
            }
            out.push_back(std::move(s));
        }
}

std::vector<CRISPRSite> findTargetSites(const std::string &fasta, const ParameterPack &pack, int flank, BitPosition *posEncoder) {
    gzFile f = gzopen(fasta.c_str(), "rb");  // reads plain text as well as .gz (fileToSource :76-82)
    if (!f) throw Error("cannot open " + fasta);
    std::vector<CRISPRSite> out;
    std::string contig, seq, line;
    bool have = false;
    std::vector<char> buf(1 << 16);
    auto flush = [&]() { if (have) scanContig(contig, seq, pack, flank, out); seq.clear(); };
    while (gzgets(f, buf.data(), (int)buf.size())) {
        line.assign(buf.data());
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        // lines longer than the buffer continue in the next gzgets call; FASTA headers are assumed to fit
        if (!line.empty() && line[0] == '>') {
            flush();
            contig = line.substr(1);
            for (auto &c : contig) if (c == ' ' || c == '\t') c = '_';  // :56
            if (posEncoder) posEncoder->addReference(contig);
            have = true;
        } else {
            for (char c : line) seq.push_back((char)std::toupper((unsigned char)c));  // :63
        }
    }
    gzclose(f);
    flush();
    return out;
}

// ---- java.lang.Double.toString ---------------------------------------------------------------------------------------
std::string javaDoubleToString(double d) {
    if (std::isnan(d)) return "NaN";
    if (std::isinf(d)) return d < 0 ? "-Infinity" : "Infinity";
    if (d == 0.0) return std::signbit(d) ? "-0.0" : "0.0";
    char buf[64];
    int prec = 1;
    for (; prec <= 17; ++prec) {  // shortest digit string that round-trips
        std::snprintf(buf, sizeof buf, "%.*e", prec - 1, d);
        if (std::strtod(buf, nullptr) == d) break;
    }
    std::string digits;
    const char *q = buf;
    const bool neg = *q == '-';
    if (neg) ++q;
    for (; *q && *q != 'e'; ++q)
        if (*q >= '0' && *q <= '9') digits.push_back(*q);
    const int exp10 = std::atoi(q + 1);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string o = neg ? "-" : "";
    const double a = std::fabs(d);
    if (a >= 1e-3 && a < 1e7) {
        if (exp10 >= 0) {
            for (int i = 0; i <= exp10; ++i) o.push_back(i < (int)digits.size() ? digits[(size_t)i] : '0');
            o.push_back('.');
            if ((int)digits.size() > exp10 + 1) o.append(digits, (size_t)exp10 + 1, std::string::npos);
            else o.push_back('0');
        } else {
            o += "0.";
            o.append((size_t)(-exp10 - 1), '0');
            o += digits;
        }
    } else {
        o.push_back(digits[0]);
        o.push_back('.');
        if (digits.size() > 1) o.append(digits, 1, std::string::npos);
        else o.push_back('0');
        o += "E" + std::to_string(exp10);
    }
    return o;
}

// ---- score columns -------------------------------------------------------------------------------------------------
Metric metricByName(const std::string &name) {
    std::string l = name;
    for (auto &c : l) c = (char)std::tolower((unsigned char)c);
    if (l == "hsu2013") return Metric::Hsu2013;
    if (l == "doench2016cfd") return Metric::Doench2016CFD;
    if (l == "minot") return Metric::MinOT;
    if (l == "dangerous") return Metric::Dangerous;
    if (l == "jostandsantos") return Metric::JostAndSantos;
    if (l == "reciprocalofftargets") return Metric::Reciprocal;
    throw Error("Unknown scoring metric: " + name +
                " (this build scores hit lists on the GPU: hsu2013, doench2016cfd, minot, dangerous, jostandsantos, reciprocalofftargets)");
}

bool metricValidOverEnzyme(Metric m, const ParameterPack &p) {
    if (m == Metric::JostAndSantos) return p.index != 1 && (p.totalScanLength == 23 || p.totalScanLength == 22);  // JostAndSantosCRISPRi.scala:53-58
    return (m == Metric::Hsu2013 || m == Metric::Doench2016CFD) ? p.cas9_23 : true;
}

std::vector<std::string> metricHeaderColumns(Metric m) {
    switch (m) {
        case Metric::Hsu2013: return {"Hsu2013"};
        case Metric::Doench2016CFD: return {"DoenchCFD_maxOT", "DoenchCFD_specificityscore"};
        case Metric::MinOT: return {"basesDiffToClosestHit", "closestHitCount", "0-1-2-3-4_mismatch"};
        case Metric::Dangerous: return {"dangerous_GC", "dangerous_polyT", "dangerous_in_genome"};
        case Metric::JostAndSantos: return {"JostCRISPRi_maxOT", "JostCRISPRi_specificityscore"};  // JostAndSantosCRISPRi.scala:132-134
        case Metric::Reciprocal: return {"ReciprocalOffTargets"};                                    // ReciprocalOffTargets.scala:98
    }
    return {};
}

std::vector<std::string> metricColumns(Metric m, const CRISPRSiteOT &g, const ParameterPack &p, bool numeric) {
    const ffh_guide_summary &s = g.summary;
    switch (m) {
        case Metric::Hsu2013:  // CrisprMitEduOffTarget.getScore :103-105
            return {javaDoubleToString((100.0 / (100.0 + s.hsu_sum)) * 100.0)};
        case Metric::Doench2016CFD: {  // Doench2016CFDScore.scoreGuide :76-87 (maxOT thresholded at 0.023, specificity not)
            const double spec = s.n_scored ? 1.0 / (1.0 + s.cfd_sum) : 1.0;
            return {s.cfd_max >= 0.023 ? javaDoubleToString(s.cfd_max) : std::string("0.0"), javaDoubleToString(spec)};
        }
        case Metric::MinOT: {  // ClosestHit.scoreGuide :71-75
            const std::string hist = std::to_string(s.hist[0]) + "," + std::to_string(s.hist[1]) + "," + std::to_string(s.hist[2]) + "," + std::to_string(s.hist[3]) +
                                     "," + std::to_string(s.hist[4]);
            if (s.closest == 0xFFFFFFFFu) return {"UNK", "0", hist};
            return {std::to_string(s.closest), std::to_string(s.closest_count), hist};
        }
        case Metric::Dangerous: {  // DangerousSequences.scoreGuide :49-68
            std::vector<std::string> prob(3, numeric ? "0" : "NONE");
            const double gc = gcContent(g.target.bases);
            if (numeric) prob[0] = javaDoubleToString(gc);
            else if (gc < .25 || gc > .75) prob[0] = "GC_" + javaDoubleToString(gc);
            if (g.target.bases.substr((size_t)p.guideLo, (size_t)(p.guideHi - p.guideLo)).find("TTTT") != std::string::npos) prob[1] = numeric ? "1" : "PolyT";
            if (!g.offTargets.empty() && s.in_genome > 0) prob[2] = numeric ? std::to_string(s.in_genome) : "IN_GENOME=" + std::to_string(s.in_genome);
            return prob;
        }
        case Metric::JostAndSantos:  // JostAndSantosCRISPRi.scoreGuide :42-45 ("0.0" when nothing was scored)
            return {javaDoubleToString(s.jost_max), javaDoubleToString(1.0 / (1.0 + s.jost_sum))};
        case Metric::Reciprocal: {   // ReciprocalOffTargets.scoreGuides :54-62; a guide without partners keeps the missing annotation (TabDelimitedHandler.scala:142)
            if (g.reciprocal.empty()) return {"NA"};
            std::string joined;
            for (const auto &b : g.reciprocal) joined += (joined.empty() ? "" : ",") + b;
            return {joined};
        }
    }
    return {};
}

}  // namespace ffhost
