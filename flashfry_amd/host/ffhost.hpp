// ffhost.hpp -- C++ host layer above the C ABI (include/flashfry_hip.h).
//
// The reference's host side is Scala; no JVM exists in this environment, so the host side of this repository is
// C++ and mirrors the reference's interfaces for the discover/score path -- same names, same argument meaning,
// same error behaviour -- so that the CLI (`flashfry-hip index|discover|score`) is a drop-in for
// `java -jar flashfry.jar index|discover|score` on that path and the tests read like the reference's own:
//
//   ParameterPack            standards/StandardScanParameters.scala:31-215
//   BitEncoding              bitcoding/BitEncoding.scala:35-228
//   BitPosition              bitcoding/BitPosition.scala:36-92
//   CRISPRSite / SiteFinder  crispr/CRISPRSite.scala:34-53, reference/ReferenceEncoder.scala:46-175
//   CRISPRHit / CRISPRSiteOT crispr/CRISPRHit.scala:39-104, crispr/CRISPRSiteOT.scala:31-64
//   TabDelimitedOutput/Input targetio/TabDelimitedHandler.scala:38-335
//   GpuTraverser             the Traverser.scan plug-in point, reference/traverser/Traverser.scala:38-61
//   score columns            scoring/{Doench2016CFDScore,CrisprMitEduOffTarget,ClosestHit,DangerousSequences,
//                            JostAndSantosCRISPRi,ReciprocalOffTargets}.scala
//   DatabaseWriter / index   reference/binary/DatabaseWriter.scala:58-111, modules/BuildOffTargetDatabase.scala:57-89
//
// All comparisons, the cut-off and the scores are computed by the HIP library; this layer only moves text.
#pragma once
#include <stdint.h>

#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/flashfry_hip.h"

namespace ffhost {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---- standards/StandardScanParameters.scala -------------------------------------------------------------
struct ParameterPack {
    int index;               // parameterPackToIndex :72-80
    const char *name;        // nameToParameterPack :51-59 (upper-case)
    int totalScanLength;
    int pamLength;
    bool fivePrimePam;
    uint64_t comparisonBitEncoding;
    int guideLo, guideHi;    // guideRange
    bool cas9_23;            // Cas9Type && totalScanLength == 23: CFD and Hsu2013 are defined

    static const ParameterPack &indexToParameterPack(int index);
    static const ParameterPack &nameToParameterPack(const std::string &name);
    bool fwdMatch(const char *s, size_t remaining) const;  // fwdRegex at this position
    bool revMatch(const char *s, size_t remaining) const;  // revRegex at this position
};

// ---- bitcoding/BitEncoding.scala ---------------------------------------------------------------------------
struct StringCount {
    std::string str;
    int count;
};

class BitEncoding {
public:
    static constexpr uint64_t stringMask = 0xFFFFFFFFFFFFULL;
    static constexpr uint64_t upperBits = 0xAAAAAAAAAAAAULL;
    explicit BitEncoding(const ParameterPack &p) : mParameterPack(p) {}
    uint64_t bitEncodeString(const std::string &str, int count = 1) const;        // :46-67
    StringCount bitDecodeString(uint64_t encoding, int actualSize = -1) const;   // :85-99
    static int getCount(uint64_t encoding) { return (int)(int16_t)(encoding >> 48); }  // :114
    int mismatches(uint64_t e1, uint64_t e2, uint64_t additionalMask = stringMask) const;  // :127-132
    const ParameterPack &mParameterPack;
};

// ---- bitcoding/BitPosition.scala ----------------------------------------------------------------------------
struct PositionInformation {
    std::string contig;
    uint32_t start;
    int length;
    bool forwardStrand;
};

class BitPosition {
public:
    void addReference(const std::string &refName);                                            // :42-49
    uint64_t encode(const std::string &refName, uint32_t position, int targetLength, bool forwardStrand) const;  // :51-63
    PositionInformation decode(uint64_t encoding) const;                                      // :65-72
    const std::vector<std::string> &contigs() const { return indexToContig; }

private:
    std::map<std::string, int> contigMap;
    std::vector<std::string> indexToContig;  // id - 1
};

// ---- guide discovery ---------------------------------------------------------------------------------------
struct CRISPRSite {  // crispr/CRISPRSite.scala:34-53
    std::string contig, bases;
    bool forwardStrand = true;
    int position = 0;
    bool hasContext = false;
    std::string sequenceContext;
};

// ReferenceEncoder.findTargetSites :46-70 with SimpleSiteFinder :104-175 (plain or .gz FASTA)
std::vector<CRISPRSite> findTargetSites(const std::string &fasta, const ParameterPack &pack, int flankingSequence, BitPosition *posEncoder = nullptr);
double gcContent(const std::string &s);  // utils/Utils.scala:46

// ---- hits ---------------------------------------------------------------------------------------------------
struct CRISPRHit {  // crispr/CRISPRHit.scala:39-43
    uint64_t sequence = 0;
    std::vector<uint64_t> coordinates;     // the encoded positions; left empty when they are neither known nor printed
    uint32_t nCoordinates = 0;             // coordinates.size of the reference's object (an all-zero array there, crispr/CRISPRHit.scala:39-43)
    bool validOffTargetCoordinates = true;
    bool hasCfd = false;  // scores(Doench2016CFDScore), CRISPRHit.addScore
    double cfd = 0;
};

struct CRISPRSiteOT {  // crispr/CRISPRSiteOT.scala:31-64
    CRISPRSite target;
    uint64_t longEncoding = 0;
    int overflow = 0;
    bool inheritedOverflow = false;
    long currentTotal = 0;
    std::vector<CRISPRHit> offTargets;
    ffh_guide_summary summary{};                 // aggregates delivered by the HIP epilogue
    std::vector<std::string> reciprocal;         // namedAnnotations("ReciprocalOffTargets"): bases of the other guides within reach
    bool full() const { return currentTotal >= overflow; }
};

std::string javaDoubleToString(double d);  // java.lang.Double.toString

// ---- score columns (the reference's ScoreModel plug-ins that work on hit lists) ------------------------------
enum class Metric { Hsu2013, Doench2016CFD, MinOT, Dangerous, JostAndSantos, Reciprocal };
Metric metricByName(const std::string &name);                   // ScoreResults.getRegisteredScoringMetric :159-226
bool metricValidOverEnzyme(Metric m, const ParameterPack &p);   // ScoreModel.validOverEnzyme
std::vector<std::string> metricHeaderColumns(Metric m);         // ScoreModel.headerColumns
std::vector<std::string> metricColumns(Metric m, const CRISPRSiteOT &g, const ParameterPack &p, bool numericOutput);

// ---- targetio/TabDelimitedHandler.scala -----------------------------------------------------------------------
class TabDelimitedOutput {
public:
    TabDelimitedOutput(const std::string &outputFile, const BitEncoding &bitEncoding, const BitPosition &bitPosition, const std::vector<Metric> &models,
                       bool writeOTs, bool writePositions, bool numericOutput = false);  // :103-125
    ~TabDelimitedOutput();
    void write(const CRISPRSiteOT &guide);  // :131-153
    void writeAll(const std::vector<CRISPRSiteOT> &guides);  // the same rows, formatted on all usable CPUs
    void close();

private:
    void format(const CRISPRSiteOT &guide, std::string &row) const;
    struct Sink;
    Sink *out;
    const BitEncoding &enc;
    const BitPosition &pos;
    std::vector<Metric> models;
    bool writeOTs, writePositions, numeric;
};

// TabDelimitedInput :169-335 -- overflowed guides are dropped when filterOutOverflowedGuides (what `score` does)
std::vector<CRISPRSiteOT> readTabDelimited(const std::string &inputFile, const BitEncoding &bitEncoding, const BitPosition &bitPosition, int maximumMismatches,
                                           bool filterOutOverflowedGuides);

// ---- the Traverser plug-in: scan a database with the HIP library and fill the guides' hit lists -------------
struct ScanStats {
    uint64_t executedComparisons = 0;  // what the reference logs as Traverser.allComparisons (OffTargetDiscovery.scala:137)
    uint64_t targets = 0, positions = 0;
    double createMs = 0, loadMs = 0, scanMs = 0, finalizeMs = 0, deliverMs = 0;
    ffh_load_stats load{};  // stages of loadMs on the first shard
    int gpus = 1;
    int transport = 0;      // ffh_comm_transport of the shards' exchange: 0 copies (shards share a device), 1 RCCL
};
class GpuTraverser {
public:
    // devices: GPU ids; the bins are sharded contiguously over them (one host thread + one context per GPU)
    // header: what readHeaderInfo(binaryFile) returned to a caller that has read it already (nullptr: read here)
    static ScanStats scan(const std::string &binaryFile, std::vector<CRISPRSiteOT> &guides, int maxMismatch, int maximumOffTargets,
                          const std::vector<int> &devices, bool wantPositions, const struct HeaderInfo *header = nullptr);
};

// header of an on-disk database (enzyme + contig table): BinaryHeader.readHeader, reference/binary/BinaryHeader.scala:115-160
struct HeaderInfo {
    int enzymeIndex = 0;
    std::vector<std::string> contigs;
    std::vector<uint64_t> binBytes;
};
HeaderInfo readHeaderInfo(const std::string &databasePath);

// modules/BuildOffTargetDatabase.scala:57-89: FASTA streamed into the library's GPU indexer (ffh_indexer_*), written by ffh_db_write
void buildOffTargetDatabase(const std::string &reference, const std::string &output, const ParameterPack &pack, int binSize, int device = 0);

// ---- CLI modules ---------------------------------------------------------------------------------------------
int runIndex(int argc, char **argv);     // modules/BuildOffTargetDatabase.scala
int runDiscover(int argc, char **argv);  // modules/OffTargetDiscovery.scala
int runScore(int argc, char **argv);     // modules/ScoreResults.scala

}  // namespace ffhost
