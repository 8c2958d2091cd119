// Test-only driver over the PRODUCT's host layer (flashfry_amd/host/ffhost_table.cpp, ffhost_core.cpp) for the reference's own
// fixtures -- no oracle, no GPU:
//   roundtrip <in> <out>            TabDelimitedHanderTest.scala:40-51: TabDelimitedInput(file, enc, pos, 4, false) -> TabDelimitedOutput(no
//                                   score models, off-targets and positions written) -> the caller compares the bytes
//   sites <enzyme> <flank> <fasta>  SimpleSiteFinderTest.scala:13-173: one line "bases start fwd hasContext context" per site
//   encode <enzyme> <bases> <count> BitEncodingTest.scala: the long as decimal, then its decoded string and count
//   mismatches <enzyme> <a> <ca> <b> <cb>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <set>
#include <string>

#include "../flashfry_amd/host/ffhost.hpp"

using namespace ffhost;

int main(int argc, char **argv) {
    try {
        if (argc >= 4 && !std::strcmp(argv[1], "roundtrip")) {
            const ParameterPack &pack = ParameterPack::indexToParameterPack(2);   // Cas9ParameterPack, as the reference test
            BitEncoding enc(pack);
            BitPosition pos;
            {   // the contig table: the names of the file's positions in order of first appearance (the reference test registers hg19's)
                std::ifstream in(argv[2]);
                std::string line;
                std::set<std::string> seen;
                while (std::getline(in, line)) {
                    size_t at = 0;
                    while ((at = line.find_first_of("<|", at)) != std::string::npos) {
                        const size_t colon = line.find(':', at);
                        if (colon == std::string::npos) break;
                        const std::string name = line.substr(at + 1, colon - at - 1);
                        if (!name.empty() && name.find_first_of("\t,>") == std::string::npos && seen.insert(name).second) pos.addReference(name);
                        at = colon;
                    }
                }
            }
            const std::vector<CRISPRSiteOT> guides = readTabDelimited(argv[2], enc, pos, 4, false);
            TabDelimitedOutput out(argv[3], enc, pos, {}, true, true);
            for (const auto &g : guides) out.write(g);
            out.close();
            std::printf("%zu guides\n", guides.size());
            return 0;
        }
        if (argc >= 5 && !std::strcmp(argv[1], "sites")) {
            const ParameterPack &pack = ParameterPack::indexToParameterPack(std::atoi(argv[2]));
            for (const CRISPRSite &s : findTargetSites(argv[4], pack, std::atoi(argv[3])))
                std::printf("%s %d %d %d %s\n", s.bases.c_str(), s.position, s.forwardStrand ? 1 : 0, s.hasContext ? 1 : 0, s.hasContext ? s.sequenceContext.c_str() : "-");
            return 0;
        }
        if (argc >= 5 && !std::strcmp(argv[1], "encode")) {
            BitEncoding enc(ParameterPack::indexToParameterPack(std::atoi(argv[2])));
            const uint64_t v = enc.bitEncodeString(argv[3], std::atoi(argv[4]));
            const StringCount sc = enc.bitDecodeString(v, (int)std::strlen(argv[3]));
            std::printf("%llu %s %d\n", (unsigned long long)v, sc.str.c_str(), sc.count);
            return 0;
        }
        if (argc >= 7 && !std::strcmp(argv[1], "mismatches")) {
            BitEncoding enc(ParameterPack::indexToParameterPack(std::atoi(argv[2])));
            std::printf("%d\n", enc.mismatches(enc.bitEncodeString(argv[3], std::atoi(argv[4])), enc.bitEncodeString(argv[5], std::atoi(argv[6]))));
            return 0;
        }
        std::fprintf(stderr, "usage: host_golden roundtrip|sites|encode|mismatches ...\n");
        return 2;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
