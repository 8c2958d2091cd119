// ffhost_index.cpp -- `index`: build the on-disk off-target database from a reference FASTA, in the reference's format
// (text <db>.header, BinaryHeader.scala:69-97, + BGZF body written bin by bin, DatabaseWriter.scala:58-111).
// The host only streams the FASTA (plain or .gz, ReferenceEncoder.scala:52-66) contig by contig into the library's
// indexer (include/flashfry_hip.h: ffh_indexer_*): site discovery, sort and duplicate merging run on the GPU, the blocks
// and the BGZF members are produced by ffh_db_write on the host cores.
#include <zlib.h>

#include <cstdio>
#include <cstring>

#include "ffhost.hpp"

namespace ffhost {

// calls fn(name, sequence) for every record; the sequence is the concatenation of the record's lines exactly as they are
// in the file (case is left alone: the device scan is case-insensitive, which is what line.toUpperCase :63 achieves)
static void forEachContig(const std::string &fasta, const std::function<void(const std::string &, const std::string &)> &fn) {
    gzFile f = gzopen(fasta.c_str(), "rb");  // reads plain text as well as .gz (fileToSource :76-82)
    if (!f) throw Error("cannot open " + fasta);
    gzbuffer(f, 1 << 20);
    std::vector<char> buf(8u << 20);
    std::string name, seq, header;
    bool have = false, in_header = false, at_line_start = true;
    auto finishHeader = [&]() {
        if (have) fn(name, seq);
        seq.clear();
        while (!header.empty() && header.back() == '\r') header.pop_back();
        name = header.substr(1);
        for (auto &c : name) if (c == ' ' || c == '\t') c = '_';  // :56
        have = true;
        header.clear();
        in_header = false;
    };
    for (;;) {
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) { gzclose(f); throw Error("read error on " + fasta); }
        if (n == 0) break;
        const char *p = buf.data(), *end = p + n;
        while (p < end) {
            if (at_line_start && !in_header && *p == '>') in_header = true;
            const char *nl = (const char *)std::memchr(p, '\n', (size_t)(end - p));
            const char *stop = nl ? nl : end;
            if (in_header) header.append(p, stop);
            else if (have) seq.append(p, stop);
            if (nl) {
                if (in_header) finishHeader();
                else if (have && !seq.empty() && seq.back() == '\r') seq.pop_back();  // getLines drops \r\n as well
                at_line_start = true;
                p = nl + 1;
            } else {
                at_line_start = false;
                p = end;
            }
        }
    }
    gzclose(f);
    if (in_header) finishHeader();  // header line without a trailing newline
    if (have) fn(name, seq);
}

void buildOffTargetDatabase(const std::string &reference, const std::string &output, const ParameterPack &pack, int binSize, int device) {
    if (binSize < 1 || binSize > 12) throw Error("binSize must be within 1..12");
    if (FILE *probe = std::fopen(reference.c_str(), "rb")) std::fclose(probe);
    else throw Error("cannot open " + reference);
    ffh_indexer *ix = ffh_indexer_create(device, pack.index);
    if (!ix) throw Error(ffh_indexer_last_error(nullptr));
    std::fprintf(stderr, "Discovering target sites in the input genome file...\n");
    try {
        forEachContig(reference, [&](const std::string &name, const std::string &seq) {
            std::fprintf(stderr, "Switching to chromosome >%s\n", name.c_str());  // ReferenceEncoder.scala:58
            if (ffh_indexer_add_contig(ix, name.c_str(), seq.data(), seq.size())) throw Error(ffh_indexer_last_error(ix));
        });
        std::fprintf(stderr, "Creating the final binary database file...\n");
        ffh_index_stats st;
        if (ffh_indexer_finish(ix, output.c_str(), binSize, &st)) throw Error(ffh_indexer_last_error(ix));
        std::fprintf(stderr, "Wrote %llu unique targets (%llu sites, %llu bases in %u contigs) into %u bins; site scan %.1f ms, sort + merge %.1f ms, write %.1f ms\n",
                     (unsigned long long)st.n_targets, (unsigned long long)st.n_sites, (unsigned long long)st.n_bases, st.n_contigs, 1u << (2 * binSize), st.scan_ms,
                     st.sort_ms, st.write_ms);
    } catch (...) {
        ffh_indexer_destroy(ix);
        throw;
    }
    ffh_indexer_destroy(ix);
}

}  // namespace ffhost
