// ffh_dbfile.hpp -- host-side reader of FlashFry's on-disk off-target database for the HIP library:
// text "<db>.header" (reference/binary/BinaryHeader.scala:69-160) + BGZF body written bin by bin
// (reference/binary/DatabaseWriter.scala:58-111), and the decoder of bin payloads into structure-of-arrays
// (the walk of BlockManager.compareLinearBlock / compareIndexedBlock, blocks/BlockManager.scala:143-254,
// without the comparisons).  BGZF members are independent, so they are inflated on all host cores.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace ffh {

struct DbHeader {
    int enzyme_index = 0;
    int bin_width = 0;
    uint32_t n_bins = 0;
    std::vector<uint64_t> virtual_offset;     // per bin: BGZF virtual file pointer
    std::vector<uint64_t> uncompressed_bytes;  // per bin
    std::vector<uint32_t> n_targets;           // per bin (informational, quirk 17)
    std::vector<std::string> contigs;          // id = index + 1
};

// returns "" on success, else the error message
std::string read_db_header(const std::string &header_path, DbHeader &out);

// inflates the body and returns the payload longs of bins [bin_begin, bin_end) concatenated, plus offsets (in longs)
std::string read_db_bins(const std::string &body_path, const DbHeader &h, uint32_t bin_begin, uint32_t bin_end,
                         std::vector<int64_t> &longs, std::vector<uint64_t> &bin_offsets);

// decodes concatenated bin payloads into targets[] / positions[] (database order)
std::string decode_blocks(const int64_t *longs, const uint64_t *bin_offsets, uint32_t n_bins, std::vector<uint64_t> &targets,
                          std::vector<uint64_t> &positions);

}  // namespace ffh
