// ffh_dbfile.hpp -- host-side reader of FlashFry's on-disk off-target database for the HIP library:
// text "<db>.header" (reference/binary/BinaryHeader.scala:69-160) + BGZF body written bin by bin
// (reference/binary/DatabaseWriter.scala:58-111; read side SeekTraverser.scala:113-120, LinearTraverser.scala:122-130).
// BGZF members are independent, so they are inflated on all host cores, straight into page-locked staging buffers
// whose copies to the device overlap the next inflate.  The bin payloads are decoded on the device (ffh_ingest.hpp).
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

// one BGZF member (gzip member with a 'BC' extra subfield, <= 64 KiB of payload)
struct Member {
    size_t coff, cdata_off, cdata_len;  // file offset of the member / of its deflate stream
    uint32_t isize, crc;
    uint64_t uoff;                      // offset of its payload in the uncompressed stream
};

struct BodyFile {  // the memory-mapped BGZF body and its member directory
    const uint8_t *data = nullptr;
    size_t size = 0;
    int fd = -1;
    std::vector<Member> members;
    uint64_t utotal = 0;
    BodyFile() = default;
    BodyFile(const BodyFile &) = delete;
    BodyFile &operator=(const BodyFile &) = delete;
    ~BodyFile();
};

struct IngestStats {
    unsigned threads = 0;
    uint64_t compressed_bytes = 0, raw_bytes = 0;
};

// hardware threads this process can really use (affinity mask, cgroup CPU quota, FFH_LOAD_THREADS override), <= 128
unsigned usable_cpus();

std::string open_body(const std::string &body_path, BodyFile &out);

// where the payloads of bins [bin_begin, bin_end) lie: the byte range [need_lo, need_hi) of the uncompressed stream that
// holds them, and per bin its offset (in longs, relative to need_lo) and length (in longs)
std::string locate_bins(const BodyFile &bf, const DbHeader &h, uint32_t bin_begin, uint32_t bin_end, uint64_t &need_lo, uint64_t &need_hi,
                        std::vector<uint64_t> &bin_off, std::vector<uint64_t> &bin_len);

// members [m0, m1) whose payload overlaps [need_lo, need_hi) of the uncompressed stream
void member_range(const BodyFile &bf, uint64_t need_lo, uint64_t need_hi, size_t &m0, size_t &m1);

// raw_copy == false: inflates [need_lo, need_hi) of the uncompressed stream on the host threads straight into device memory at d_raw.
// raw_copy == true : copies the file bytes of the overlapping members, still compressed, to d_raw (d_raw[0] = first byte of member m0);
//                    the device inflates them (ffh_inflate.hpp).  Same pipeline either way: page-locked slices, one stream per thread.
constexpr uint64_t kSmallBodyBytes = 256ull << 20;   // host inflate (FFH_INFLATE=host): bodies of up to this many INFLATED bytes are inflated by host threads and go over in one copy; device inflate: a COMPRESSED span up to this size goes over in one copy from the file mapping; beyond: the threaded page-locked pipeline
std::string inflate_to_device(const BodyFile &bf, uint64_t need_lo, uint64_t need_hi, uint8_t *d_raw, int device, IngestStats &stats, bool raw_copy,
                              bool force_pipeline = false /* FFH_LOAD_PIPELINE=1: the threaded copy pipeline whatever the size */);

}  // namespace ffh
