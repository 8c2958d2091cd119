// ffh_inflate.hpp -- DEFLATE (RFC 1951) decoding of BGZF members ON THE DEVICE, one thread per member.
//
// A FlashFry database body is a series of independent gzip members of <= 64 KiB payload each (htsjdk
// BlockCompressedOutputStream; read side: BlockCompressedInputStream behind SeekTraverser.scala:113-120 /
// LinearTraverser.scala:122-130).  An hg38 database has ~80 000 of them: enough independent streams to fill the chip with
// one sequential decoder per thread, which takes the inflate off the host cores (16 usable CPUs inflate 6 GB/s).
//
//   k_inflate   thread = member: stored / fixed / dynamic blocks, canonical Huffman decoding with a 9-bit (literal/length)
//               and 6-bit (distance) first-level table per thread in global scratch, bit-serial (code counts in LDS) for longer codes
//   k_crc32     thread = member: CRC-32 of the produced payload (slicing-by-8, tables in LDS) against the member trailer
// Both report the first failing member through one atomicMin word.  Included by ffh_api.hip (single translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ffh {

struct InflateMember {
    uint64_t in_off;    // first byte of the deflate stream inside the compressed buffer
    uint64_t out_off;   // first byte of the payload inside the output buffer
    uint32_t in_len, out_len, crc, pad;
};

enum InflateError : uint32_t {
    kInfBadBlock = 1, kInfBadLengths = 2, kInfBadSymbol = 3, kInfBadDistance = 4, kInfOutput = 5, kInfInput = 6, kInfStored = 7, kInfSize = 8, kInfCrc = 9,
};

constexpr int kLitFastBits = 9, kDistFastBits = 6;
// per-thread work area, u16 entries
constexpr int kOffLitFast = 0;                                 // 512: (symbol << 4) | code length, 0 = longer than the table
constexpr int kOffLitSym = kOffLitFast + (1 << kLitFastBits);  // 288: symbols in canonical order
constexpr int kOffLitCnt = kOffLitSym + 288;                   // 16 : codes per length
constexpr int kOffDistFast = kOffLitCnt + 16;                  // 64
constexpr int kOffDistSym = kOffDistFast + (1 << kDistFastBits);  // 32
constexpr int kOffDistCnt = kOffDistSym + 32;                  // 16
constexpr int kOffLens = kOffDistCnt + 16;                     // 320: code lengths being read
constexpr int kOffWork = kOffLens + 320;                       // 16 : running offsets
constexpr int kOffClSym = kOffWork + 16;                       // 19 (+1): code-length alphabet
constexpr int kOffClCnt = kOffClSym + 20;                      // 16
constexpr int kInflateWorkU16 = kOffClCnt + 16;

// LSB-first bit buffer fed by aligned 32-bit words, ONE WORD AHEAD of need: the word OR-ed into the buffer was loaded a
// refill earlier, so the decoder never waits for the input stream (with byte-wise refills every symbol stalled on a load
// whose bits it did not need yet: 3 us per symbol).  Reads up to 8 bytes past the member (the caller pads the buffer);
// bits past the end are whatever follows -- a valid stream never uses them, and overrun() catches a corrupt one.
struct BitReader {
    const uint32_t *wp;  // next word to fetch
    uint64_t buf;
    int cnt;             // valid bits in buf
    uint32_t ahead;      // fetched, not yet in buf
    uint32_t bits_in;    // stream bits moved into buf so far
    uint32_t bits_len;   // length of the stream in bits
};

__device__ __forceinline__ BitReader br_open(const uint8_t *in, uint32_t len) {
    const uint32_t a = (uint32_t)((uintptr_t)in & 3u);
    const uint32_t *wp = reinterpret_cast<const uint32_t *>(in - a);
    BitReader b;
    b.buf = (uint64_t)(wp[0] >> (8 * a));
    b.cnt = (int)(32 - 8 * a);
    b.ahead = wp[1];
    b.wp = wp + 2;
    b.bits_in = (uint32_t)b.cnt;
    b.bits_len = len * 8;
    return b;
}
__device__ __forceinline__ void br_refill(BitReader &b) {  // afterwards cnt >= 33
    if (b.cnt <= 32) {
        b.buf |= (uint64_t)b.ahead << b.cnt;
        b.cnt += 32;
        b.bits_in += 32;
        b.ahead = *b.wp++;
    }
}
__device__ __forceinline__ uint32_t br_take(BitReader &b, int n) {  // n <= 16
    const uint32_t v = (uint32_t)b.buf & ((1u << n) - 1u);
    b.buf >>= n;
    b.cnt -= n;
    return v;
}
__device__ __forceinline__ uint32_t br_consumed(const BitReader &b) { return b.bits_in - (uint32_t)b.cnt; }
__device__ __forceinline__ bool br_overrun(const BitReader &b) { return b.cnt < 0 || br_consumed(b) > b.bits_len; }

// canonical Huffman tables from code lengths (RFC 1951 3.2.2); false when the lengths over-subscribe the code space
// `fast` may be addressed with a stride (a table interleaved across the threads of a block).
// `cnt` and `work` (16 entries each) are LDS arrays with stride kLdsStride as well: the bit-serial decoder reads cnt[] once
// per code bit, and from global scratch each of those reads was a full memory round trip.
constexpr int kLdsStride = 64;
__device__ inline bool huff_build(const uint16_t *lens, int n, uint16_t *cnt, uint16_t *sym, uint16_t *fast, int fast_bits, uint16_t *work, int fstride = 1) {
    constexpr int S = kLdsStride;
    for (int l = 0; l < 16; ++l) cnt[l * S] = 0;
    for (int s = 0; s < n; ++s) cnt[lens[s] * S] = (uint16_t)(cnt[lens[s] * S] + 1);
    int left = 1;
    for (int l = 1; l < 16; ++l) {
        left = (left << 1) - (int)cnt[l * S];
        if (left < 0) return false;
    }
    work[1 * S] = 0;
    for (int l = 1; l < 15; ++l) work[(l + 1) * S] = (uint16_t)(work[l * S] + cnt[l * S]);
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l) { sym[work[l * S]] = (uint16_t)s; work[l * S] = (uint16_t)(work[l * S] + 1); }
    }
    if (fast) {
        const uint32_t size = 1u << fast_bits;
        for (uint32_t i = 0; i < size; ++i) fast[i * fstride] = 0;
        uint32_t code = 0, idx = 0;
        for (int l = 1; l <= fast_bits; ++l) {
            for (uint32_t k = 0; k < cnt[l * S]; ++k) {
                const uint32_t rev = __brev(code) >> (32 - l);  // the stream carries Huffman codes most significant bit first
                const uint16_t e = (uint16_t)((sym[idx] << 4) | l);
                for (uint32_t j = rev; j < size; j += 1u << l) fast[j * fstride] = e;
                ++idx; ++code;
            }
            code <<= 1;
        }
    }
    return true;
}

// one symbol; -1 = no code matches
__device__ __forceinline__ int huff_decode(BitReader &b, const uint16_t *fast, int fast_bits, const uint16_t *cnt, const uint16_t *sym, int fstride = 1) {
    if (fast) {
        const uint32_t e = fast[((uint32_t)b.buf & ((1u << fast_bits) - 1u)) * fstride];
        if (e & 15u) { const int l = (int)(e & 15u); b.buf >>= l; b.cnt -= l; return (int)(e >> 4); }
    }
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
        code |= (int)(b.buf & 1u);
        b.buf >>= 1; b.cnt -= 1;
        const int count = cnt[l * kLdsStride];
        if (code - count < first) return sym[index + (code - first)];
        index += count; first += count;
        first <<= 1; code <<= 1;
    }
    return -1;
}

__device__ __forceinline__ void inflate_fail(unsigned long long *err, uint32_t member, uint32_t code) {
    atomicMin(err, ((unsigned long long)member << 8) | code);
}

// Output side.  On gfx9-class hardware one counter (vmcnt) tracks loads AND stores in issue order, so a table lookup cannot
// be consumed before every older store of the wave has been acknowledged: a byte store per literal put a full memory round
// trip into every symbol.  Literals are therefore gathered into aligned 8-byte words (one store per 8 symbols), and a match
// fetches its source bytes eight at a time before storing them.
struct OutWriter {
    uint8_t *out;
    uint32_t o;      // bytes produced
    uint32_t from;   // bytes [from, o) are still only in `acc`
    uint32_t mis;    // address of out[0] modulo 8: groups are aligned in memory, not in the member
    uint64_t acc;    // output byte i at bits 8 * ((i + mis) & 7) of its group
    __device__ __forceinline__ void literal(uint32_t s) {
        acc |= (uint64_t)s << (8 * ((o + mis) & 7u));
        ++o;
        if (((o + mis) & 7u) == 0) {
            if (o >= 8 && from == o - 8) *reinterpret_cast<uint64_t *>(out + from) = acc;  // the whole group: one aligned store
            else for (uint32_t i = from; i < o; ++i) out[i] = (uint8_t)(acc >> (8 * ((i + mis) & 7u)));
            acc = 0; from = o;
        }
    }
    __device__ __forceinline__ void flush() {  // before a match reads the output, and at the end
        for (uint32_t i = from; i < o; ++i) out[i] = (uint8_t)(acc >> (8 * ((i + mis) & 7u)));
        acc = 0; from = o;
    }
    __device__ __forceinline__ void match(uint32_t dist, uint32_t len) {
        flush();
        if (dist >= 8) {  // the 8 source bytes of a step never overlap its 8 destinations
            for (uint32_t k = 0; k < len; k += 8) {
                const uint32_t n = min(8u, len - k);
                uint8_t v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (uint32_t)i < n ? out[o - dist + i] : (uint8_t)0;
#pragma unroll
                for (int i = 0; i < 8; ++i) if ((uint32_t)i < n) out[o + i] = v[i];
                o += n;
            }
        } else {
            for (uint32_t k = 0; k < len; ++k, ++o) out[o] = out[o - dist];
        }
        from = o;
    }
};

// LANES lanes of every wave carry a member.  Measured on an hg38-scale body (82 700 members, 5.4 GB) with the first version
// (byte-wise input and output, everything in global scratch): 64 or 32 lanes 208 ms, 16 lanes 282 ms, 8 lanes 415 ms, 4 lanes
// 511 ms -- with all members resident at once the time is one member's serial chain, idle lanes buy nothing.  What shortened
// the chain: input words fetched one refill ahead (183 ms), literals stored as aligned 8-byte words and match sources fetched
// eight at a time (148 ms), the code counts of the bit-serial path in LDS (127 ms).  What did not: first-level tables in LDS
// (238 ms: 2.5 rounds of residency), an LDS ring of recent output for the match sources (145 ms).
template <int LANES>
__global__ __launch_bounds__(64) void k_inflate(const uint8_t *__restrict__ comp, const InflateMember *__restrict__ members, uint32_t first, uint32_t n,
                                                uint8_t *__restrict__ out_base, uint16_t *__restrict__ scratch, unsigned long long *__restrict__ err) {
    static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    // The first-level tables stay in the per-thread global scratch: in LDS (72 KB per block) only two blocks fit a CU, the
    // 82 700 members of an hg38 body then need 2.5 rounds and the kernel takes 238 ms instead of 148 ms -- every step of
    // a wave costs what its slowest lane costs (a match: stores acknowledged, sources loaded), not what the table lookup costs.
    __shared__ uint16_t lds_small[4 * 16 * 64];  // code counts of the three alphabets + the build's running offsets
    uint16_t *lit_cnt = lds_small + threadIdx.x, *dist_cnt = lit_cnt + 16 * 64, *cl_cnt = lit_cnt + 32 * 64, *work = lit_cnt + 48 * 64;
    if (threadIdx.x >= LANES) return;
    const uint32_t t = blockIdx.x * LANES + threadIdx.x;
    if (t >= n) return;
    const uint32_t mi = first + t;
    const InflateMember m = members[mi];
    uint16_t *W = scratch + (size_t)t * kInflateWorkU16;
    OutWriter w{out_base + m.out_off, 0u, 0u, (uint32_t)((uintptr_t)(out_base + m.out_off) & 7u), 0ull};
    BitReader b = br_open(comp + m.in_off, m.in_len);
    bool last = false;
    while (!last) {
        br_refill(b);
        last = br_take(b, 1) != 0;
        const uint32_t type = br_take(b, 2);
        if (type == 0) {  // stored: skip to the byte boundary, LEN, ~LEN, bytes
            br_take(b, (int)((8u - (br_consumed(b) & 7u)) & 7u));
            br_refill(b);
            const uint32_t len = br_take(b, 16);
            br_refill(b);
            const uint32_t nlen = br_take(b, 16);
            if (br_overrun(b) || (len ^ 0xFFFFu) != nlen) { inflate_fail(err, mi, kInfStored); return; }
            if (w.o + len > m.out_len) { inflate_fail(err, mi, kInfOutput); return; }
            for (uint32_t k = 0; k < len; ++k) {
                br_refill(b);
                w.literal(br_take(b, 8));
            }
            if (br_overrun(b)) { inflate_fail(err, mi, kInfInput); return; }
            continue;
        }
        if (type == 3) { inflate_fail(err, mi, kInfBadBlock); return; }
        uint16_t *lens = W + kOffLens;
        int nlit, ndist;
        if (type == 1) {  // fixed code, RFC 1951 3.2.6
            for (int s = 0; s < 144; ++s) lens[s] = 8;
            for (int s = 144; s < 256; ++s) lens[s] = 9;
            for (int s = 256; s < 280; ++s) lens[s] = 7;
            for (int s = 280; s < 288; ++s) lens[s] = 8;
            for (int s = 0; s < 30; ++s) lens[288 + s] = 5;
            nlit = 288; ndist = 30;
        } else {          // dynamic code, 3.2.7
            nlit = (int)br_take(b, 5) + 257;
            ndist = (int)br_take(b, 5) + 1;
            const int ncl = (int)br_take(b, 4) + 4;
            if (nlit > 286 || ndist > 30) { inflate_fail(err, mi, kInfBadLengths); return; }
            for (int i = 0; i < 19; ++i) lens[i] = 0;
            for (int i = 0; i < ncl; ++i) {
                br_refill(b);
                lens[kClOrder[i]] = (uint16_t)br_take(b, 3);
            }
            if (!huff_build(lens, 19, cl_cnt, W + kOffClSym, nullptr, 0, work)) { inflate_fail(err, mi, kInfBadLengths); return; }
            int i = 0;
            while (i < nlit + ndist) {
                br_refill(b);
                const int s = huff_decode(b, nullptr, 0, cl_cnt, W + kOffClSym);
                if (s < 0 || br_overrun(b)) { inflate_fail(err, mi, kInfBadLengths); return; }
                if (s < 16) { lens[i++] = (uint16_t)s; continue; }
                uint16_t val = 0;
                int rep;
                if (s == 16) {
                    if (i == 0) { inflate_fail(err, mi, kInfBadLengths); return; }
                    val = lens[i - 1];
                    rep = 3 + (int)br_take(b, 2);
                } else if (s == 17) rep = 3 + (int)br_take(b, 3);
                else rep = 11 + (int)br_take(b, 7);
                if (i + rep > nlit + ndist) { inflate_fail(err, mi, kInfBadLengths); return; }
                while (rep--) lens[i++] = val;
            }
            if (lens[256] == 0) { inflate_fail(err, mi, kInfBadLengths); return; }  // no end-of-block code
        }
        // the distance lengths follow the literal/length lengths; move them before the first build overwrites nothing (separate areas)
        if (!huff_build(lens, nlit, lit_cnt, W + kOffLitSym, W + kOffLitFast, kLitFastBits, work) ||
            !huff_build(lens + nlit, ndist, dist_cnt, W + kOffDistSym, W + kOffDistFast, kDistFastBits, work)) {
            // incomplete codes are legal (a single distance code); only over-subscription is refused
            inflate_fail(err, mi, kInfBadLengths);
            return;
        }
        for (;;) {
            br_refill(b);
            int s = huff_decode(b, W + kOffLitFast, kLitFastBits, lit_cnt, W + kOffLitSym);
            if (s < 0) { inflate_fail(err, mi, kInfBadSymbol); return; }
            if (s < 256) {
                if (w.o >= m.out_len) { inflate_fail(err, mi, kInfOutput); return; }
                w.literal((uint32_t)s);
                continue;
            }
            if (s == 256) break;
            s -= 257;
            if (s >= 29) { inflate_fail(err, mi, kInfBadSymbol); return; }
            const uint32_t len = kLenBase[s] + br_take(b, kLenExtra[s]);
            br_refill(b);  // up to 15 + 13 more bits for the distance
            const int ds = huff_decode(b, W + kOffDistFast, kDistFastBits, dist_cnt, W + kOffDistSym);
            if (ds < 0 || ds >= 30) { inflate_fail(err, mi, kInfBadDistance); return; }
            const uint32_t dist = kDistBase[ds] + br_take(b, kDistExtra[ds]);
            if (dist > w.o) { inflate_fail(err, mi, kInfBadDistance); return; }
            if (w.o + len > m.out_len) { inflate_fail(err, mi, kInfOutput); return; }
            w.match(dist, len);
        }
        if (br_overrun(b)) { inflate_fail(err, mi, kInfInput); return; }
    }
    w.flush();
    if (w.o != m.out_len) inflate_fail(err, mi, kInfSize);
}

// tables[k][v]: CRC-32 (reflected 0xEDB88320) of byte v followed by k zero bytes -- slicing-by-8
__global__ __launch_bounds__(64) void k_crc32(const uint8_t *__restrict__ out_base, const InflateMember *__restrict__ members, uint32_t n,
                                              const uint32_t *__restrict__ tables, unsigned long long *__restrict__ err) {
    __shared__ uint32_t T[8][256];
    for (uint32_t i = threadIdx.x; i < 8 * 256; i += blockDim.x) T[i >> 8][i & 255] = tables[i];
    __syncthreads();
    const uint32_t mi = blockIdx.x * blockDim.x + threadIdx.x;
    if (mi >= n) return;
    const InflateMember m = members[mi];
    const uint8_t *p = out_base + m.out_off;
    uint32_t len = m.out_len, crc = 0xFFFFFFFFu;
    while (len && ((uintptr_t)p & 7)) { crc = T[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8); --len; }
    for (; len >= 8; len -= 8, p += 8) {
        const uint64_t w = *reinterpret_cast<const uint64_t *>(p);
        const uint32_t lo = (uint32_t)w ^ crc, hi = (uint32_t)(w >> 32);
        crc = T[7][lo & 0xFF] ^ T[6][(lo >> 8) & 0xFF] ^ T[5][(lo >> 16) & 0xFF] ^ T[4][lo >> 24] ^ T[3][hi & 0xFF] ^ T[2][(hi >> 8) & 0xFF] ^ T[1][(hi >> 16) & 0xFF] ^
              T[0][hi >> 24];
    }
    while (len--) crc = T[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    if ((crc ^ 0xFFFFFFFFu) != m.crc) inflate_fail(err, mi, kInfCrc);
}

}  // namespace ffh
