// zwz_common.h -- constants and small tables shared by every stage of the chunk codec.
//
// The unit of work is the reference's Chunk (process.hpp:21-28): <= 65535 raw bytes, deflated
// independently with zlib level 6 (compression.cpp:119-134).  All parameters below are the ones
// deflateInit(level 6) fixes -- they are format-defining, not tunables.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZWZ_HD __host__ __device__ __forceinline__
#define ZWZ_D __device__ __forceinline__
#else
#define ZWZ_HD inline
#define ZWZ_D inline
#endif

namespace zwz {

constexpr uint32_t kChunk = 65535;        // process.hpp:12 CHUNK_SIZE
constexpr uint32_t kMaxDist = 32506;      // w_size - MIN_LOOKAHEAD
constexpr uint32_t kTooFar = 4096;
constexpr uint32_t kMinMatch = 3;
constexpr uint32_t kMaxMatch = 258;
constexpr uint32_t kGoodLen = 8;          // chain 128 -> 32 once prev_length >= 8
constexpr uint32_t kMaxLazy = 16;
constexpr uint32_t kNiceLen = 128;
constexpr uint32_t kMaxChain = 128;
constexpr uint32_t kShortChain = 32;
constexpr uint32_t kSlidePos = 65274;     // w_size + MAX_DIST: zlib slides its window here
constexpr uint32_t kWSize = 32768;
constexpr uint32_t kSymsPerBlock = 16383; // lit_bufsize - 1
constexpr uint32_t kMaxBlocks = 5;        // ceil(65535 / 16383) -> 4 non-final + 1 final
constexpr uint32_t kLCodes = 286, kDCodes = 30, kBLCodes = 19, kHeapSize = 2 * 286 + 1;
constexpr uint32_t kMaskWords = 1024;     // 65536 positions / 64

// Per-position match table entry: len | dist << 16, 0 = no match (len is 3..258).
ZWZ_HD uint32_t entry_pack(uint32_t len, uint32_t dist) { return len | (dist << 16); }
ZWZ_HD uint32_t entry_len(uint32_t e) { return e & 0xffffu; }
ZWZ_HD uint32_t entry_dist(uint32_t e) { return e >> 16; }

// 15-bit rolling hash of 3 bytes with shift 5 == closed form below.
ZWZ_HD uint32_t hash3(uint32_t b0, uint32_t b1, uint32_t b2) { return ((b0 << 10) ^ (b1 << 5) ^ b2) & 0x7fffu; }

// RFC 1951 length / distance code mapping, computed (no tables: cheap ALU beats LDS lookups).
// lc = len - 3 (0..255) -> length code 0..28, base, extra bits.
ZWZ_HD uint32_t length_code(uint32_t lc) {
    if (lc < 8) return lc;
    if (lc == 255) return 28;
    uint32_t hb = 31u - (uint32_t)__builtin_clz(lc);  // 3..7
    return ((hb - 1) << 2) + ((lc >> (hb - 2)) & 3u);
}
ZWZ_HD uint32_t length_extra_bits(uint32_t code) { return (code < 8 || code == 28) ? 0u : (code >> 2) - 1u; }
ZWZ_HD uint32_t length_base(uint32_t code) {  // in lc units (len - 3)
    if (code < 8) return code;
    if (code == 28) return 255;
    uint32_t xb = (code >> 2) - 1u;
    return ((4u + (code & 3u)) << xb);
}
// dm1 = dist - 1 (0..32767) -> distance code 0..29
ZWZ_HD uint32_t dist_code(uint32_t dm1) {
    if (dm1 < 4) return dm1;
    uint32_t hb = 31u - (uint32_t)__builtin_clz(dm1);  // >= 2
    return (hb << 1) + ((dm1 >> (hb - 1)) & 1u);
}
ZWZ_HD uint32_t dist_extra_bits(uint32_t code) { return code < 4 ? 0u : (code >> 1) - 1u; }
ZWZ_HD uint32_t dist_base(uint32_t code) {  // in dm1 units
    if (code < 4) return code;
    uint32_t xb = (code >> 1) - 1u;
    return (2u + (code & 1u)) << xb;
}

ZWZ_HD uint32_t static_lit_len(uint32_t sym) { return sym < 144 ? 8u : sym < 256 ? 9u : sym < 280 ? 7u : 8u; }
ZWZ_HD uint32_t bit_reverse(uint32_t v, uint32_t len) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v) >> (32u - len);
#else
    uint32_t r = 0;
    for (uint32_t i = 0; i < len; i++) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
#endif
}
// Static lit/len code (already bit-reversed for LSB-first emission).
ZWZ_HD uint32_t static_lit_code(uint32_t sym) {
    uint32_t c = sym < 144 ? 0x30u + sym : sym < 256 ? 0x190u + (sym - 144) : sym < 280 ? (sym - 256) : 0xC0u + (sym - 280);
    return bit_reverse(c, static_lit_len(sym));
}

}  // namespace zwz
