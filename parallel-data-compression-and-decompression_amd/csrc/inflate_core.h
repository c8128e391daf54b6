// inflate_core.h -- chunk decoder behind decompress_chunk() (decompression.cpp:11-37): zlib
// header check, RFC 1951 block decoding, and the reference's "return codes ignored" semantics:
// output = everything zlib emits before it stops (stream end, input exhausted -- the reference
// truncates payloads at 65535 bytes, compression.cpp:127-132 -- or a data error); the Adler-32
// trailer never rejects (Inflate).
//
// Portable (host + device).  Decoding one Huffman stream is sequential, so the device gives a
// chunk to one wave: lane 0 runs the functions below and hands batches of decoded symbols to the
// whole wave, which does the byte moving (literal scatter, LZ copies, stored-block copies).
#pragma once
#include "zwz_common.h"

namespace zwz {

constexpr uint32_t kLitFastBits = 10, kDistFastBits = 8;
constexpr uint32_t kBatch = 64;

enum InflateStatus : uint32_t {
    kInfEnd = 0,        // final block decoded (trailer not checked here)
    kInfNeedInput = 1,  // payload ended early: truncated by the reference, or incomplete
    kInfDataError = 2,  // invalid stream
    kInfOverflow = 3,   // would produce more than the 65535-byte slot
    kInfRunning = 4,
    kInfPending = 5     // device only, between the two inflate kernels: a Huffman block was met, the chunk is still to be decoded
};

// Decoding tables of one block, built by lane 0 into LDS.
struct InflateTables {
    uint16_t lit_fast[1u << kLitFastBits];   // (sym << 4) | len, 0 = code longer than the index or invalid
    uint16_t dist_fast[1u << kDistFastBits];
    uint16_t lit_count[16], dist_count[16];  // canonical fallback: codes per length
    uint16_t lit_sym[288], dist_sym[32];     // symbols sorted by (length, symbol)
    uint16_t lit_walk[2], dist_walk[2];      // canonical walk resumed behind the fast table: (first, index) at length fast_bits + 1
};

// LSB-first bit reader over payload bytes [0, n).  `in` is either the payload itself (mask = ~0)
// or a power-of-two ring that holds a sliding window of it (mask = ring size - 1, the first 8 ring
// bytes mirrored past its end), indexed by absolute payload position.
struct BitReader {
    const uint8_t* in; uint32_t n, pos, mask;
    uint64_t hold; uint32_t bits;
    ZWZ_HD void init(const uint8_t* p, uint32_t len, uint32_t ring_mask = 0xffffffffu) {
        in = p; n = len; pos = 0; mask = ring_mask; hold = 0; bits = 0;
    }
    ZWZ_HD uint64_t load8(uint32_t at) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t o = at & mask;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(in) + (o >> 2);
        const uint32_t lo = __builtin_amdgcn_alignbyte(w[1], w[0], o & 3u), hi = __builtin_amdgcn_alignbyte(w[2], w[1], o & 3u);
        return ((uint64_t)hi << 32) | lo;
#else
        uint64_t v = 0;
        for (uint32_t i = 0; i < 8 && at + i < n; i++) v |= (uint64_t)in[(at + i) & mask] << (8 * i);
        return v;
#endif
    }
    // top up to >= 56 bits while input lasts: one 8-byte read instead of a byte loop
    ZWZ_HD void refill() {
        if (bits > 56 || pos >= n) return;
        uint32_t adv = (63u - bits) >> 3;            // whole bytes that fit
        const uint32_t avail = n - pos;
        uint64_t w = load8(pos);
        if (adv > avail) adv = avail;
        if (adv < 8) w &= (1ull << (8 * adv)) - 1ull;   // never expose bytes past the payload or the budget
        hold |= w << bits;
        pos += adv; bits += 8 * adv;
    }
    ZWZ_HD uint32_t peek(uint32_t k) const { return (uint32_t)(hold & ((1ull << k) - 1ull)); }
    ZWZ_HD void drop(uint32_t k) { hold >>= k; bits -= k; }
    ZWZ_HD bool take(uint32_t k, uint32_t& v) {  // all-or-nothing
        if (bits < k) { refill(); if (bits < k) return false; }
        v = peek(k); drop(k);
        return true;
    }
    // absolute bit position of the next unread bit / reposition at an absolute bit
    ZWZ_HD uint32_t bit_pos() const { return pos * 8u - bits; }
    ZWZ_HD void seek_bit(uint32_t bp) {
        pos = bp >> 3; hold = 0; bits = 0;
        refill();
        const uint32_t sk = bp & 7u;
        if (sk && bits >= sk) drop(sk);
    }
};

// Build fast + canonical tables from code lengths.  Returns 0 ok, -1 over-subscribed,
// 1 incomplete (caller decides whether that is legal).  max_len receives the longest code.
ZWZ_HD int build_decode_table(const uint8_t* lens, uint32_t n, uint16_t* fast, uint32_t fast_bits, uint16_t* count,
                              uint16_t* sorted, uint32_t& max_len, uint16_t* walk0 = nullptr) {
    uint16_t offs[16];
    for (uint32_t l = 0; l < 16; l++) count[l] = 0;
    for (uint32_t i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    max_len = 0;
    int left = 1;
    for (uint32_t l = 1; l < 16; l++) {
        if (count[l]) max_len = l;
        left <<= 1; left -= count[l];
        if (left < 0) return -1;
    }
    if (walk0) {   // decode_symbol's walk after fast_bits lengths without a hit: its `first` and `index` do not depend on the bits
        uint32_t first = 0, index = 0;
        for (uint32_t l = 1; l <= fast_bits; l++) { index += count[l]; first += count[l]; first <<= 1; }
        walk0[0] = (uint16_t)first; walk0[1] = (uint16_t)index;
    }
    offs[1] = 0;
    for (uint32_t l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (uint32_t i = 0; i < n; i++) if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
    for (uint32_t i = 0; i < (1u << fast_bits); i++) fast[i] = 0;
    // canonical codes in (length, symbol) order; entries replicated over the unused high bits
    uint32_t code = 0, idx = 0;
    for (uint32_t l = 1; l <= fast_bits; l++) {
        for (uint32_t k = 0; k < count[l]; k++, idx++, code++) {
            uint32_t rev = bit_reverse(code, l);
            uint16_t e = (uint16_t)((sorted[idx] << 4) | l);
            for (uint32_t j = rev; j < (1u << fast_bits); j += 1u << l) fast[j] = e;
        }
        code <<= 1;
    }
    return left > 0 ? 1 : 0;
}

// The device's fast tables in the form its window decode reads them (inflate_kernel; packed in place by the wave once a block's
// tables exist): an entry says at once what KIND of symbol starts at a bit offset and how many bits it SKIPS, so that a slot of the
// speculative window -- 512 of them a round, 7 % of which are symbols -- costs two lookups and no arithmetic on the symbol's value:
//   literal/length entry   kind << 13 | symbol << 4 | skip       skip = code length (+ the length code's extra bits), 1..15
//                          kind: 0 literal, 2 end of block, 5 a symbol that is no length code (286 / 287), 7 length code, 3 no entry
//   distance entry         kind << 13 | code << 5 | skip         skip = code length + extra bits, 1..21; kind: 1 fine, 5 code 30 / 31, 3 no entry
// (kinds are the window decode's own: kLit, kMatch, kEob, kSlow, kNeed, kErr.)
constexpr uint32_t kPkLit = 0, kPkMatch = 1, kPkEob = 2, kPkNone = 3, kPkErr = 5, kPkLen = 7;
ZWZ_HD uint16_t pack_lit_entry(uint32_t e /* classic: sym << 4 | len, 0 = none */) {
    if (e == 0) return (uint16_t)(kPkNone << 13);
    const uint32_t l = e & 15u, s = e >> 4;
    const uint32_t kind = s < 256u ? kPkLit : s == 256u ? kPkEob : s < 286u ? kPkLen : kPkErr;
    return (uint16_t)(kind << 13 | s << 4 | (l + (kind == kPkLen ? length_extra_bits(s - 257u) : 0u)));
}
ZWZ_HD uint16_t pack_dist_entry(uint32_t e) {
    if (e == 0) return (uint16_t)(kPkNone << 13);
    const uint32_t l = e & 15u, d = e >> 4;
    return (uint16_t)((d < 30u ? kPkMatch : kPkErr) << 13 | d << 5 | (l + (d < 30u ? dist_extra_bits(d) : 0u)));
}
// classic view of a packed entry (symbol << 4 | code length; 0 = none): what decode_symbol works on
ZWZ_HD uint32_t unpack_lit_entry(uint32_t e) {
    const uint32_t kind = e >> 13, s = (e >> 4) & 511u, skip = e & 15u;
    if (kind == kPkNone) return 0u;
    return s << 4 | (skip - (kind == kPkLen ? length_extra_bits(s - 257u) : 0u));
}
ZWZ_HD uint32_t unpack_dist_entry(uint32_t e) {
    const uint32_t kind = e >> 13, d = (e >> 5) & 31u, skip = e & 31u;
    if (kind == kPkNone) return 0u;
    return d << 4 | (skip - (kind == kPkMatch ? dist_extra_bits(d) : 0u));
}

// Decode one symbol.  Returns symbol >= 0, -1 if the payload ends inside the code (nothing
// consumed), -2 if the bits match no code.  kFmt: 0 = classic fast table, 1 = packed literal/length table, 2 = packed distance table.
template <uint32_t kFmt = 0>
ZWZ_HD int decode_symbol(BitReader& br, const uint16_t* fast, uint32_t fast_bits, const uint16_t* count,
                         const uint16_t* sorted) {
    if (br.bits < 15) br.refill();
    uint32_t e = fast[br.peek(fast_bits)];
    if (kFmt == 1u) e = unpack_lit_entry(e); else if (kFmt == 2u) e = unpack_dist_entry(e);
    uint32_t l = e & 15u;
    if (e != 0 && l <= br.bits) { br.drop(l); return (int)(e >> 4); }
    // slow path: canonical walk bit by bit (long codes, invalid codes, or the stream's last bits)
    int code = 0, first = 0, index = 0;
    for (uint32_t len = 1; len <= 15; len++) {
        if (br.bits < len) return -1;
        code |= (int)((br.hold >> (len - 1)) & 1u);
        int c = count[len];
        if (code - c < first) { br.drop(len); return sorted[index + (code - first)]; }
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -2;
}

struct InflateState {
    BitReader br;
    uint32_t out_pos;       // bytes produced so far
    uint32_t last;          // current block is final
    uint32_t status;        // InflateStatus
};

// zlib stream header (RFC 1950).  Returns false (status set) if decoding cannot start.
ZWZ_HD bool inflate_begin(InflateState& st, const uint8_t* in, uint32_t n, uint32_t ring_mask = 0xffffffffu) {
    st.br.init(in, n, ring_mask); st.out_pos = 0; st.last = 0; st.status = kInfRunning;
    uint32_t cmf, flg;
    if (!st.br.take(8, cmf) || !st.br.take(8, flg)) { st.status = kInfNeedInput; return false; }
    if (((cmf << 8) + flg) % 31u || (cmf & 15u) != 8u || (cmf >> 4) > 7u || (flg & 0x20u)) { st.status = kInfDataError; return false; }
    return true;
}

enum BlockKind : uint32_t { kBlkStored = 0, kBlkHuffman = 1, kBlkStop = 2 };

// Read one block header.  Stored: returns kBlkStored with (src_off, len) = bytes to copy from
// the payload (clipped to what the payload holds: a clipped stored block also sets status
// NeedInput).  Huffman: tables are built, returns kBlkHuffman.  kBlkStop: status says why.
// The header's first three bits (BFINAL, BTYPE); false (status set) if the payload ends inside them.
ZWZ_HD bool inflate_block_type(InflateState& st, uint32_t& type) {
    if (!st.br.take(1, st.last)) { st.status = kInfNeedInput; return false; }
    if (!st.br.take(2, type)) { st.status = kInfNeedInput; return false; }
    return true;
}

// A dynamic block's header in pieces (the device builds the three tables with the whole wave between them, see inflate_kernel;
// inflate_block_rest below strings them together for one thread):
//   inflate_dyn_begin    HLIT / HDIST / HCLEN and the code-length code's lengths -> cl[19]
//   inflate_dyn_lengths  the nlen + ndist code lengths (RLE symbols 16 / 17 / 18) through the code-length code's table, which
//                        sits in the distance-table slots -> lens[]
//   inflate_table_ok     zlib's verdict on a literal/length or distance code: over-subscribed never; incomplete only as a
//                        single one-bit code (literal/length) / at most one code (distance)
ZWZ_HD bool inflate_dyn_begin(InflateState& st, uint8_t* cl /* 19 */, uint32_t& nlen, uint32_t& ndist) {
    BitReader& br = st.br;
    uint32_t ncode, v;
    if (br.bits < 14) br.refill();
    if (br.bits < 14) { st.status = kInfNeedInput; return false; }
    br.take(5, nlen); br.take(5, ndist); br.take(4, ncode);
    nlen += 257; ndist += 1; ncode += 4;
    if (nlen > 286 || ndist > 30) { st.status = kInfDataError; return false; }
    for (uint32_t i = 0; i < 19; i++) cl[i] = 0;
    for (uint32_t i = 0; i < ncode; i++) {
        if (!br.take(3, v)) { st.status = kInfNeedInput; return false; }
        uint32_t o = i < 3 ? 16u + i : i == 3 ? 0u : (i & 1u) ? 8u - ((i - 3u) >> 1) : 8u + ((i - 4u) >> 1);
        cl[o] = (uint8_t)v;
    }
    return true;
}

ZWZ_HD bool inflate_dyn_lengths(InflateState& st, const InflateTables& t, uint8_t* lens, uint32_t nlen, uint32_t ndist) {
    BitReader& br = st.br;
    uint32_t have = 0;
    while (have < nlen + ndist) {
        BitReader save = br;
        int sym = decode_symbol(br, t.dist_fast, 7, t.dist_count, t.dist_sym);
        if (sym == -1) { st.status = kInfNeedInput; return false; }
        if (sym == -2) { st.status = kInfDataError; return false; }
        if (sym < 16) { lens[have++] = (uint8_t)sym; continue; }
        uint32_t prev = 0, rep, xb = sym == 16 ? 2u : sym == 17 ? 3u : 7u, xv;
        if (!br.take(xb, xv)) { br = save; st.status = kInfNeedInput; return false; }
        if (sym == 16) {
            if (have == 0) { st.status = kInfDataError; return false; }
            prev = lens[have - 1]; rep = 3u + xv;
        } else rep = (sym == 17 ? 3u : 11u) + xv;
        if (have + rep > nlen + ndist) { st.status = kInfDataError; return false; }
        while (rep--) lens[have++] = (uint8_t)prev;
    }
    if (lens[256] == 0) { st.status = kInfDataError; return false; }
    return true;
}

ZWZ_HD bool inflate_table_ok(InflateState& st, int rc, uint32_t max_len, bool literal) {
    if (rc < 0 || (rc > 0 && (literal ? max_len != 1 : max_len > 1))) { st.status = kInfDataError; return false; }
    return true;
}

ZWZ_HD uint32_t inflate_block_rest(InflateState& st, InflateTables* tp, uint8_t* lens, uint32_t v, uint32_t& src_off, uint32_t& len);

ZWZ_HD uint32_t inflate_block_header(InflateState& st, InflateTables& t, uint8_t* lens /* 320 B scratch */,
                                     uint32_t& src_off, uint32_t& len) {
    uint32_t v;
    if (!inflate_block_type(st, v)) return kBlkStop;
    return inflate_block_rest(st, &t, lens, v, src_off, len);
}

// Everything behind the type bits.  With tp == nullptr only stored blocks are handled: a Huffman block returns kBlkHuffman
// with nothing built (the caller hands the chunk to the full decoder).
ZWZ_HD uint32_t inflate_block_rest(InflateState& st, InflateTables* tp, uint8_t* lens, uint32_t v, uint32_t& src_off, uint32_t& len) {
    BitReader& br = st.br;
    if (v == 0) {
        br.drop(br.bits & 7u);
        uint32_t a, b;
        // LEN/NLEN: zlib needs all 32 bits before it copies anything
        if (br.bits + 8u * (br.n - br.pos) < 32u) { st.status = kInfNeedInput; return kBlkStop; }
        br.take(16, a); br.take(16, b);
        if ((a ^ 0xffffu) != b) { st.status = kInfDataError; return kBlkStop; }
        // give back whole bytes still in the bit buffer: the block body is byte-addressed
        br.pos -= br.bits >> 3; br.hold = 0; br.bits = 0;
        uint32_t avail = br.n - br.pos;
        src_off = br.pos;
        len = a < avail ? a : avail;
        br.pos += len;
        if (len < a) st.status = kInfNeedInput;
        return kBlkStored;
    }
    if (v == 3) { st.status = kInfDataError; return kBlkStop; }
    if (!tp) return kBlkHuffman;
    InflateTables& t = *tp;
    uint32_t nlen, ndist, max_len;
    if (v == 1) {
        nlen = 288; ndist = 30;
        for (uint32_t i = 0; i < 288; i++) lens[i] = (uint8_t)static_lit_len(i);
        for (uint32_t i = 0; i < 30; i++) lens[288 + i] = 5;
        build_decode_table(lens, 288, t.lit_fast, kLitFastBits, t.lit_count, t.lit_sym, max_len, t.lit_walk);
        build_decode_table(lens + 288, 30, t.dist_fast, kDistFastBits, t.dist_count, t.dist_sym, max_len, t.dist_walk);
        return kBlkHuffman;
    }
    uint8_t cl[19];
    if (!inflate_dyn_begin(st, cl, nlen, ndist)) return kBlkStop;
    // the code-length code reuses the distance-table slots (7-bit fast index fits in 8)
    if (build_decode_table(cl, 19, t.dist_fast, 7, t.dist_count, t.dist_sym, max_len) != 0) { st.status = kInfDataError; return kBlkStop; }
    if (!inflate_dyn_lengths(st, t, lens, nlen, ndist)) return kBlkStop;
    int lr = build_decode_table(lens, nlen, t.lit_fast, kLitFastBits, t.lit_count, t.lit_sym, max_len, t.lit_walk);
    if (!inflate_table_ok(st, lr, max_len, true)) return kBlkStop;
    int dr = build_decode_table(lens + nlen, ndist, t.dist_fast, kDistFastBits, t.dist_count, t.dist_sym, max_len, t.dist_walk);
    if (!inflate_table_ok(st, dr, max_len, false)) return kBlkStop;
    return kBlkHuffman;
}

// Decode up to kBatch symbols of the current Huffman block.  batch[i] = literal byte, or
// (len << 16) | dist for a match (dist >= 1).  pos[i] = output offset of symbol i.
// Returns the symbol count; *block_done is set at end-of-block or when decoding stops.
// out_cap bounds the output slot (65535): a symbol that would cross it stops with kInfOverflow.
// kPacked: t's two fast tables are in the device's packed form (see pack_lit_entry).
template <bool kPacked = false>
ZWZ_HD uint32_t inflate_decode_batch(InflateState& st, const InflateTables& t, uint32_t out_cap, uint32_t* batch,
                                     uint32_t* pos, bool& block_done, uint32_t max_syms = kBatch) {
    BitReader& br = st.br;
    uint32_t k = 0;
    block_done = false;
    while (k < max_syms) {
        int sym = decode_symbol<kPacked ? 1u : 0u>(br, t.lit_fast, kLitFastBits, t.lit_count, t.lit_sym);
        if (sym < 0) { st.status = sym == -1 ? kInfNeedInput : kInfDataError; block_done = true; break; }
        if (sym < 256) {
            if (st.out_pos >= out_cap) { st.status = kInfOverflow; block_done = true; break; }
            batch[k] = (uint32_t)sym; pos[k] = st.out_pos; k++; st.out_pos++;
            continue;
        }
        if (sym == 256) { block_done = true; break; }
        uint32_t c = (uint32_t)sym - 257u;
        if (c >= 29u) { st.status = kInfDataError; block_done = true; break; }
        uint32_t xv = 0, xb = length_extra_bits(c);
        if (!br.take(xb, xv)) { st.status = kInfNeedInput; block_done = true; break; }
        uint32_t len = (c == 28u ? 258u : length_base(c) + 3u + xv);
        int ds = decode_symbol<kPacked ? 2u : 0u>(br, t.dist_fast, kDistFastBits, t.dist_count, t.dist_sym);
        if (ds < 0) { st.status = ds == -1 ? kInfNeedInput : kInfDataError; block_done = true; break; }
        if (ds >= 30) { st.status = kInfDataError; block_done = true; break; }
        xb = dist_extra_bits((uint32_t)ds);
        if (!br.take(xb, xv)) { st.status = kInfNeedInput; block_done = true; break; }
        uint32_t dist = dist_base((uint32_t)ds) + 1u + xv;
        if (dist > st.out_pos) { st.status = kInfDataError; block_done = true; break; }  // too far back
        if (st.out_pos + len > out_cap) { st.status = kInfOverflow; block_done = true; break; }
        batch[k] = (len << 16) | dist; pos[k] = st.out_pos; k++; st.out_pos += len;
    }
    return k;
}

}  // namespace zwz
