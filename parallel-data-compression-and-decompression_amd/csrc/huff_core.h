// huff_core.h -- per-block Huffman stage of the chunk compressor (zlib's _tr_flush_block as it
// runs under consumer(), compression.cpp:131): exact emulation of zlib's heap-built trees
// (tie-break on depth), length limiting, code-length RLE, and the stored / static / dynamic
// decision.  SURVEY.md Appendix B "Block flush", "Tree build", "Length RLE".
//
// Portable (host + device).  Everything here is sequential per block; the device runs one
// block per wave with the scratch in LDS.
#pragma once
#include "zwz_common.h"

namespace zwz {

template <uint32_t kCapT>
struct TreeScratchT {
    // heap entries pack (freq:16 | depth:6 | node:10): smaller(n,m) == (key(n) <= key(m)) on the
    // (freq, depth) part, which is exactly zlib's comparison
    static constexpr uint32_t kCap = kCapT;      // zlib's HEAP_SIZE for this tree: 2 * elems + 1
    uint32_t heap[kCapT + 1];
    uint16_t freq[kCapT];
    uint16_t dad[kCapT];
    uint8_t depth[kCapT];
    uint8_t len[kCapT];
    uint16_t bl_count[16];
};
using TreeScratch = TreeScratchT<kHeapSize>;                 // any of the three trees
using BlTreeScratch = TreeScratchT<2 * kBLCodes + 1>;        // the code-length tree alone (the device's plan kernel: 0.4 KB of LDS)

struct BlockCodes {
    uint8_t llen[kLCodes + 2];    // lit/len code lengths (0 = unused)
    uint16_t lcode[kLCodes + 2];  // bit-reversed codes
    uint8_t dlen[kDCodes + 2];
    uint16_t dcode[kDCodes + 2];
    uint8_t bllen[kBLCodes];
    uint16_t blcode[kBLCodes];
    int32_t l_max_code, d_max_code, max_blindex;
    uint32_t opt_len, static_len;   // bits
};

ZWZ_HD uint32_t heap_key(uint32_t e) { return e >> 10; }

// pqdownheap.  Both sons are read at once and the chosen one is kept in hand (on the device: one LDS round trip a level, not
// two dependent ones).  heap[heap_len + 1] is read beside the last son but never used: the array has that element (zlib's
// sorted tail / the merge list start there).
ZWZ_HD void heap_sift(uint32_t* heap, int heap_len, int k) {
    const uint32_t v = heap[k], kv = heap_key(v);
    int j = k << 1;
    while (j <= heap_len) {
        const uint32_t a = heap[j], b = heap[j + 1];
        const bool right = j < heap_len && heap_key(b) <= heap_key(a);
        const uint32_t c = right ? b : a;
        if (kv <= heap_key(c)) break;
        heap[k] = c; k = j + (int)right; j = k << 1;
    }
    heap[k] = v;
}

// zlib build_tree + gen_bitlen + gen_codes.  freq_in[0..elems) are the symbol counts; lens/codes
// receive the result.  extra_bits(sym) / static_len_of(sym) feed opt_len / static_len.
// Returns max_code.  opt_len/static_len are updated in place (may be decremented for forced codes).
template <class Scratch, class ExtraFn, class StatFn>
ZWZ_HD int build_tree(Scratch& s, const uint16_t* freq_in, int elems, int max_length, ExtraFn extra_bits,
                      StatFn static_len_of, bool has_static, uint8_t* lens, uint16_t* codes, uint32_t& opt_len,
                      uint32_t& static_len) {
    constexpr int kCap = (int)Scratch::kCap;
    int heap_len = 0, heap_max = kCap, max_code = -1;
    for (int n = 0; n < elems; n++) {
        s.freq[n] = freq_in[n];
        s.len[n] = 0;
        if (freq_in[n] != 0) { max_code = n; s.depth[n] = 0; s.heap[++heap_len] = ((uint32_t)freq_in[n] << 16) | (uint32_t)n; }
    }
    while (heap_len < 2) {
        int node = max_code < 2 ? ++max_code : 0;
        s.freq[node] = 1; s.depth[node] = 0;
        s.heap[++heap_len] = (1u << 16) | (uint32_t)node;
        opt_len--;
        if (has_static) static_len -= static_len_of((uint32_t)node);
    }
    for (int n = heap_len / 2; n >= 1; n--) heap_sift(s.heap, heap_len, n);
    // the sorted tail shares the array with the heap (zlib's layout): heap[heap_max..kCap)
    int node = elems;
    do {
        uint32_t en = s.heap[1];
        s.heap[1] = s.heap[heap_len--];
        heap_sift(s.heap, heap_len, 1);
        uint32_t em = s.heap[1];
        uint32_t n = en & 1023u, m = em & 1023u;
        s.heap[--heap_max] = n; s.heap[--heap_max] = m;
        uint32_t f = (uint32_t)s.freq[n] + s.freq[m];
        uint32_t d = (s.depth[n] >= s.depth[m] ? s.depth[n] : s.depth[m]) + 1u;
        s.freq[node] = (uint16_t)f; s.depth[node] = (uint8_t)d;
        s.dad[n] = s.dad[m] = (uint16_t)node;
        s.heap[1] = (f << 16) | (d << 10) | (uint32_t)node;
        node++;
        heap_sift(s.heap, heap_len, 1);
    } while (heap_len >= 2);
    s.heap[--heap_max] = s.heap[1] & 1023u;

    // gen_bitlen
    for (int b = 0; b < 16; b++) s.bl_count[b] = 0;
    int overflow = 0, h;
    s.len[s.heap[heap_max]] = 0;
    for (h = heap_max + 1; h < kCap; h++) {
        uint32_t n = s.heap[h];
        int bits = s.len[s.dad[n]] + 1;
        if (bits > max_length) { bits = max_length; overflow++; }
        s.len[n] = (uint8_t)bits;
        if ((int)n > max_code) continue;
        s.bl_count[bits]++;
        uint32_t f = s.freq[n];
        opt_len += f * ((uint32_t)bits + extra_bits(n));
        if (has_static) static_len += f * (static_len_of(n) + extra_bits(n));
    }
    if (overflow > 0) {
        do {
            int bits = max_length - 1;
            while (s.bl_count[bits] == 0) bits--;
            s.bl_count[bits]--; s.bl_count[bits + 1] += 2; s.bl_count[max_length]--;
            overflow -= 2;
        } while (overflow > 0);
        for (int bits = max_length; bits != 0; bits--) {
            int n = s.bl_count[bits];
            while (n != 0) {
                uint32_t m = s.heap[--h];
                if ((int)m > max_code) continue;
                if (s.len[m] != (uint32_t)bits) {
                    opt_len += ((uint32_t)bits - s.len[m]) * s.freq[m];
                    s.len[m] = (uint8_t)bits;
                }
                n--;
            }
        }
    }
    // gen_codes
    uint32_t next_code[16], code = 0;
    next_code[0] = 0;
    for (int b = 1; b <= 15; b++) { code = (code + s.bl_count[b - 1]) << 1; next_code[b] = code; }
    for (int n = 0; n < elems; n++) {
        uint32_t l = n <= max_code ? s.len[n] : 0;
        lens[n] = (uint8_t)l;
        codes[n] = l ? (uint16_t)bit_reverse(next_code[l]++, l) : 0;
    }
    return max_code;
}

// ---------------------------------------------------------------------------------------------
// build_tree in pieces, for the device's plan stage (zwz_plan.hip): the heap is the only part of it that is a chain, so it
// runs a lane per tree with nothing but the heap in LDS; everything before and behind it is done by a wave per block.
//   heap_merge_all   zlib's heapify + merge loop over heap[1..heap_len] (entries freq << 16 | symbol, in symbol order:
//                    build_tree's first loop).  Merge s = 0, 1, ... joins the two least nodes n_s, m_s into node
//                    elems + s; on return heap[heap_len0 - s] = n_s | m_s << 16 (the slot the merge itself vacates) and
//                    heap[1] is the root.  n_0, m_0, n_1, m_1, ... is zlib's sorted tail read from the top
//                    (heap[HEAP_SIZE - 1] downwards), the order gen_bitlen's overflow repair walks.
//   tree_fix_overflow  gen_bitlen's repair of an over-long tree, on the lengths the merges gave.
//   rle_run          what scan_tree / send_tree make of ONE maximal run of equal lengths: runs are independent of each
//                    other (a new run always differs from the last length flushed), so a lane per run emits them.
ZWZ_HD void heap_merge_all(uint32_t* heap, uint32_t heap_len, uint32_t elems) {
    for (int n = (int)heap_len / 2; n >= 1; n--) heap_sift(heap, (int)heap_len, n);
    uint32_t node = elems;
    while (heap_len >= 2u) {
        const uint32_t en = heap[1];
        heap[1] = heap[heap_len--];
        heap_sift(heap, (int)heap_len, 1);
        const uint32_t em = heap[1];
        const uint32_t dn = (en >> 10) & 63u, dm = (em >> 10) & 63u;
        heap[heap_len + 1u] = (en & 1023u) | (em & 1023u) << 16;
        heap[1] = ((en >> 16) + (em >> 16)) << 16 | ((dn >= dm ? dn : dm) + 1u) << 10 | node++;
        heap_sift(heap, (int)heap_len, 1);
    }
}

// zlib's forced symbols: a tree with fewer than two used symbols gets symbol 0 / 1 (or 0 beside a lone larger one) with
// count 1.  m / max_code: used symbols and the highest of them (-1: none); returns the forced symbols in node[0..n).
ZWZ_HD uint32_t tree_forced_nodes(uint32_t& m, int& max_code, uint32_t node[2]) {
    uint32_t n = 0;
    while (m < 2u) { node[n++] = (uint32_t)(max_code < 2 ? ++max_code : 0); m++; }
    return n;
}

// gen_bitlen's overflow repair.  bl_count[1..max_length] as counted on the capped lengths, overflow = nodes (leaves and
// internal ones) deeper than max_length, pair(s) = n_s | m_s << 16, n_merges of them; rewrites len[] of the leaves.
template <class PairFn, class CountT>
ZWZ_HD void tree_fix_overflow(CountT* bl_count, int max_length, int overflow, PairFn pair, uint32_t n_merges, int max_code, uint8_t* len) {
    do {
        int bits = max_length - 1;
        while (bl_count[bits] == 0) bits--;
        bl_count[bits]--; bl_count[bits + 1] += 2; bl_count[max_length]--;
        overflow -= 2;
    } while (overflow > 0);
    uint32_t h = 0;                                     // index into n_0, m_0, n_1, m_1, ...
    for (int bits = max_length; bits != 0; bits--) {
        int n = (int)bl_count[bits];
        while (n != 0) {
            const uint32_t pr = pair(h >> 1), m = (h & 1u) ? pr >> 16 : pr & 0xffffu;
            h++;
            if ((int)m > max_code) continue;
            len[m] = (uint8_t)bits;
            n--;
        }
    }
    (void)n_merges;
}

// One maximal run of `c` equal code lengths `v` (c >= 1) as scan_tree / send_tree flush it; sink(sym, extra_val, extra_nbits).
template <class Sink>
ZWZ_HD void rle_run(uint32_t v, uint32_t c, Sink sink) {
    if (v == 0u) {
        for (; c >= 138u; c -= 138u) sink(18u, 127u, 7u);
        if (c == 0u) return;
        if (c < 3u) { do sink(0u, 0u, 0u); while (--c); }
        else if (c <= 10u) sink(17u, c - 3u, 3u);
        else sink(18u, c - 11u, 7u);
        return;
    }
    if (c < 4u) { do sink(v, 0u, 0u); while (--c); return; }
    sink(v, 0u, 0u);
    if (c <= 7u) { sink(16u, c - 4u, 2u); return; }
    sink(16u, 3u, 2u);
    for (c -= 7u; c >= 6u; c -= 6u) sink(16u, 3u, 2u);
    if (c == 0u) return;
    if (c < 3u) { do sink(v, 0u, 0u); while (--c); }
    else sink(16u, c - 3u, 2u);
}

// Code-length RLE over lens[0..max_code] (zlib scan_tree / send_tree).  sink(sym, extra_val,
// extra_nbits) is called per emitted bl symbol.
template <class Sink>
ZWZ_HD void rle_lengths(const uint8_t* lens, int max_code, Sink sink) {
    int prevlen = -1, curlen, nextlen = lens[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    for (int n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = n == max_code ? 0xffff : lens[n + 1];
        if (++count < max_count && curlen == nextlen) continue;
        if (count < min_count) {
            do { sink((uint32_t)curlen, 0u, 0u); } while (--count != 0);
        } else if (curlen != 0) {
            if (curlen != prevlen) { sink((uint32_t)curlen, 0u, 0u); count--; }
            sink(16u, (uint32_t)count - 3u, 2u);
        } else if (count <= 10) {
            sink(17u, (uint32_t)count - 3u, 3u);
        } else {
            sink(18u, (uint32_t)count - 11u, 7u);
        }
        count = 0; prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
}

ZWZ_HD uint32_t bl_order(uint32_t i) {
    // 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15
    return i < 3 ? 16u + i : i == 3 ? 0u : (i & 1u) ? 8u - ((i - 3u) >> 1) : 8u + ((i - 4u) >> 1);
}

enum BlockType : uint32_t { kStored = 0, kStatic = 1, kDynamic = 2 };

// Little bit sink over 32-bit words (LSB first), used for the dynamic block header.
struct BitSink {
    uint32_t* words; uint32_t nbits;
    ZWZ_HD void put(uint32_t v, uint32_t n) {
        if (n == 0) return;
        uint32_t w = nbits >> 5, o = nbits & 31u;
        if (o == 0) words[w] = v;
        else {
            words[w] |= v << o;
            if (o + n > 32) words[w + 1] = v >> (32u - o);
        }
        nbits += n;
    }
};

constexpr uint32_t kHdrWords = 96;  // dynamic header <= 14 + 57 + 316*(7+7) bits < 3072 bits

// Everything zlib decides when it flushes one block.  lfreq[286] (with lfreq[256] == 1) and
// dfreq[30] are the block's symbol histograms; stored_len the raw bytes it covers; stored_ok is
// false when zlib no longer has the block's bytes addressable (block began before a window slide).
// Outputs: codes, block type, the dynamic header bits (incl. the 3 block-type bits; `last` goes
// in bit 0), and the bit length of the block's symbol payload (incl. end-of-block) under the
// chosen code.
struct BlockPlan {
    uint32_t type;          // BlockType
    uint32_t hdr_bits;      // bits in hdr[] (3 for stored/static)
    uint32_t body_bits;     // symbols + EOB under the chosen tree (0 for stored)
};

ZWZ_HD BlockPlan plan_block(TreeScratch& ts, const uint16_t* lfreq, const uint16_t* dfreq, uint32_t stored_len,
                            bool stored_ok, uint32_t last, BlockCodes& bc, uint32_t* hdr /* kHdrWords */) {
    bc.opt_len = 0; bc.static_len = 0;
    bc.l_max_code = build_tree(ts, lfreq, (int)kLCodes, 15,
                               [](uint32_t n) { return n >= 257u ? length_extra_bits(n - 257u) : 0u; },
                               [](uint32_t n) { return static_lit_len(n); }, true, bc.llen, bc.lcode, bc.opt_len,
                               bc.static_len);
    bc.d_max_code = build_tree(ts, dfreq, (int)kDCodes, 15, [](uint32_t n) { return dist_extra_bits(n); },
                               [](uint32_t) { return 5u; }, true, bc.dlen, bc.dcode, bc.opt_len, bc.static_len);
    uint16_t blfreq[kBLCodes];
    for (uint32_t i = 0; i < kBLCodes; i++) blfreq[i] = 0;
    auto tally = [&](uint32_t sym, uint32_t, uint32_t) { blfreq[sym]++; };
    rle_lengths(bc.llen, bc.l_max_code, tally);
    rle_lengths(bc.dlen, bc.d_max_code, tally);
    uint32_t dummy_static = 0;
    build_tree(ts, blfreq, (int)kBLCodes, 7, [](uint32_t n) { return n < 16u ? 0u : n == 16u ? 2u : n == 17u ? 3u : 7u; },
               [](uint32_t) { return 0u; }, false, bc.bllen, bc.blcode, bc.opt_len, dummy_static);
    int mbi;
    for (mbi = (int)kBLCodes - 1; mbi >= 3; mbi--)
        if (bc.bllen[bl_order((uint32_t)mbi)] != 0) break;
    bc.max_blindex = mbi;
    bc.opt_len += 3u * ((uint32_t)mbi + 1u) + 5u + 5u + 4u;

    uint32_t opt_lenb = (bc.opt_len + 3u + 7u) >> 3, static_lenb = (bc.static_len + 3u + 7u) >> 3;
    if (static_lenb <= opt_lenb) opt_lenb = static_lenb;

    BlockPlan bp;
    BitSink sink{hdr, 0};
    if (stored_len + 4u <= opt_lenb && stored_ok) {
        bp.type = kStored;
        sink.put((0u << 1) + last, 3);
        bp.body_bits = 0;
    } else if (static_lenb == opt_lenb) {
        bp.type = kStatic;
        sink.put((1u << 1) + last, 3);
        bp.body_bits = bc.static_len;
        // re-express the codes as the static ones so the encoder needs no special case
        for (uint32_t n = 0; n < kLCodes; n++) { bc.llen[n] = (uint8_t)static_lit_len(n); bc.lcode[n] = (uint16_t)static_lit_code(n); }
        for (uint32_t n = 0; n < kDCodes; n++) { bc.dlen[n] = 5; bc.dcode[n] = (uint16_t)bit_reverse(n, 5); }
    } else {
        bp.type = kDynamic;
        sink.put((2u << 1) + last, 3);
        sink.put((uint32_t)(bc.l_max_code + 1 - 257), 5);
        sink.put((uint32_t)(bc.d_max_code + 1 - 1), 5);
        sink.put((uint32_t)(mbi + 1 - 4), 4);
        for (int r = 0; r <= mbi; r++) sink.put(bc.bllen[bl_order((uint32_t)r)], 3);
        auto send = [&](uint32_t sym, uint32_t xv, uint32_t xn) { sink.put(bc.blcode[sym], bc.bllen[sym]); sink.put(xv, xn); };
        rle_lengths(bc.llen, bc.l_max_code, send);
        rle_lengths(bc.dlen, bc.d_max_code, send);
        // opt_len counts header + symbols + EOB; the symbol part is what remains
        bp.body_bits = bc.opt_len - (sink.nbits - 3u);
    }
    bp.hdr_bits = sink.nbits;
    return bp;
}

// ---------------------------------------------------------------------------------------------
// Stored-block shortcut.  zlib stores a block iff stored_len + 4 <= min(opt_lenb, static_lenb).
// static_len is a plain weighted sum; opt_len needs zlib's exact trees -- unless a LOWER bound on
// it already clears the threshold.  Every prefix code costs at least the optimal Huffman cost
// (tie-break independent: the sum of the internal node weights), so
//     opt_len >= huff(lit) + huff(dist) + extra bits + 14 + 3*4 + ceil(used_codes / 6)
// (at least 4 code-length-code lengths are sent; one code-length symbol of >= 1 bit describes at
// most 6 non-zero lengths).  Incompressible data clears the threshold by ~100 bits per block, so
// the heap-built trees are only constructed for blocks that may really be Huffman coded.

// Optimal prefix-code cost of ascending weights sorted[0..m) (two-queue merge; queue: m + 2 words,
// sorted: m + 2 entries readable).  The two smallest live items are always among the next two
// leaves and the next two internal nodes, so each step reads those four at once (one LDS round
// trip on the device) and decides in registers.
ZWZ_HD uint32_t huffman_cost_sorted(const uint16_t* sorted, uint32_t m, uint32_t* queue) {
    if (m < 2) return 0;
    constexpr uint32_t kInf = 0xffffffffu;
    uint32_t li = 0, qi = 0, qn = 0, cost = 0;
    for (uint32_t step = 1; step < m; step++) {
        const uint32_t l0 = li < m ? sorted[li] : kInf, l1 = li + 1 < m ? sorted[li + 1] : kInf;
        const uint32_t q0 = qi < qn ? queue[qi] : kInf, q1 = qi + 1 < qn ? queue[qi + 1] : kInf;
        uint32_t sum;
        if (l0 <= q0) {
            if (l1 <= q0) { sum = l0 + l1; li += 2; } else { sum = l0 + q0; li++; qi++; }
        } else {
            if (l0 <= q1) { sum = q0 + l0; li++; qi++; } else { sum = q0 + q1; qi += 2; }
        }
        cost += sum;
        queue[qn++] = sum;
    }
    return cost;
}

struct StoredProbe { uint32_t static_len, extra_bits, used; };

// Exact static_len, exact extra bits and the number of used codes of one block's histograms.
ZWZ_HD StoredProbe probe_block(const uint16_t* lfreq, const uint16_t* dfreq) {
    StoredProbe r{0, 0, 0};
    uint32_t dused = 0;
    for (uint32_t n = 0; n < kLCodes; n++) {
        const uint32_t f = lfreq[n];
        if (!f) continue;
        const uint32_t x = n >= 257u ? length_extra_bits(n - 257u) : 0u;
        r.static_len += f * (static_lit_len(n) + x); r.extra_bits += f * x; r.used++;
    }
    for (uint32_t n = 0; n < kDCodes; n++) {
        const uint32_t f = dfreq[n];
        if (!f) continue;
        r.static_len += f * (5u + dist_extra_bits(n)); r.extra_bits += f * dist_extra_bits(n); dused++;
    }
    r.used += dused;
    return r;
}

// What the lower bound settles without building zlib's trees (_tr_flush_block's decision):
//   stored  iff stored_ok and stored_len + 4 <= min(opt_lenb, static_lenb)
//   static  iff not stored and static_lenb <= opt_lenb
// With opt_lbb <= opt_lenb: "stored" is certain when stored_len + 4 clears min(opt_lbb, static_lenb);
// "static" is certain when static_lenb <= opt_lbb and stored is ruled out (tiny blocks: the 14+12 bit
// dynamic header alone outweighs them).  Returns kShortNone when the exact trees are needed.
enum : uint32_t { kShortNone = 0, kShortStored = 1, kShortStatic = 2 };
ZWZ_HD uint32_t shortcut_type(const StoredProbe& pr, uint32_t huff_lit, uint32_t huff_dist, uint32_t stored_len, bool stored_ok) {
    const uint32_t opt_lb = huff_lit + huff_dist + pr.extra_bits + 14u + 12u + (pr.used + 5u) / 6u;
    const uint32_t opt_lbb = (opt_lb + 10u) >> 3, static_lenb = (pr.static_len + 10u) >> 3;
    if (stored_ok && stored_len + 4u <= opt_lbb && stored_len + 4u <= static_lenb) return kShortStored;
    if (static_lenb <= opt_lbb) return kShortStatic;     // min(opt_lenb, static_lenb) = static_lenb, and stored_len + 4 > static_lenb or stored not allowed
    return kShortNone;
}

// Bits of one symbol under a block's codes, LSB-first.  Literal: entry == 0, lit = byte.
// Match: entry = packed (len, dist).  At most 15+5+15+13 = 48 bits.
ZWZ_HD void symbol_bits(const uint16_t* lcode, const uint8_t* llen, const uint16_t* dcode, const uint8_t* dlen,
                        uint32_t entry, uint32_t lit, uint64_t& bits, uint32_t& nbits) {
    if (entry == 0) { bits = lcode[lit]; nbits = llen[lit]; return; }
    uint32_t lc = entry_len(entry) - kMinMatch, dm1 = entry_dist(entry) - 1u;
    uint32_t c = length_code(lc), d = dist_code(dm1);
    uint64_t v = lcode[257u + c];
    uint32_t n = llen[257u + c];
    v |= (uint64_t)(lc - length_base(c)) << n; n += length_extra_bits(c);
    v |= (uint64_t)dcode[d] << n; n += dlen[d];
    v |= (uint64_t)(dm1 - dist_base(d)) << n; n += dist_extra_bits(d);
    bits = v; nbits = n;
}

// Code length only (first pass of the encoder's two-pass scan).
ZWZ_HD uint32_t symbol_nbits(const uint8_t* llen, const uint8_t* dlen, uint32_t entry, uint32_t lit) {
    if (entry == 0) return llen[lit];
    const uint32_t c = length_code(entry_len(entry) - kMinMatch), d = dist_code(entry_dist(entry) - 1u);
    return llen[257u + c] + length_extra_bits(c) + dlen[d] + dist_extra_bits(d);
}

}  // namespace zwz
