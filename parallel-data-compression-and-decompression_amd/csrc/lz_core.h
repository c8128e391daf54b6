// lz_core.h -- LZ77 stage of the chunk compressor, as position-parallel tables + a table walk.
//
// Replaces zlib's deflate_slow()/longest_match() as called from consumer()
// (compression.cpp:119-131) with a formulation that is parallel over positions
// (SURVEY.md section 7.3): every position is inserted in the hash chains before it is searched, so
// the candidate list of position p (earlier positions with the same 15-bit hash, newest first)
// does not depend on parse decisions.  The outcome of longest_match(p, prev_length = b) is
// "best record among the first C candidates if its length > b", C = 32 if b >= 8 else 128, with
// an early stop at the first length >= min(128, lookahead).  So two (len, dist) records per
// position -- after 32 and after 128 candidates -- determine the whole lazy parse.
//
// Portable (host + device): the CPU build of this header is exercised against the oracle by
// tests/ so that the GPU kernels only add the parallel glue.
#pragma once
#include "zwz_common.h"

namespace zwz {

// 4 bytes at an arbitrary byte offset of a 4-byte-aligned buffer (LDS on the device).
ZWZ_HD uint32_t load_u32(const uint8_t* base, uint32_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t* w = reinterpret_cast<const uint32_t*>(base) + (off >> 2);
    return __builtin_amdgcn_alignbyte(w[1], w[0], off & 3u);
#else
    return (uint32_t)base[off] | (uint32_t)base[off + 1] << 8 | (uint32_t)base[off + 2] << 16 | (uint32_t)base[off + 3] << 24;
#endif
}

// Common prefix length of data[a..] and data[b..] given the first k0 bytes are known equal,
// capped at max_len.  Reads up to 7 bytes past a + max_len / b + max_len.
ZWZ_HD uint32_t match_len_from(const uint8_t* data, uint32_t a, uint32_t b, uint32_t k0, uint32_t max_len) {
    uint32_t k = k0;
    while (k < max_len) {
        uint32_t x = load_u32(data, a + k) ^ load_u32(data, b + k);
        if (x) {
            k += (uint32_t)__builtin_ctz(x) >> 3;
            break;
        }
        k += 4;
    }
    return k < max_len ? k : max_len;
}

// Match records of position p.  data/link are window views: data[i - org], link[i - org] hold
// byte i / chain predecessor of position i (0 = NIL).  L = chunk length.
//
// Candidate filter: a candidate can only beat `best` if it agrees with the scan on bytes
// 0..best; the loop tests the three bytes best-2..best on one word (for best == 2 that is the
// trigram itself, afterwards it is zlib's scan_end test, one byte stronger).  The link of the
// next candidate is read together with the filter word, so a rejected candidate costs one LDS
// round trip, and every stop condition is folded into the loop predicate (a version with early
// exits spent ~35 exec-mask SALU instructions per candidate).
ZWZ_HD void lz_search(const uint8_t* data, const uint16_t* link, uint32_t org, uint32_t p, uint32_t L,
                      uint32_t& e128, uint32_t& e32) {
    e128 = 0; e32 = 0;
    if (p + kMinMatch > L) return;                       // lookahead < 3: not inserted, not searched
    uint32_t cur = link[p - org];
    if (cur == 0 || p - cur > kMaxDist) return;          // first candidate: distance <= MAX_DIST
    if (p >= kSlidePos && cur <= kWSize) return;         // zlib's window has slid: <= 32768 reads as NIL
    const uint32_t lookahead = L - p;
    const uint32_t max_len = lookahead < kMaxMatch ? lookahead : kMaxMatch;
    const uint32_t nice = lookahead < kNiceLen ? lookahead : kNiceLen;
    const uint32_t limit = p > kMaxDist ? p - kMaxDist : 0;
    const uint32_t pp = p - org;
    uint32_t best = kMinMatch - 1, best_pos = 0, n = 0, snap = 0xffffffffu;
    // filter word = the last bytes a longer match must share with the scan: bytes 0..2 while
    // best == 2 (the trigram itself), bytes best-3..best afterwards (zlib's scan_end test widened
    // to four bytes: fewer false hits, and a hit costs a full comparison)
    uint32_t f_off = 0, f_mask = 0xffffffu;
    uint32_t scan_w = load_u32(data, pp) & f_mask;
    // Single-exit loops: a "nice" match ends the search by zeroing the next link, so each loop
    // predicate is just (next > limit && n < bound) and the rare full comparison never touches
    // control flow outside its own branch.  The walk is split at 32 candidates so that the snapshot
    // zlib's short chain would return costs nothing per candidate (taking it right after a nice
    // match is harmless: it then equals the final record).
    auto examine = [&]() {
        const uint32_t c = cur - org;
        const uint32_t x = (load_u32(data, c + f_off) & f_mask) ^ scan_w;
        uint32_t next = link[c];
        if (x == 0) {                                    // rare: worth a full comparison
            const uint32_t len = match_len_from(data, c, pp, 0u, max_len);
            if (len > best) {
                best = len; best_pos = cur;
                if (len >= nice) next = 0;
                else { f_off = best - 3u; f_mask = 0xffffffffu; scan_w = load_u32(data, pp + f_off); }
            }
        }
        n++;
        cur = next;
    };
    do examine(); while (cur > limit && n < kShortChain);
    if (n == kShortChain) {
        snap = best >= kMinMatch ? entry_pack(best, p - best_pos) : 0u;
        while (cur > limit && n < kMaxChain) examine();
    }
    e128 = best >= kMinMatch ? entry_pack(best, p - best_pos) : 0;
    e32 = snap != 0xffffffffu ? snap : e128;
    // TOO_FAR: a minimum-length match further than 4096 back is dropped (deflate_slow)
    if (entry_len(e128) == kMinMatch && entry_dist(e128) > kTooFar) e128 = 0;
    if (entry_len(e32) == kMinMatch && entry_dist(e32) > kTooFar) e32 = 0;
}

// ---------------------------------------------------------------------------------------------
// Lazy-evaluation walk over the tables (deflate_slow's control flow with longest_match replaced
// by lookups).  Output is positional, three bit masks over the chunk's positions:
//   sym   bit p = position p starts a symbol (literal or match)
//   mst   bit p = the symbol starting at p is a match
//   m32   bit p = that match is the e32 record (else the e128 record)
// Literals are implicit: sym & ~mst.  has128 (input) marks positions whose e128 record is
// non-empty, letting the walk jump over literal runs.
struct MaskWriter {
    uint64_t* dst; uint32_t word; uint64_t cur;
    ZWZ_HD void init(uint64_t* d, uint64_t fill) { dst = d; word = 0; cur = fill; }
};

struct ParseResult { uint32_t n_sym; uint32_t n_match; uint32_t last_is_match; };

template <class EntryFn, class HasFn>
ZWZ_HD ParseResult lz_parse(EntryFn entries /* (p, which32) -> packed entry */, HasFn has128 /* word index -> mask */, uint32_t L,
                            uint64_t* sym, uint64_t* mst, uint64_t* m32) {
    const uint32_t nwords = (L + 63) >> 6;
    // masks are built word by word: the walk is monotone in position
    uint32_t w = 0;                 // current output word
    uint64_t sym_w = ~0ull, mst_w = 0, m32_w = 0;
    uint32_t n_match = 0, covered = 0;
    uint32_t p = 0, b = kMinMatch - 1, b_entry = 0, b_sel = 0;
    ParseResult r; r.last_is_match = 0;

    auto flush_to = [&](uint32_t nw) {   // emit words [w, nw)
        while (w < nw) {
            sym[w] = sym_w; mst[w] = mst_w; m32[w] = m32_w;
            w++; sym_w = ~0ull; mst_w = 0; m32_w = 0;
        }
    };
    auto emit_match = [&](uint32_t q, uint32_t len, uint32_t sel) {
        flush_to(q >> 6);
        mst_w |= 1ull << (q & 63);
        if (sel) m32_w |= 1ull << (q & 63);
        n_match++; covered += len - 1;
        // clear sym bits (q, q+len)
        uint32_t a = q + 1, e = q + len;
        while (a < e) {
            uint32_t wi = a >> 6;
            flush_to(wi);
            uint32_t lo = a & 63, hi = (e - (wi << 6)) < 64 ? (e - (wi << 6)) : 64;
            uint64_t m = (hi == 64 ? ~0ull : ((1ull << hi) - 1)) & ~((1ull << lo) - 1);
            sym_w &= ~m;
            a = (wi << 6) + hi;
        }
    };

    while (p < L) {
        if (b < kMinMatch) {
            // nothing pending: every position without an e128 record is a plain literal
            uint32_t wi = p >> 6;
            uint64_t bits = has128(wi) >> (p & 63);
            if (bits == 0) {
                uint64_t hw = 0;
                wi++;
                while (wi < nwords && (hw = has128(wi)) == 0) wi++;
                if (wi >= nwords) break;
                p = (wi << 6) + (uint32_t)__builtin_ctzll(hw);
            } else {
                p += (uint32_t)__builtin_ctzll(bits);
            }
            if (p >= L) break;
            b_entry = entries(p, 0u); b_sel = 0;
            b = entry_len(b_entry);           // >= 3 by construction of has128
            p++;
            continue;
        }
        // a match of length b found at p-1 is pending
        uint32_t cur = 0, sel = 0;
        if (b < kMaxLazy && p < L) {
            sel = b >= kGoodLen ? 1u : 0u;
            cur = entries(p, sel);
            if (entry_len(cur) <= b) cur = 0;     // no improvement
        }
        if (cur == 0) {
            emit_match(p - 1, b, b_sel);
            r.last_is_match = (p - 1 + b == L);
            p = p - 1 + b; b = kMinMatch - 1;
        } else {
            b_entry = cur; b_sel = sel; b = entry_len(cur);  // literal at p-1, longer match pending
            p++;
        }
    }
    if (b >= kMinMatch) {                        // data ended with a match pending (p == L)
        emit_match(p - 1, b, b_sel);
        r.last_is_match = 1;
    }
    flush_to(nwords);
    if (L & 63) {                                // clear bits >= L in the last word
        uint64_t keep = (1ull << (L & 63)) - 1;
        sym[nwords - 1] &= keep;
    }
    r.n_match = n_match;
    r.n_sym = L - covered;
    return r;
}

// ---------------------------------------------------------------------------------------------
// Block-parallel form of the walk (what the lz_parse kernel runs; the sequential lz_parse above is
// its specification and both are diffed on the host by tests/).
//
// "Fresh" state at position q = nothing pending (zlib's prev_length == 2).  From a fresh q the
// walk's next fresh position depends only on the records at q .. q+13:
//   no e128 record at q        -> literal, next fresh = q + 1
//   record of length b at q    -> lazy chain: while b < 16 and the record consulted at the next
//                                 position (e32 if b >= 8 else e128) is longer, move on; the match
//                                 emitted starts at m (q <= m <= q + 12), the positions q .. m-1
//                                 are literals, next fresh = m + len.
// Every position's transition is computed independently; the walk itself is then the orbit of
// position 0 under "next fresh", found 64 positions at a time.
struct FreshStep {
    uint32_t next;      // next fresh position (q + 1 for a literal)
    uint32_t mpos;      // start of the emitted match (unused for a literal)
    uint32_t sel;       // 1 = the match is the e32 record of mpos
    uint32_t is_lit;
};

template <class EntryFn>
ZWZ_HD FreshStep fresh_step(EntryFn entries /* (p, sel) -> record, 0 beyond the chunk */, uint32_t q, uint32_t L) {
    FreshStep r;
    uint32_t e = entries(q, 0u);
    if (e == 0) { r.next = q + 1; r.mpos = q; r.sel = 0; r.is_lit = 1; return r; }
    uint32_t b = entry_len(e), m = q, sel = 0;
    while (b < kMaxLazy && m + 1 < L) {
        const uint32_t s2 = b >= kGoodLen ? 1u : 0u;
        const uint32_t c = entries(m + 1, s2);
        if (entry_len(c) <= b) break;
        b = entry_len(c); sel = s2; m++;
    }
    r.next = m + b; r.mpos = m; r.sel = sel; r.is_lit = 0;
    return r;
}

// Inclusive prefix XOR over the bits of a word: bit i of the result = XOR of bits 0..i.
ZWZ_HD uint64_t prefix_xor64(uint64_t x) {
    x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32;
    return x;
}

}  // namespace zwz
