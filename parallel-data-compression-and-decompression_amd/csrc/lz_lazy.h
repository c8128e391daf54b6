// lz_lazy.h -- zlib's deflate_slow() as zlib runs it: the lazy parse asks for a search only where it stands
// (consumer(), compression.cpp:119-131: deflate(Z_FINISH) at level 6 -> deflate_slow + longest_match).
//
// lz_match / lz_match_band compute both records of EVERY position and lz_parse then walks the table.  zlib itself hands
// longest_match only the positions its parse visits with fewer than max_lazy bytes pending: on the text corpus a quarter of
// the positions and 5.3 candidates a position against the band's 38.6 (tools/exp/searched_set.py).  This file is that
// algorithm in a form many lanes can run on ONE chunk:
//
//   * the chunk is cut into segments; lane i starts at its segment's first position ASSUMING nothing is pending there
//     ("fresh": zlib's prev_length == 2) and parses on, searching on demand over the positions sorted by (bucket, position)
//     -- lz_sort / lz_place's arrays: the candidates of p are the entries in front of dest[p];
//   * what "fresh at q" leads to -- a literal, or a lazy chain that ends in the match (m, len, dist) and is fresh again at
//     m + len -- depends on q alone, not on how q was reached: a lane memoises it in step[q] (lazy_chain);
//   * a lane marks the positions where it starts a fresh search: in its own segment in F; beyond it (it runs on until it
//     meets its successor's path) in G.  Reaching, fresh, a position an OWNER marked in F, it stops: from there the two
//     parses are one (term[i]);
//   * lane 0's assumption is true; a lane's path is true from the point a true lane stopped at in its segment: a chain
//     0 -> seg(term[0]) -> ... of lanes, each owning the piece [merge, term) of the true parse (lazy_resolve);
//   * a piece is replayed from the marks: the next true fresh search is the first mark at or behind the position the parse
//     is fresh at -- a position that was searched has candidates, and the first position with candidates is where the true
//     parse searches next, so no mark of a false path can come first (lazy_emit_piece).
// The searches are longest_match itself (best starts at prev_length, chain 32 once prev_length >= 8, nice 128, MAX_DIST,
// position 0 = NIL, the slid window at 65 274); lz_core.h's lz_search + lz_parse remain the specification:
// tests/emu runs this decomposition, lanes in random order, against them.
//
// Portable (host + device).
#pragma once
#include "lz_core.h"

namespace zwz {

// step word of a fresh position q.  0 = searched, nothing found: literal, fresh again at q + 1.
// Else the lazy chain from q emits literals q .. m-1 and the match (m, len, dist): len | (m - q) << 9 | dist << 16.
ZWZ_HD uint32_t lazy_step_pack(uint32_t moff, uint32_t len, uint32_t dist) { return len | moff << 9 | dist << 16; }
ZWZ_HD uint32_t lazy_step_len(uint32_t w) { return w & 0x1ffu; }
ZWZ_HD uint32_t lazy_step_moff(uint32_t w) { return (w >> 9) & 0xfu; }
ZWZ_HD uint32_t lazy_step_record(uint32_t w) { return (w & 0x1ffu) | (w & 0xffff0000u); }     // entry_pack(len, dist)

// Candidates of position p = the `avail` sorted entries in front of index u = dest[p] that share its bucket, position 0
// (zlib's NIL, first of its bucket) not counted.  bend[h] = end of bucket h in the sorted array (lz_sort's table after ranking).
ZWZ_HD uint32_t lazy_avail(uint32_t u, uint32_t h, uint32_t bucket_start /* h ? bend[h - 1] : 0 */, uint32_t h0 /* bucket of position 0 */) {
    const uint32_t first = bucket_start + (h == h0 ? 1u : 0u);
    return u > first ? u - first : 0u;
}

// longest_match(p) with prev_length = prev_len (2 = nothing pending).  spos(i) -> position of sorted index i.
// Returns the best length: > prev_len = a longer match at best_pos, else nothing (prev_len).  deflate_slow's TOO_FAR rule
// (a length-3 match further than 4096 back is dropped) is applied here: it can only strike when prev_len == 2.
template <class SposFn>
ZWZ_HD uint32_t lazy_search(const uint8_t* data, SposFn spos, uint32_t u, uint32_t avail, uint32_t p, uint32_t L, uint32_t prev_len, uint32_t& best_pos) {
    if (avail == 0u || p + kMinMatch > L) return prev_len;
    uint32_t c = spos(u - 1u);
    if (p - c > kMaxDist) return prev_len;                    // first candidate: distance <= MAX_DIST
    if (p >= kSlidePos && c <= kWSize) return prev_len;       // zlib's window has slid: reads as NIL
    const uint32_t lookahead = L - p;
    const uint32_t max_len = lookahead < kMaxMatch ? lookahead : kMaxMatch;
    const uint32_t nice = lookahead < kNiceLen ? lookahead : kNiceLen;
    const uint32_t limit = p > kMaxDist ? p - kMaxDist : 0u;
    const uint32_t chain = prev_len >= kGoodLen ? kShortChain : kMaxChain;
    const uint32_t n = avail < chain ? avail : chain;
    uint32_t best = prev_len;
    // filter: a candidate beats `best` only if it agrees with p on bytes best-3 .. best (the trigram itself while best == 2)
    uint32_t f_off = best >= kMinMatch ? best - 3u : 0u, f_mask = best >= kMinMatch ? 0xffffffffu : 0xffffffu;
    uint32_t scan_w = load_u32(data, p + f_off) & f_mask;
    for (uint32_t k = 1;; k++) {
        if (((load_u32(data, c + f_off) & f_mask) ^ scan_w) == 0u) {
            const uint32_t len = match_len_from(data, c, p, 0u, max_len);
            if (len > best) {
                best = len; best_pos = c;
                if (len >= nice) break;
                f_off = best - 3u; f_mask = 0xffffffffu; scan_w = load_u32(data, p + f_off);
            }
        }
        if (k == n) break;
        c = spos(u - k - 1u);
        if (c <= limit) break;                                // later candidates: strictly nearer than MAX_DIST
    }
    if (best == kMinMatch && prev_len < kMinMatch && p - best_pos > kTooFar) return prev_len;
    return best;
}

// What "fresh at q" leads to (lz_core.h fresh_step, searches on demand).  search(p, prev_len, best_pos&) -> best length.
struct LazyChain { uint32_t step, next; };
template <class SearchFn>
ZWZ_HD LazyChain lazy_chain(SearchFn search, uint32_t q, uint32_t L) {
    uint32_t bp = 0;
    uint32_t b = search(q, kMinMatch - 1u, bp);
    LazyChain r;
    if (b < kMinMatch) { r.step = 0u; r.next = q + 1u; return r; }
    uint32_t m = q;
    while (b < kMaxLazy && m + 1u < L) {
        uint32_t bp2 = 0;
        const uint32_t b2 = search(m + 1u, b, bp2);
        if (b2 <= b) break;
        b = b2; bp = bp2; m++;
    }
    r.step = lazy_step_pack(m - q, b, m - bp); r.next = m + b;
    return r;
}

// The chain of true lanes.  term[i] = where lane i stopped (a position an owner had marked, or L); seg(q) = owner of q.
// Fills merge[i] (0xffffffff = lane i's path never becomes the true one).  Lanes stop beyond their own segment, so one
// ascending pass settles it.
template <class SegFn>
ZWZ_HD void lazy_resolve(const uint32_t* term, uint32_t n_lanes, uint32_t L, SegFn seg, uint32_t* merge) {
    for (uint32_t i = 0; i < n_lanes; i++) merge[i] = 0xffffffffu;
    if (n_lanes) merge[0] = 0u;
    for (uint32_t i = 0; i < n_lanes; i++) {
        if (merge[i] == 0xffffffffu || term[i] >= L) continue;
        const uint32_t k = seg(term[i]);
        if (term[i] < merge[k]) merge[k] = term[i];
    }
}

// Replay of one piece [from, to) of the true parse over the marks.  marks(w) -> 64 marks of positions 64 w ..; step(q) -> step word.
// on_match(m, len, record) is called for every match in order.  Returns the position the parse is fresh at when it leaves the piece.
template <class MarksFn, class StepFn, class MatchFn>
ZWZ_HD uint32_t lazy_emit_piece(MarksFn marks, StepFn step, uint32_t from, uint32_t to, MatchFn on_match) {
    uint32_t nf = from;
    for (uint32_t w = from >> 6; (w << 6) < to; w++) {
        uint64_t bits = marks(w);
        while (bits) {
            const uint32_t q = (w << 6) + (uint32_t)__builtin_ctzll(bits);
            bits &= bits - 1ull;
            if (q < nf) continue;                 // inside a lazy chain or a match: a mark of some false path
            if (q >= to) return nf;
            const uint32_t s = step(q);
            if (s == 0u) { nf = q + 1u; continue; }
            const uint32_t m = q + lazy_step_moff(s), len = lazy_step_len(s);
            on_match(m, len, lazy_step_record(s));
            nf = m + len;
        }
    }
    return nf;
}

}  // namespace zwz
