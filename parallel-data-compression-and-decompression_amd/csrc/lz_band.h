// lz_band.h -- match records from the chunk's positions SORTED by (hash bucket, position): the candidates zlib's
// longest_match() would walk for a position (consumer(), compression.cpp:119-131) are then the entries right in front
// of it, so "the chain walk" becomes a banded comparison of neighbouring array elements -- no pointer chase.
//
// Same records as lz_core.h's lz_search() (which stays the specification and what sparse chunks run):
//   * entry u of the sorted array = (h << 16 | p); entry u - k is p's k-th candidate, newest first, as long as it is in
//     the same bucket, is not position 0 (zlib's NIL), and lies within MAX_DIST (band_count);
//   * zlib keeps the FIRST candidate of the greatest length, so the answer is max over k of (length, -k): one
//     v_max_u32 per candidate on a packed key (band_key).  Lengths are compared on eight bytes kept beside each
//     entry -- bytes 3..10 behind the trigram where every bucket of the tile holds one trigram only ("pure"), else bytes
//     0..7 -- which settles every length below kDeep (11 / 8);
//   * a position whose best candidate agrees on all eight bytes is finished by a second pass that visits only the
//     candidates agreeing on all eight (its "sharers", chained nearest-first through link[]), comparing real bytes;
//   * zlib's short chain (32 candidates once the previous match is "good") is the key after the 32nd candidate.
// The last positions of a chunk (lookahead < kDeep): their eight bytes reach past the data, so their XORs are masked
// down to the bytes that exist (band_tail_mask) -- "all of them agree" then means "agrees to the end of the data", which
// is as long as a match can get there (and zlib stops at the first such candidate: the nearest).
//
// Portable (host + device): tests/emu builds band_records() on the CPU and diffs it against lz_search().
#pragma once
#include "lz_core.h"

namespace zwz {

constexpr uint32_t kBand = kMaxChain;            // candidates a position may look at: the band's width
constexpr uint32_t kBandNoLink = 0xffffu;
constexpr uint32_t kBandHaloWord = 0xffff0000u;  // "no entry": bucket 0xffff (no hash is), position 0

// lz_dense_list's sample: the chunk's first kDenseSample trigrams.
constexpr uint32_t kDenseSample = 2048;
// Round 5: the rule weighs what the search would have to walk, not how often buckets repeat.  The sample counts the positions that are at least the SECOND
// in their bucket (repeats) and those that are at least the THIRD (thirds).  With the tail taken as geometric the mean number of earlier same-bucket
// positions in the sample would be repeats + thirds / (1 - thirds / repeats), over `sampled`; text's buckets are heavier-tailed than that (its commonest
// trigrams: 12.6 candidates a position in an 8 KB chunk where the geometric tail says 5.9), so the tail term counts three times; over the whole chunk
// the mean is L / sampled times the sample's.  Chain-heavy = 3.6 or more: where the two paths cross on image-like files (mean 16 KB: sort + band 26.7 ms,
// chain walk 28.1 per 1.3 GB; mean 6.8 KB -- BASELINE configs[3] --: 33.1 against 22.5), with text-like chunks on the band from 4 KB up (44 against
// 89 ms) -- tools/exp/dense_crossover.sh, tools/exp/dense_sample.py; tests/test_emu.py pins the choices.  (Rounds 3 - 4: repeats >= sampled / 5, whatever
// the length: the 7 KB image-like files, 0.24 repeats and 1.2 candidates a position, went to the band.)  Either path gives the same records: the
// choice is speed only.
ZWZ_HD bool sample_is_dense(uint32_t repeats, uint32_t thirds, uint32_t sampled, uint32_t L) {
    if (repeats == 0u || sampled == 0u) return false;
    if (thirds >= repeats) return true;                                       // every repeat is a third one: a run
    const float walk = ((float)repeats + 3.0f * (float)thirds * (float)repeats / (float)(repeats - thirds)) * (float)L / ((float)sampled * (float)sampled);
    return walk >= 3.6f;
}

ZWZ_HD uint32_t band_word(uint32_t h, uint32_t p) { return h << 16 | p; }
ZWZ_HD uint32_t band_pos(uint32_t w) { return w & 0xffffu; }
ZWZ_HD uint32_t band_hash(uint32_t w) { return w >> 16; }

// Is `c` (a sorted-array word) a candidate of the position behind `own`, as the second or a later one of its chain?
// (same bucket; not position 0, which reads back as NIL and ends any chain; nearer than MAX_DIST.)  Monotone along the
// array: once an entry fails, every entry further back fails too.
ZWZ_HD bool band_valid(uint32_t own, uint32_t c) {
    // (with equal buckets the words' difference IS the positions' difference: c precedes own in its bucket)
    const uint32_t d = own - c;
    return (own ^ c) < 0x10000u && d < kMaxDist && d != band_pos(own);
}
// The first candidate alone may sit at exactly MAX_DIST -- unless zlib's window has slid by then (lz_search's start test).
ZWZ_HD bool band_first_at_max_dist(uint32_t own, uint32_t c) {
    const uint32_t p = band_pos(own), q = band_pos(c);
    return band_hash(c) == band_hash(own) && q != 0u && p - q == kMaxDist && !(p >= kSlidePos && q <= kWSize);
}

// Number of candidates of entry u: binary search over the monotone band_valid.  S(i) -> word of array index i (the
// caller's array starts kBand entries before its first own entry, filled with kBandHaloWord where nothing exists).
template <class SFn>
ZWZ_HD uint32_t band_count(SFn S, uint32_t u) {
    const uint32_t own = S(u);
    uint32_t k = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t step = kBand; step >= 1u; step >>= 1) {      // largest k in [0, 128] with band_valid(S(u - k))
        const uint32_t t = k + step;
        if (t <= kBand && band_valid(own, S(u - t))) k = t;
    }
    if (k == 0u && band_first_at_max_dist(own, S(u - 1u))) k = 1u;
    return k;
}

// Packed comparison key of the k-th candidate: (equal leading bytes of the two 8-byte words) << 8 | (129 - k); 15 in the
// length field = all eight agree.  Greater key = longer, then nearer.
ZWZ_HD uint32_t band_ctz64(uint32_t lo, uint32_t hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t t0, t1;
    asm("v_ffbl_b32 %0, %1" : "=v"(t0) : "v"(lo));            // 0xffffffff for 0
    asm("v_ffbl_b32 %0, %1" : "=v"(t1) : "v"(hi));
    t1 |= 32u;
    return t0 < t1 ? t0 : t1;
#else
    if (lo) return (uint32_t)__builtin_ctz(lo);
    if (hi) return 32u + (uint32_t)__builtin_ctz(hi);
    return 0xffffffffu;
#endif
}
ZWZ_HD uint32_t band_key(uint32_t own_lo, uint32_t own_hi, uint32_t c_lo, uint32_t c_hi, uint32_t k) {
    const uint32_t t = band_ctz64(own_lo ^ c_lo, own_hi ^ c_hi);
    return ((t & 0x78u) << 5) | (129u - k);
}
// Masked form for a position whose word reaches past the end of the data: only the first `nbytes` bytes count.
ZWZ_HD uint32_t band_key_masked(uint32_t own_lo, uint32_t own_hi, uint32_t c_lo, uint32_t c_hi, uint32_t m_lo, uint32_t m_hi, uint32_t k) {
    const uint32_t t = band_ctz64((own_lo ^ c_lo) & m_lo, (own_hi ^ c_hi) & m_hi);
    return ((t & 0x78u) << 5) | (129u - k);
}
ZWZ_HD uint32_t band_tail_bytes(bool pure, uint32_t lookahead) {       // bytes of the word inside the data (lookahead >= 3)
    const uint32_t nb = pure ? lookahead - 3u : lookahead;
    return nb < 8u ? nb : 8u;
}
ZWZ_HD uint32_t band_tail_mask(uint32_t nbytes, uint32_t word /* 0 = low, 1 = high */) {
    const uint32_t nb = nbytes > 4u * word ? nbytes - 4u * word : 0u;
    return nb >= 4u ? 0xffffffffu : (1u << (8u * nb)) - 1u;
}
constexpr uint32_t kBandKeyNonePure = 0u;          // below every key
constexpr uint32_t kBandKeyNoneImpure = 0x2ffu;    // above every key of fewer than three equal bytes: the trigram itself differs
ZWZ_HD uint32_t band_key_len(uint32_t key) { return key >> 8; }            // equal bytes of the eight, 15 = all
ZWZ_HD uint32_t band_key_k(uint32_t key) { return 129u - (key & 0xffu); }

// Record of a key: cand = position of the winner.  "All eight agree" (15) is a record only for a tail position, whose
// match then runs to the end of the data (lookahead bytes); elsewhere it is the second pass's business.
ZWZ_HD uint32_t band_record(uint32_t key, bool pure, uint32_t p, uint32_t cand_pos, uint32_t tail_len = 0) {
    const uint32_t len = band_key_len(key) == 15u ? tail_len : band_key_len(key) + (pure ? 3u : 0u), dist = p - cand_pos;
    if (len == kMinMatch && dist > kTooFar) return 0u;
    return entry_pack(len, dist);
}

// The chunk's last positions (and the specification of the whole scheme): candidates u - 1 .. u - cnt, real bytes.
template <class SFn>
ZWZ_HD void band_generic(const uint8_t* data, SFn S, uint32_t u, uint32_t cnt, uint32_t L, uint32_t& e128, uint32_t& e32) {
    const uint32_t p = band_pos(S(u)), lookahead = L - p;
    const uint32_t max_len = lookahead < kMaxMatch ? lookahead : kMaxMatch, nice = lookahead < kNiceLen ? lookahead : kNiceLen;
    uint32_t best = kMinMatch - 1u, best_pos = 0, snap = 0xffffffffu;
    for (uint32_t k = 1; k <= cnt; k++) {
        const uint32_t c = band_pos(S(u - k));
        const uint32_t len = match_len_from(data, c, p, 0u, max_len);
        if (len > best) { best = len; best_pos = c; }
        if (k == kShortChain) snap = best >= kMinMatch ? entry_pack(best, p - best_pos) : 0u;
        if (best >= nice) break;
    }
    e128 = best >= kMinMatch ? entry_pack(best, p - best_pos) : 0u;
    e32 = snap != 0xffffffffu ? snap : e128;
    if (entry_len(e128) == kMinMatch && entry_dist(e128) > kTooFar) e128 = 0;
    if (entry_len(e32) == kMinMatch && entry_dist(e32) > kTooFar) e32 = 0;
}

// Second pass of a position whose nearest sharer is its k1-th candidate: the sharers, nearest first, compared byte by
// byte from kDeep on.  link(j) -> array index of the nearest sharer of entry j, kBandNoLink if none in ITS band; valid
// for the tile's own entries (j >= first_own) -- through the halo in front of them the sharers are found by their
// eight bytes (E(j) -> the 8-byte word as lo | hi << 32, asked for halo entries only).
template <class SFn, class LinkFn, class EFn>
ZWZ_HD void band_deep(const uint8_t* data, SFn S, LinkFn link, EFn E, uint32_t first_own, uint32_t u, uint32_t cnt, uint32_t k1,
                      uint32_t deep, uint32_t L, uint64_t own /* entry u's 8-byte word */, uint32_t& e128,
                      uint32_t& e32 /* in: the first pass's, kept if k1 > 32 */) {
    const uint32_t p = band_pos(S(u)), lookahead = L - p;
    const uint32_t max_len = lookahead < kMaxMatch ? lookahead : kMaxMatch, nice = lookahead < kNiceLen ? lookahead : kNiceLen;
    uint32_t best = 0, best_pos = 0, snap = 0xffffffffu;
    uint32_t j = u - k1;
    for (;;) {
        const uint32_t k = u - j;
        if (k > kShortChain && snap == 0xffffffffu) snap = best ? entry_pack(best, p - best_pos) : 0u;   // (0 only if k1 > 32: then unused)
        const uint32_t c = band_pos(S(j));
        const uint32_t len = match_len_from(data, c, p, deep, max_len);
        if (len > best) { best = len; best_pos = c; if (len >= nice) break; }
        if (j >= first_own) {
            const uint32_t j2 = link(j);
            if (j2 == kBandNoLink || u - j2 > cnt) break;
            j = j2;
        } else {
            uint32_t k2 = k + 1u;
            while (k2 <= cnt && E(u - k2) != own) k2++;
            if (k2 > cnt) break;
            j = u - k2;
        }
    }
    e128 = entry_pack(best, p - best_pos);
    if (k1 <= kShortChain) e32 = snap != 0xffffffffu ? snap : e128;
}

}  // namespace zwz
