// zwz_lazy.hip -- lz_lazy: match search AND lazy parse of a chain-heavy chunk in one kernel, the searches on demand.
// Replaces, for those chunks, lz_match_band + lz_parse (consumer()'s deflate(Z_FINISH), compression.cpp:119-131: zlib's
// deflate_slow + longest_match).  The algorithm, its proof and its CPU form are csrc/lz_lazy.h; tests/emu runs that form, lanes
// in random order, against lz_core.h's records + table walk.
//
// Why: the band computes both records of every position (38.6 candidates a position on the text corpus); zlib's parse hands
// longest_match a quarter of the positions, 5.3 candidates a position (tools/exp/searched_set.py).  The parse is sequential, so
// the chunk is cut into 512 segments of 128 positions, one LANE each, every lane starting on the assumption that nothing is
// pending at its segment's head; a lane runs on past its segment until it stands, fresh, on a position its successor's
// owner has searched fresh too.  On the text corpus that costs 5 % more searches than one sequential parse (19 415 against
// 18 404 a chunk).
//
//   phase 0   the chunk's bytes -> LDS (64 KB); F (the owners' fresh-search marks, LDS) and G (the marks of lanes beyond their
//             segment, global) cleared
//   phase 1   every lane: zlib's loop.  A search walks the entries in front of dest[p] in the array sorted by (bucket, position)
//             (lz_sort + lz_place): eight candidates a 16-byte load, the next eight asked for before these are looked at; the filter
//             (the four bytes a longer match must share) is gathered for all eight at once, a full comparison only on a hit.
//             What "fresh at q" led to goes into step[q] (global, 4 bytes a fresh search)
//   phase 2   the chain of true lanes (lazy_resolve), every true lane replays its piece over F | G (lazy_emit_piece): match
//             starts and covered positions into the sym / mst masks (LDS), then once more with its matches' place in the
//             stream known, for the chosen records -- exactly lz_parse's outputs (sym, mst, chosen, ChunkInfo)
//
// LDS: 65 552 + 8 192 bytes and a few words: two workgroups of eight waves a CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include "lz_lazy.h"
#include "zwz_kernels.h"
#include "zwz_device.h"

namespace zwz {

#ifndef ZWZ_LAZY_EXP
#define ZWZ_LAZY_EXP 0
#endif

constexpr uint32_t kLazyDataBytes = 65536 + 16;                     // the chunk + slack for comparisons that read past its end
constexpr uint32_t kLazyLdsBytes = kLazyDataBytes + 8192;           // + F
// phase 2 reuses the bytes: sym and mst masks (2 x 8 KB), then per-lane words
constexpr uint32_t kLazyOffSym = 0, kLazyOffMst = 8192, kLazyOffTerm = 16384, kLazyOffMerge = kLazyOffTerm + 4 * kLazyThreads,
                   kLazyOffCnt = kLazyOffMerge + 4 * kLazyThreads, kLazyOffWave = kLazyOffCnt + 4 * kLazyThreads;
static_assert(kLazyThreads * kLazySeg == 65536 && kLazyOffWave + 256 <= kLazyDataBytes, "lz_lazy layout");

// (ZWZ_LAZY_EXP & 16, experiment builds: lane 0 of every wave adds cycles >> 8 per phase and trip statistics to tickets[16 ..]; launch_lazy prints them
// when ZWZ_LAZY_TIMES is set)
#if ZWZ_LAZY_EXP & 16
#define ZWZ_LSTAMP(ph) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); lacc_[(ph)] += (uint32_t)(now_ - lstamp_); lstamp_ = now_; } while (0)
#define ZWZ_LCOUNT(i, v) do { lcnt_[(i)] += (v); } while (0)
#else
#define ZWZ_LSTAMP(ph) do { } while (0)
#define ZWZ_LCOUNT(i, v) do { } while (0)
#endif

enum : uint32_t { kLzPick = 0, kLzWalk = 1, kLzDone = 3 };

static __device__ __forceinline__ uint32_t lds_u32(const uint8_t* base, uint32_t off) {       // 4 bytes at any offset of the LDS bytes
    const uint32_t* w = reinterpret_cast<const uint32_t*>(base) + (off >> 2);
    return __builtin_amdgcn_alignbyte(w[1], w[0], off);                   // (v_alignbyte reads the low two bits of its shift)
}

struct __attribute__((packed, aligned(2))) LazyVec { uint32_t x, y, z, w; };          // eight sorted positions at any 2-byte boundary
// A position the parse may go to next, looked up ahead of time.  stage: 0 empty, 1 (dest, bucket start) on their way, 2 its candidates known
// (u = sorted index, bs = how many) and the first eight on their way, 3 those are in, 4 it has no candidates.
struct LazySlot { uint32_t pos, stage, u, bs, first0; LazyVec vec; };

__global__ __launch_bounds__(kLazyThreads, 2) void lz_lazy_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                                  const uint32_t* __restrict__ in_len, const uint32_t* __restrict__ list,
                                                                  uint32_t* __restrict__ tickets, const uint32_t* __restrict__ sorted,
                                                                  uint32_t* __restrict__ scratch /* the chunks' entries space */,
                                                                  uint64_t* __restrict__ sym, uint64_t* __restrict__ mst,
                                                                  ChunkInfo* __restrict__ info, uint16_t* __restrict__ links) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* sdata = smem;
    uint32_t* sF = reinterpret_cast<uint32_t*>(smem + kLazyDataBytes);
    __shared__ uint32_t s_chunk, s_nsym, s_nmatch;
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const uint32_t n_list = tickets[kTicketDenseCount];
#if ZWZ_LAZY_EXP & 16
    uint64_t lstamp_ = __builtin_amdgcn_s_memtime();
    uint32_t lacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lcnt_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        ZWZ_LSTAMP(7);
        __syncthreads();
        if (tid == 0) { s_chunk = atomicAdd(&tickets[kTicketLazyNext], 1u); s_nsym = 0; s_nmatch = 0; }
        __syncthreads();
        const uint32_t t = s_chunk;
        if (t >= n_list) break;
        const uint32_t chunk = list[t];
        const uint32_t L = in_len[chunk];
        const uint16_t* dest = reinterpret_cast<const uint16_t*>(sorted + (size_t)chunk * kSortedStride);
        const uint16_t* spos = dest + 65536;
        uint32_t* scr = scratch + (size_t)chunk * kLazyScratchWords;
        uint32_t* stepw = scr + kLazyStepOff;
        uint32_t* gG = scr + kLazyMarkOff;
        const uint16_t* bend = reinterpret_cast<const uint16_t*>(scr + kLazyBendOff);

        // ---- phase 0 ----
        {
            const uint4* src = reinterpret_cast<const uint4*>(in + in_off[chunk]);
            copy_vec16(reinterpret_cast<uint4*>(sdata), src, (L + 15u) >> 4);
            for (uint32_t i = tid; i < 2048u; i += kLazyThreads) { sF[i] = 0; gG[i] = 0; }
        }
        __syncthreads();
        const uint32_t h0 = L >= kMinMatch ? hash3(sdata[0], sdata[1], sdata[2]) : 0u;
        ZWZ_LSTAMP(0);

        // ---- phase 1: the lanes ----
        // One trip of the loop below = at most ONE global round trip for a lane: what was asked for in the last trip has arrived at the top of
        // this one, everything this trip needs next is asked for before its eight candidates are looked at.  A position's search needs two
        // dependent fetches -- (dest[p], its bucket's start), then the eight entries in front of dest[p] -- so both are asked for ahead of time,
        // for the two positions the parse can go to from the one it is searching: p + 1 (slot A) and, behind a pending match, the match's end
        // (slot B).  By the time a search of two or more trips ends, the next one's first candidates are in registers.
        // The trip is written as straight-line code on selects: a lane makes ONE step of deflate_slow's bookkeeping a trip (the decision behind
        // a search, or a position without one, or the start of the next search); every nested lane-divergent branch was a dozen scalar
        // instructions of exec-mask bookkeeping for the whole wave (the first form: 770 instructions a trip, half of them scalar).
        uint32_t p = tid * kLazySeg;
        const uint32_t own_end = p + kLazySeg;
        uint32_t b = kMinMatch - 1u, bpos = 0, q0 = 0, term = L;
        uint32_t st = p < L ? kLzPick : kLzDone;
        bool have = false;                                        // a search has ended: its result waits for deflate_slow's decision
        uint32_t ui = 0, nleft = 0, best = 0, best_pos = 0, f_off = 0, f_mask = 0, scan_w = 0, kfirst = 0, max_len = 0, nice = 0;
        LazyVec cur = {0, 0, 0, 0}, nxt = {0, 0, 0, 0};
        uint32_t posA = 0, stA = 0, uA = 0, nA = 0, posB = 0, stB = 0, uB = 0, nB = 0;          // slots: position, stage, sorted index, (bucket start + NIL, then) candidates
        LazyVec vecA = {0, 0, 0, 0}, vecB = {0, 0, 0, 0};
        for (;;) {
            if (__builtin_amdgcn_ballot_w64(st != kLzDone) == 0ull) break;
            ZWZ_LCOUNT(0, 1u); ZWZ_LCOUNT(1, (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(st != kLzDone)));
            // -- (1) what the last trip asked for is here: a looked-up slot knows its candidates (and asks for the first eight below), a slot's eight are in --
            const bool a1 = stA == 1u, b1 = stB == 1u;
            nA = a1 ? (uA > nA ? uA - nA : 0u) : nA;
            nB = b1 ? (uB > nB ? uB - nB : 0u) : nB;
            stA = stA == 2u ? 3u : a1 ? (nA ? 2u : 4u) : stA;
            stB = stB == 2u ? 3u : b1 ? (nB ? 2u : 4u) : stB;
            bool wantA = a1 && nA != 0u, wantB = b1 && nB != 0u;
            if (st == kLzWalk) cur = nxt;
            // -- (2) one step of the parse for the lanes that are not searching --
            bool reqA = false, reqB = false;
            uint32_t rposB = 0;
            if (st == kLzPick) {
                // deflate_slow's decision behind a search
                const bool toofar = best == kMinMatch && b < kMinMatch && p - best_pos > kTooFar;
                const bool imp = have && best > b && !toofar;
                const bool out1 = have && !imp;                                           // the pending match goes out, or a literal
                // a position that needs no search: behind a match of max_lazy bytes or at the end of the data the pending match goes out; the
                // chunk's last two positions and positions without candidates are literals (or end the pending match)
                const bool pend0 = b >= kMinMatch;
                const bool selB = !have && stB != 0u && posB == p;
                if (selB) { posA = posB; stA = stB; uA = uB; nA = nB; vecA = vecB; wantA = wantB; }
                const bool hasA = !have && stA != 0u && posA == p;
                const bool t1 = !have && pend0 && (b >= kMaxLazy || p >= L);
                const bool fin = !have && !t1 && p >= L;
                const bool t3 = !have && !t1 && !fin && (p + kMinMatch > L || (hasA && stA == 4u));
                const bool out2 = t1 || t3;
                const bool emit = (out1 || out2) && pend0;
                const bool lit = (out1 || out2) && !pend0;
                if (emit) stepw[q0] = lazy_step_pack(p - 1u - q0, b, p - 1u - bpos);
                if (out1 && !pend0) stepw[q0] = 0u;                                       // searched, nothing found
                const bool ready = hasA && stA == 3u && !out2 && !fin;
                p = emit ? p - 1u + b : (imp || lit) ? p + 1u : p;
                bpos = imp ? best_pos : bpos;
                b = imp ? best : emit ? kMinMatch - 1u : b;
                if (fin) { term = L; st = kLzDone; }
                bool go = ready;
                if (ready && b < kMinMatch) {                                             // a fresh search: mark it, or meet an owner's mark
                    const uint32_t bit = 1u << (p & 31u);
                    const bool own = p < own_end;
                    const bool met = !own && (sF[p >> 5] & bit) != 0u;
                    if (own) atomicOr(&sF[p >> 5], bit);
                    else if (!met) atomicOr(&gG[p >> 5], bit);
                    if (met) { term = p; st = kLzDone; go = false; }
                    q0 = p;
                }
                if (go) {
                    const uint32_t lookahead = L - p;
                    max_len = lookahead < kMaxMatch ? lookahead : kMaxMatch;
                    nice = lookahead < kNiceLen ? lookahead : kNiceLen;
                    const uint32_t chain = b >= kGoodLen ? kShortChain : kMaxChain;
                    nleft = nA < chain ? nA : chain;
                    ui = uA; best = b; best_pos = 0; kfirst = 1u;
                    f_off = b >= kMinMatch ? b - 3u : 0u; f_mask = b >= kMinMatch ? 0xffffffffu : 0xffffffu;
                    scan_w = lds_u32(sdata, p + f_off) & f_mask;
                    cur = vecA;
                    st = kLzWalk;
                    reqB = b >= kMinMatch; rposB = p - 1u + b;
                }
                // the slot is used up by a search or a step; a changed position asks for its own look-up
                const bool moved = have || out2;
                have = false;
                if (ready || moved || !hasA) { stA = 0u; wantA = false; }
                stB = 0u; wantB = false;
                reqA = st != kLzDone && stA == 0u;                                      // (a search that has just started looks up p + 1)
            }
            const uint32_t rposA = st == kLzWalk ? p + 1u : p;
            // -- (3) everything the next trip needs is asked for now --
            if (wantA) vecA = *reinterpret_cast<const LazyVec*>(spos + uA - 8u);          // (u < 8: reads back into dest[], inside the chunk's arrays)
            if (wantB) vecB = *reinterpret_cast<const LazyVec*>(spos + uB - 8u);
            if (reqA) {
                posA = rposA; stA = 4u;
                if (rposA + kMinMatch <= L) { const uint32_t w = lds_u32(sdata, rposA), h = hash3(w & 0xffu, (w >> 8) & 0xffu, (w >> 16) & 0xffu);
                                              uA = dest[rposA]; nA = (h ? bend[h - 1u] : 0u) + (h == h0 ? 1u : 0u); stA = 1u; }
            }
            if (reqB) {
                posB = rposB; stB = 4u;
                if (rposB + kMinMatch <= L) { const uint32_t w = lds_u32(sdata, rposB), h = hash3(w & 0xffu, (w >> 8) & 0xffu, (w >> 16) & 0xffu);
                                              uB = dest[rposB]; nB = (h ? bend[h - 1u] : 0u) + (h == h0 ? 1u : 0u); stB = 1u; }
            }
            if (st == kLzWalk && nleft > 8u) nxt = *reinterpret_cast<const LazyVec*>(spos + ui - 16u);
            ZWZ_LSTAMP(1);
            ZWZ_LCOUNT(2, (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(st == kLzWalk)));
            // -- (4) eight candidates of every walking lane --
            if (st == kLzWalk) {
                const uint32_t nb = nleft < 8u ? nleft : 8u;
                // nearest first: sorted index ui - 1 is the vector's last half-word
                const uint32_t c[8] = {cur.w >> 16, cur.w & 0xffffu, cur.z >> 16, cur.z & 0xffffu, cur.y >> 16, cur.y & 0xffffu, cur.x >> 16, cur.x & 0xffffu};
                // a fresh search's filter is the trigram itself, which a bucket's entries nearly all share: its nearest candidate is compared in full
                // first, and the filter moves behind the length that gives (zlib looks at it first too).  Out of the window: no search, as in zlib.
                bool stop = false;
                if (kfirst && best < kMinMatch) {
                    if (p - c[0] > (p != kSlidePos ? kMaxDist : kMaxDist - 1u)) stop = true;
                    else {
                        const uint32_t len = match_len_from(sdata, c[0], p, 0u, max_len);
                        if (len >= kMinMatch) {
                            best = len; best_pos = c[0];
                            if (len >= nice) stop = true;
                            else { f_off = best - 3u; f_mask = 0xffffffffu; scan_w = lds_u32(sdata, p + f_off); }
                        }
                    }
                }
                // the filter of all eight (the four bytes a longer match must share with the scan); a candidate's distance is looked at when it hits
                uint32_t hits = 0;
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) hits |= (((lds_u32(sdata, c[j] + f_off) & f_mask) ^ scan_w) == 0u ? 1u : 0u) << j;
                hits &= (1u << nb) - 1u;
                if (stop) hits = 0u;
                bool over = stop || nleft <= 8u;
                // a hit: the full comparison.  The hits behind it stay hits (a candidate that beats the new best shares the old filter's bytes too), so
                // nothing is gathered again.  A hit too far back ends the search: positions fall along the chain.
                while (__builtin_amdgcn_ballot_w64(hits != 0u) != 0ull) {
                    if (hits) {
                        const uint32_t j = (uint32_t)__builtin_ctz(hits);
                        hits &= hits - 1u;
                        const uint32_t hw = 7u - j, wsel = hw >> 1;
                        const uint32_t word = wsel == 0u ? cur.x : wsel == 1u ? cur.y : wsel == 2u ? cur.z : cur.w;
                        const uint32_t cj = (word >> ((hw & 1u) << 4)) & 0xffffu;
                        const uint32_t maxd = (j == 0u && kfirst && p != kSlidePos) ? kMaxDist : kMaxDist - 1u;
                        if (p - cj > maxd) { hits = 0u; over = true; }
                        else {
                            const uint32_t len = match_len_from(sdata, cj, p, 0u, max_len);
                            if (len > best) {
                                best = len; best_pos = cj;
                                if (len >= nice) { hits = 0u; over = true; }
                                else { f_off = best - 3u; f_mask = 0xffffffffu; scan_w = lds_u32(sdata, p + f_off); }
                            }
                        }
                    }
                }
                if (nb == 8u && p - c[7] > kMaxDist - 1u) over = true;                    // the chain has left the window
                nleft -= nb; ui -= 8u; kfirst = 0u;
                if (over) { st = kLzPick; have = true; }
            }
            ZWZ_LSTAMP(2);
        }
        ZWZ_LSTAMP(3);
        __syncthreads();                                       // every lane is done with the bytes
        ZWZ_LSTAMP(4);

        // ---- phase 2 ----
        uint32_t* s_sym = reinterpret_cast<uint32_t*>(smem + kLazyOffSym);
        uint32_t* s_mst = reinterpret_cast<uint32_t*>(smem + kLazyOffMst);
        uint32_t* s_term = reinterpret_cast<uint32_t*>(smem + kLazyOffTerm);
        uint32_t* s_merge = reinterpret_cast<uint32_t*>(smem + kLazyOffMerge);
        uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem + kLazyOffCnt);
        uint32_t* s_wave = reinterpret_cast<uint32_t*>(smem + kLazyOffWave);
        s_term[tid] = term; s_merge[tid] = tid == 0u ? 0u : 0xffffffffu;
        for (uint32_t i = tid; i < 2048u; i += kLazyThreads) {
            const uint32_t lo = i << 5;
            s_sym[i] = lo + 32u <= L ? 0xffffffffu : lo < L ? (1u << (L - lo)) - 1u : 0u;
            s_mst[i] = 0u;
        }
        __threadfence();                                       // the steps and G marks of phase 1 are read by other lanes below
        __syncthreads();
        if (tid == 0) {                                        // the chain of true lanes: lane i's stop makes its owner's path true from there
            const uint32_t nl = (L + kLazySeg - 1u) / kLazySeg;
            for (uint32_t i = 0; i < nl; i++) {
                const uint32_t mi = s_merge[i], ti = s_term[i];
                if (mi == 0xffffffffu || ti >= L) continue;
                const uint32_t k = ti / kLazySeg;
                if (ti < s_merge[k]) s_merge[k] = ti;
            }
        }
        __syncthreads();
        const uint32_t from = s_merge[tid], to = term;
        const bool truelane = from != 0xffffffffu && tid * kLazySeg < L;
        // replay of the piece [from, to) over F | G, eight marks' steps asked for together
        auto replay = [&](bool second, uint32_t base) -> uint32_t {
            uint32_t nf = from, nm = 0;
            uint32_t* chosen = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(links + (size_t)chunk * kLinkStride) + kChosenOffset);
            for (uint32_t w = from >> 5; (w << 5) < to; w++) {
                uint32_t bits = sF[w] | __builtin_nontemporal_load(&gG[w]);
                if (w == (from >> 5)) bits &= ~((1u << (from & 31u)) - 1u);
                while (bits) {
                    uint32_t q[4], sw[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) { q[k] = bits ? (w << 5) + (uint32_t)__builtin_ctz(bits) : 0xffffffffu; bits &= bits - 1u; }
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) sw[k] = q[k] != 0xffffffffu && q[k] < to ? __builtin_nontemporal_load(&stepw[q[k]]) : 0u;
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        if (q[k] == 0xffffffffu || q[k] < nf || q[k] >= to) continue;
                        const uint32_t s = sw[k];
                        if (s == 0u) { nf = q[k] + 1u; continue; }
                        const uint32_t m = q[k] + lazy_step_moff(s), len = lazy_step_len(s);
                        if (!second) {
                            atomicOr(&s_mst[m >> 5], 1u << (m & 31u));
                            uint32_t a = m + 1u, e = m + len;                            // positions (m, m + len) are covered
                            while (a < e) {
                                const uint32_t wi = a >> 5, lo = a & 31u, hi = e - (wi << 5) < 32u ? e - (wi << 5) : 32u;
                                const uint32_t msk = (hi == 32u ? 0xffffffffu : (1u << hi) - 1u) & ~((1u << lo) - 1u);
                                atomicAnd(&s_sym[wi], ~msk);
                                a = (wi << 5) + hi;
                            }
                        } else chosen[min(base + nm, kChosenCap - 1u)] = lazy_step_record(s);
                        nm++;
                        nf = m + len;
                    }
                }
            }
            return nm;
        };
        const uint32_t my_matches = truelane ? replay(false, 0u) : 0u;
        // matches in front of this lane's piece
        const uint32_t incl = wave_scan_incl(my_matches);
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        uint32_t base = incl - my_matches, total = 0;
        for (uint32_t i = 0; i < kLazyThreads / 64u; i++) { const uint32_t v = s_wave[i]; base += i < wave ? v : 0u; total += v; }
        if (truelane && my_matches) replay(true, base);
        // the masks out; symbols counted
        uint32_t ns = 0;
        uint64_t* gsym = sym + (size_t)chunk * kMaskWords;
        uint64_t* gmst = mst + (size_t)chunk * kMaskWords;
        const uint32_t nwords = (L + 63u) >> 6;
        for (uint32_t i = tid; i < nwords; i += kLazyThreads) {
            const uint64_t ws = (uint64_t)s_sym[2 * i] | (uint64_t)s_sym[2 * i + 1] << 32, wm = (uint64_t)s_mst[2 * i] | (uint64_t)s_mst[2 * i + 1] << 32;
            gsym[i] = ws; gmst[i] = wm;
            ns += (uint32_t)__popcll(ws);
        }
        for (uint32_t d = 32; d >= 1; d >>= 1) ns += __shfl_down(ns, d);
        if (lane == 0) atomicAdd(&s_nsym, ns);
        __syncthreads();
        if (tid == 0) {
            uint32_t last_is_match = 0;
            for (uint32_t w = (L - 1u) >> 5;; w--) {                                      // the last symbol: a match?
                const uint32_t v = s_sym[w];
                if (v) { last_is_match = (s_mst[w] >> (31u - (uint32_t)__builtin_clz(v))) & 1u; break; }
                if (w == 0) break;
            }
            ChunkInfo ci;
            ci.n_sym = s_nsym;
            const uint32_t s_in = (ci.n_sym > 0 && !last_is_match) ? ci.n_sym - 1 : ci.n_sym;
            ci.n_blocks = s_in / kSymsPerBlock + 1;
            info[chunk] = ci;
        }
        (void)total;
        ZWZ_LSTAMP(5);
    }
#if ZWZ_LAZY_EXP & 16
    if (lane == 0) { for (uint32_t ph = 0; ph < 8; ph++) { atomicAdd(&tickets[16 + ph], lacc_[ph] >> 8); atomicAdd(&tickets[24 + ph], lcnt_[ph]); } }
#endif
}

uint32_t exp_flags_lazy() { return (uint32_t)(ZWZ_LAZY_EXP); }

hipError_t configure_lazy_kernels() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(lz_lazy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLazyLdsBytes);
}

hipError_t launch_lazy(const DeflateArgs& a, hipStream_t s) {
    const uint32_t cus = a.cu_count ? a.cu_count : 256u;
    const uint32_t G = a.n < 2u * cus ? a.n : 2u * cus;
    hipLaunchKernelGGL(lz_lazy_kernel, dim3(G), dim3(kLazyThreads), kLazyLdsBytes, s, a.in, a.in_off, a.in_len, a.dense_list, a.tickets, a.sorted,
                       reinterpret_cast<uint32_t*>(a.entries), a.sym, a.mst, a.info, a.links);
#if ZWZ_LAZY_EXP & 16
    if (getenv("ZWZ_LAZY_TIMES")) {
        uint32_t h[64];
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, a.tickets, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "ZWZ_LAZY_TIMES n=%u (cycles >> 8 summed over waves) copy=%u service=%u walk=%u tail=%u wait_wg=%u phase2=%u ticket=%u | trips=%u live_lanes=%u walking_lanes=%u filter_rounds=%u\n",
                    a.n, h[16], h[17], h[18], h[19], h[20], h[21], h[23], h[24], h[25], h[26], h[28]);
    }
#endif
    return hipGetLastError();
}

}  // namespace zwz
