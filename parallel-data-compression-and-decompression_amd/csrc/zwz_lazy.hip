// zwz_lazy.hip -- lz_lazy: match search AND lazy parse of a chain-heavy chunk in one kernel, the searches on demand.
// Replaces, for those chunks, lz_sort + lz_place + lz_match_band + lz_parse (consumer()'s deflate(Z_FINISH),
// compression.cpp:119-131: zlib's deflate_slow + longest_match).  The algorithm, its proof and its CPU form are
// csrc/lz_lazy.h; tests/emu runs that form, lanes in random order, against lz_core.h's records + table walk.
//
// Why: the band computes both records of every position (38.6 candidates a position on the text corpus); zlib's parse hands
// longest_match a quarter of the positions, 5.3 candidates a position (tools/exp/searched_set.py).  The parse is sequential, so
// the chunk is cut into segments of 32 positions, a LANE each, every lane starting on the assumption that nothing is pending
// at its segment's head; a lane runs on past its segment until it stands, fresh, on a position the segment's owner has
// searched fresh too (csrc/lz_lazy.h).  On the text corpus that costs a quarter more searches than one sequential parse.
//
// Everything a search touches is in LDS: the chunk goes through lz_match's ring (csrc/zwz_kernels.h: bytes and chain links of
// the 16 Ki positions of a tile and of the 32 506 before them, 147 KB) tile by tile; a lane whose parse reaches the end of the
// tile waits for the next one and goes on.  (A first form searched the positions sorted by (bucket, position) out of global
// memory -- dest[p], its bucket's bounds, eight candidates a 16-byte gather: bit-exact, and bound by the gathers: 77 GB
// fetched per 10 000 text chunks, 7.7 MB a chunk for 320 KB of arrays.  In history.)
//
//   per tile   bytes + links -> the ring (the next tile's are asked for first); F, G cleared
//              every lane takes segments off a counter: zlib's loop.  longest_match is lz_match's chain walk (one asm loop for
//              the wave: filter word and link of a candidate in one LDS round trip, hits parked and compared together), started
//              from prev_length, over 32 candidates once prev_length >= 8.  A lane whose search has ended is served -- the
//              lazy decision, the next position's first candidate, a fresh search's mark or its meeting an owner's -- when a
//              dozen lanes wait.  What "fresh at q" led to goes into step[q] (global, 4 bytes a fresh search); F | G go out as
//              the chunk's marks
//   at the end the chain of true segments (lazy_resolve, by pointer jumping), every true segment's piece replayed over the
//              marks (lazy_emit_piece): match starts and covered positions into the sym / mst masks (LDS), then once more
//              with the matches' places in the stream known, for the chosen records -- exactly lz_parse's outputs
//
// One workgroup of eight waves a CU (the ring); no global load in the loop.
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
// (ZWZ_LAZY_EXP & 16, experiment builds: lane 0 of every wave adds cycles >> 8 per phase and trip statistics to tickets[16 ..]; launch_lazy prints them
// when ZWZ_LAZY_TIMES is set)
#if ZWZ_LAZY_EXP & 16
#define ZWZ_LSTAMP(ph) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); lacc_[(ph)] += (uint32_t)(now_ - lstamp_); lstamp_ = now_; } while (0)
#define ZWZ_LCOUNT(i, v) do { lcnt_[(i)] += (v); } while (0)
#else
#define ZWZ_LSTAMP(ph) do { } while (0)
#define ZWZ_LCOUNT(i, v) do { } while (0)
#endif

constexpr uint32_t kLazySegs = 65536 / kLazySeg;                              // 2048 segments a chunk
constexpr uint32_t kLazyTileSegs = kTile / kLazySeg;                          // 512 a tile
// LDS: lz_match's rings, then the tile's marks, then the chunk's segment ends
constexpr uint32_t kLazyOffF = kMatchDataBytes + kMatchLinkBytes, kLazyOffG = kLazyOffF + kTile / 8, kLazyOffTerm = kLazyOffG + kTile / 8;
static_assert(kLazyOffTerm + kLazySegs * 2 == kLazyLdsBytes && kLazyOffF % 16 == 0, "lz_lazy LDS layout");
// the end phase reuses the rings: marks, masks, the chain
constexpr uint32_t kLazyOffU = 0, kLazyOffSym = 8192, kLazyOffMst = 16384, kLazyOffNext = 24576, kLazyOffJmpA = kLazyOffNext + 4096,
                   kLazyOffJmpB = kLazyOffJmpA + 4096, kLazyOffMerge = kLazyOffJmpB + 4096, kLazyOffTrue = kLazyOffMerge + 4096,
                   kLazyOffEnd = kLazyOffTrue + 2048;
static_assert(kLazyOffEnd <= kLazyOffF, "lz_lazy end-phase layout");
constexpr uint32_t kLazyServe = 12;                                           // lanes that wait before the service code runs for them

enum : uint32_t { kLzNeed = 0, kLzWalk = 1, kLzIdle = 2 };

static __device__ __forceinline__ uint32_t lazy_ring(uint32_t x) { return min(x, x - kMatchRing); }   // lz_match's ring index (x < 2 * kMatchRing)

__global__ __launch_bounds__(kLazyThreads) void lz_lazy_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                               const uint32_t* __restrict__ in_len, const uint32_t* __restrict__ list,
                                                               uint32_t* __restrict__ tickets, const uint16_t* links_in,
                                                               uint32_t* __restrict__ scratch /* the chunks' entries space */,
                                                               uint64_t* __restrict__ sym, uint64_t* __restrict__ mst,
                                                               ChunkInfo* __restrict__ info, uint16_t* links) {
    typedef __attribute__((address_space(3))) uint8_t* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* sdata = smem;
    uint16_t* slink = reinterpret_cast<uint16_t*>(smem + kMatchDataBytes);
    uint32_t* sF = reinterpret_cast<uint32_t*>(smem + kLazyOffF);
    uint32_t* sG = reinterpret_cast<uint32_t*>(smem + kLazyOffG);
    uint16_t* s_term = reinterpret_cast<uint16_t*>(smem + kLazyOffTerm);
    __shared__ uint32_t s_chunk, s_nsym, s_nextseg, s_wave[kLazyThreads / 64];
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_list = tickets[kTicketDenseCount];
    uint4* sd4 = reinterpret_cast<uint4*>(sdata);
    uint4* sl4 = reinterpret_cast<uint4*>(slink);
    const uint32_t data_a = (uint32_t)(uintptr_t)(lds_ptr)sdata;                                    // LDS byte addresses, for the walk's asm
    const uint32_t lbias = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr) reinterpret_cast<uint8_t*>(slink));
#if ZWZ_LAZY_EXP & 16
    uint64_t lstamp_ = __builtin_amdgcn_s_memtime();
    uint32_t lacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lcnt_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        ZWZ_LSTAMP(7);
        __syncthreads();
        if (tid == 0) { s_chunk = atomicAdd(&tickets[kTicketLazyNext], 1u); s_nsym = 0; }
        __syncthreads();
        const uint32_t t_ = s_chunk;
        if (t_ >= n_list) break;
        const uint32_t chunk = list[t_];
        const uint32_t L = in_len[chunk];
        uint32_t* scr = scratch + (size_t)chunk * kLazyScratchWords;
        uint32_t* stepw = scr + kLazyStepOff;
        uint32_t* gU = scr + kLazyMarkOff;
        const uint4* gd4 = reinterpret_cast<const uint4*>(in + in_off[chunk]);                     // 16-byte aligned (API contract)
        const uint4* gl4 = reinterpret_cast<const uint4*>(links_in + (size_t)chunk * kLinkStride);
        const uint32_t ntiles = (L + kTile - 1) / kTile;
        const uint32_t dvec_total = (L + 15u) >> 4;
        // vectors [dlo, dhi) of data and [llo, lhi) of links are what tile t adds to the window (lz_match's staging)
#define ZWZ_LTILE_RANGE(t)                                                                         \
    const uint32_t te_ = min(((t) + 1) * kTile, L), pe_ = (t) ? min((t) * kTile, L) : 0u;          \
    const uint32_t dlo = (t) ? min((pe_ + kMaxMatch + 8u + 15u) >> 4, dvec_total) : 0u;            \
    const uint32_t dhi = min((te_ + kMaxMatch + 8u + 15u) >> 4, dvec_total);                       \
    const uint32_t llo = (pe_ + 7u) >> 3, lhi = (te_ + 7u) >> 3;
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        uint4 pd[3] = {z4, z4, z4}, pl[4] = {z4, z4, z4, z4};      // the next tile's bytes / links in flight (<= 1 041 + 2 048 vectors over 512 threads)
#define ZWZ_LPREFETCH(t)                                                                           \
    {                                                                                              \
        ZWZ_LTILE_RANGE(t)                                                                         \
        _Pragma("unroll") for (uint32_t r = 0; r < 3; r++) if (dlo + tid + r * kLazyThreads < dhi) pd[r] = gd4[dlo + tid + r * kLazyThreads]; \
        _Pragma("unroll") for (uint32_t r = 0; r < 4; r++) if (llo + tid + r * kLazyThreads < lhi) pl[r] = gl4[llo + tid + r * kLazyThreads]; \
    }
        for (uint32_t i = tid; i < kLazySegs; i += kLazyThreads) s_term[i] = 0xffffu;
        ZWZ_LPREFETCH(0u)

        // a lane's parse, kept across tiles: where it stands (p), the pending match (b, bpos), the chain's head (q0), its segment
        uint32_t p = 0, b = kMinMatch - 1u, bpos = 0, q0 = 0, seg = 0, own_end = 0;
        bool held = false;                                          // the parse reached the tile's end and goes on in the next
        for (uint32_t t = 0; t < ntiles; t++) {
            const uint32_t ts = t * kTile, te = min(ts + kTile, L);
            {   // registers -> the rings
                ZWZ_LTILE_RANGE(t)
                auto put_d = [&](uint32_t v, const uint4& x) {
                    const uint32_t rv = min(v, v - kMatchRing / 16u);
                    sd4[rv] = x;
                    if (rv < kMatchMirror / 16u) sd4[kMatchRing / 16u + rv] = x;
                };
#pragma unroll
                for (uint32_t r = 0; r < 3; r++) if (dlo + tid + r * kLazyThreads < dhi) put_d(dlo + tid + r * kLazyThreads, pd[r]);
#pragma unroll
                for (uint32_t r = 0; r < 4; r++) if (llo + tid + r * kLazyThreads < lhi) { const uint32_t u = llo + tid + r * kLazyThreads; sl4[min(u, u - kMatchRing / 8u)] = pl[r]; }
            }
            for (uint32_t i = tid; i < kTile / 32u; i += kLazyThreads) { sF[i] = 0; sG[i] = 0; }
            if (tid == 0) s_nextseg = 0;
            __syncthreads();
            ZWZ_LSTAMP(0);
            if (t + 1 < ntiles) ZWZ_LPREFETCH(t + 1)
            const uint32_t nseg = (te - ts + kLazySeg - 1u) / kLazySeg;

            // ---- the lanes ----
            uint32_t st = kLzNeed;
            bool have = false;                                      // a search has just ended: its result waits for deflate_slow's decision
            bool fresh_start = !held;                               // no parse in hand: take a segment
            held = false;
            // the search in hand
            uint32_t cur = 0, nxt = 0, n_l = 0, bound = 0, best = 0, best_pos = 0, f_mask = 0, scan_w = 0, dbias = data_a, limit = 0, max_len = 0, nice = 0, pp = 0;
            uint64_t walk = 0;
            for (;;) {
                // -- service: the lanes whose search has ended (or that have none yet), a dozen at a time --
                const uint64_t need = __builtin_amdgcn_ballot_w64(st == kLzNeed);
                if (need != 0ull && (walk == 0ull || (uint32_t)__popcll(need) >= kLazyServe)) {
                    ZWZ_LCOUNT(0, 1u); ZWZ_LCOUNT(1, (uint32_t)__popcll(need));
                    bool start = false;
                    if (st == kLzNeed) {
                        if (have) {                                 // deflate_slow's decision
                            have = false;
                            bool improved = best > b;
                            if (best == kMinMatch && b < kMinMatch && p - best_pos > kTooFar) improved = false;
                            if (improved) { b = best; bpos = best_pos; p++; }
                            else if (b >= kMinMatch) { stepw[q0] = lazy_step_pack(p - 1u - q0, b, p - 1u - bpos); p = p - 1u + b; b = kMinMatch - 1u; }
                            else { stepw[q0] = 0u; p++; }
                        }
                        for (;;) {                                  // on to the next search
                            if (fresh_start) {                      // a segment off the counter
                                const uint32_t s_ = atomicAdd(&s_nextseg, 1u);
                                if (s_ >= nseg) { st = kLzIdle; break; }
                                seg = t * kLazyTileSegs + s_; p = ts + s_ * kLazySeg; own_end = p + kLazySeg; b = kMinMatch - 1u;
                                fresh_start = false;
                            }
                            if (b >= kMinMatch && (b >= kMaxLazy || p >= L)) {          // no search behind a match of max_lazy bytes, none at the end of the data
                                stepw[q0] = lazy_step_pack(p - 1u - q0, b, p - 1u - bpos); p = p - 1u + b; b = kMinMatch - 1u;
                                continue;
                            }
                            if (p >= L) { s_term[seg] = (uint16_t)L; fresh_start = true; continue; }   // the parse is complete
                            if (p >= te) { held = true; st = kLzIdle; break; }                       // to be continued in the next tile
                            pp = lazy_ring(p);
                            const uint32_t c1 = slink[pp];
                            const bool ok = p + kMinMatch <= L && c1 != 0u && p - c1 <= kMaxDist && !(p >= kSlidePos && c1 <= kWSize);
                            if (!ok) {                              // nothing to search: the pending match goes out, or a literal
                                if (b >= kMinMatch) { stepw[q0] = lazy_step_pack(p - 1u - q0, b, p - 1u - bpos); p = p - 1u + b; b = kMinMatch - 1u; }
                                else p++;
                                continue;
                            }
                            if (b < kMinMatch) {                    // a fresh search: mark it, or meet the owner's mark
                                const uint32_t bit = 1u << (p & 31u), w = (p - ts) >> 5;
                                if (p < own_end) atomicOr(&sF[w], bit);
                                else if (sF[w] & bit) { s_term[seg] = (uint16_t)p; fresh_start = true; continue; }
                                else atomicOr(&sG[w], bit);
                                q0 = p;
                            }
                            const uint32_t lookahead = L - p;
                            max_len = lookahead < kMaxMatch ? lookahead : kMaxMatch;
                            nice = lookahead < kNiceLen ? lookahead : kNiceLen;
                            limit = p > kMaxDist ? p - kMaxDist : 0u;
                            bound = b >= kGoodLen ? kShortChain : kMaxChain;
                            best = b; best_pos = 0; n_l = 0; cur = c1; nxt = c1;
                            const uint32_t f_off = b >= kMinMatch ? b - 3u : 0u;
                            f_mask = b >= kMinMatch ? 0xffffffffu : 0xffffffu;
                            scan_w = load_u32(sdata, pp + f_off) & f_mask;
                            dbias = data_a + f_off;
                            st = kLzWalk; start = true;
                            break;
                        }
                    }
                    walk |= __builtin_amdgcn_ballot_w64(start);
                }
                if (walk == 0ull) {
                    if (__builtin_amdgcn_ballot_w64(st == kLzNeed) == 0ull) break;          // every lane idle: the tile is done
                    continue;
                }
                ZWZ_LSTAMP(1);
                // -- the walk (lz_match's loop, csrc/zwz_kernels.hip lz_search_wave): EXEC = the walking lanes; a lane whose filter word
                //    matches is parked, one whose chain or budget ends is finished; out when a dozen lanes wait or nobody walks --
                uint64_t park = 0, fin = 0;
                {
                    uint64_t sv, cont, tmp; uint32_t ta, tb, w0, tl, cnt;
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n\t"
                        "1:\n\t"
                        "s_mov_b64 exec, %[walk]\n\t"
                        "v_subrev_u32 %[a], %[ring], %[cur]\n\t"            /* the candidate's ring index: min(cur, cur - ring) */
                        "v_min_u32 %[l], %[cur], %[a]\n\t"
                        "v_add_u32 %[a], %[l], %[dbias]\n\t"
                        "v_and_b32 %[b], -4, %[a]\n\t"
                        "ds_read2_b32 v[90:91], %[b] offset1:1\n\t"
                        "v_lshl_add_u32 %[l], %[l], 1, %[lbias]\n\t"
                        "ds_read_u16 %[nxt], %[l]\n\t"
                        "v_add_u32 %[nl], 1, %[nl]\n\t"
                        "s_waitcnt lgkmcnt(1)\n\t"
                        "v_alignbyte_b32 %[w0], v91, v90, %[a]\n\t"
                        "v_and_b32 %[w0], %[w0], %[fmask]\n\t"
                        "v_cmp_eq_u32 vcc, %[w0], %[scan]\n\t"                 /* within EXEC: the walking lanes that hit */
                        "s_or_b64 %[park], %[park], vcc\n\t"
                        "s_andn2_b64 %[walk], %[walk], vcc\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_cmp_gt_u32 vcc, %[nxt], %[limit]\n\t"               /* the chain goes on ... */
                        "v_cmp_gt_u32 %[cont], %[bound], %[nl]\n\t"            /* ... and the lane may follow it */
                        "s_and_b64 %[cont], %[cont], vcc\n\t"
                        "s_andn2_b64 %[tmp], %[walk], %[cont]\n\t"             /* walking, no hit, nowhere to go: the search is over */
                        "s_or_b64 %[fin], %[fin], %[tmp]\n\t"
                        "s_and_b64 %[walk], %[walk], %[cont]\n\t"
                        "v_cndmask_b32 %[cur], %[cur], %[nxt], %[walk]\n\t"    /* walking lanes step on; parked ones keep their candidate */
                        "s_or_b64 %[tmp], %[park], %[fin]\n\t"
                        "s_bcnt1_i32_b64 %[cnt], %[tmp]\n\t"
                        "s_cmp_ge_u32 %[cnt], %[npark]\n\t"
                        "s_cbranch_scc1 2f\n\t"
                        "s_cmp_lg_u64 %[walk], 0\n\t"
                        "s_cbranch_scc1 1b\n\t"
                        "2:\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [cur] "+v"(cur), [nxt] "+v"(nxt), [nl] "+v"(n_l), [walk] "+s"(walk), [park] "+s"(park), [fin] "+s"(fin), [sv] "=&s"(sv),
                          [cont] "=&s"(cont), [tmp] "=&s"(tmp), [cnt] "=&s"(cnt), [a] "=&v"(ta), [b] "=&v"(tb), [w0] "=&v"(w0), [l] "=&v"(tl)
                        : [dbias] "v"(dbias), [lbias] "s"(lbias), [fmask] "v"(f_mask), [scan] "v"(scan_w), [limit] "v"(limit), [bound] "v"(bound),
                          [npark] "s"(kLazyServe), [ring] "s"(kMatchRing)
                        : "vcc", "scc", "memory", "v90", "v91");
                }
                ZWZ_LSTAMP(2);
                ZWZ_LCOUNT(2, 1u);
                // -- the parked lanes: the full comparison; a longer match moves the filter on --
                bool resume = false, over = (fin >> lane) & 1ull;
                if ((park >> lane) & 1ull) {                       // cur = the candidate whose filter word matched, nxt = its link, n_l counts it
                    const uint32_t c_ = lazy_ring(cur);
                    const uint32_t len = match_len_from(sdata, c_, pp, 0u, max_len);
                    resume = nxt > limit && n_l < bound;
                    if (len > best) {
                        best = len; best_pos = cur;
                        if (len >= nice) resume = false;
                        else { const uint32_t f_off = best - 3u; f_mask = 0xffffffffu; scan_w = load_u32(sdata, pp + f_off); dbias = data_a + f_off; }
                    }
                    cur = nxt;
                    over = !resume;
                }
                if (over) { st = kLzNeed; have = true; }
                walk |= __builtin_amdgcn_ballot_w64(resume);
                ZWZ_LSTAMP(3);
            }
            ZWZ_LSTAMP(4);
            __syncthreads();                                        // every lane is idle: the tile's marks are complete
            for (uint32_t i = tid; i < kTile / 32u; i += kLazyThreads) gU[t * (kTile / 32u) + i] = sF[i] | sG[i];
            ZWZ_LSTAMP(5);
        }
#undef ZWZ_LTILE_RANGE
#undef ZWZ_LPREFETCH
        __threadfence();                                            // the steps and marks are read back below, by other lanes
        __syncthreads();

        // ---- the end phase: the rings are dead ----
        uint32_t* s_U = reinterpret_cast<uint32_t*>(smem + kLazyOffU);
        uint32_t* s_sym = reinterpret_cast<uint32_t*>(smem + kLazyOffSym);
        uint32_t* s_mst = reinterpret_cast<uint32_t*>(smem + kLazyOffMst);
        uint16_t* s_next = reinterpret_cast<uint16_t*>(smem + kLazyOffNext);
        uint16_t* s_jmpa = reinterpret_cast<uint16_t*>(smem + kLazyOffJmpA);
        uint16_t* s_jmpb = reinterpret_cast<uint16_t*>(smem + kLazyOffJmpB);
        uint16_t* s_merge = reinterpret_cast<uint16_t*>(smem + kLazyOffMerge);
        uint8_t* s_true = smem + kLazyOffTrue;
        const uint32_t nsegs = (L + kLazySeg - 1u) / kLazySeg, nmw = (L + 31u) >> 5;
        for (uint32_t i = tid; i < 2048u; i += kLazyThreads) {
            const uint32_t lo = i << 5;
            s_U[i] = i < nmw ? __builtin_nontemporal_load(&gU[i]) : 0u;
            s_sym[i] = lo + 32u <= L ? 0xffffffffu : lo < L ? (1u << (L - lo)) - 1u : 0u;
            s_mst[i] = 0u;
        }
        // the chain of true segments: segment i's path ends at term[i], in segment next[i], whose path is true from there; marks spread
        // from segment 0 by pointer jumping (a segment's end lies beyond it: eleven doublings reach everything)
        for (uint32_t i = tid; i < kLazySegs; i += kLazyThreads) {
            const uint32_t tm = s_term[i];
            const uint16_t nxv = (uint16_t)((i < nsegs && tm < L) ? tm / kLazySeg : 0xffffu);
            s_next[i] = nxv; s_jmpa[i] = nxv; s_true[i] = i == 0u ? 1u : 0u; s_merge[i] = 0xffffu;
        }
        __syncthreads();
        for (uint32_t r = 0; r < 11u; r++) {
            const uint16_t* ja = (r & 1u) ? s_jmpb : s_jmpa; uint16_t* jb = (r & 1u) ? s_jmpa : s_jmpb;
            for (uint32_t i = tid; i < kLazySegs; i += kLazyThreads) { const uint32_t j = ja[i]; if (s_true[i] && j != 0xffffu) s_true[j] = 1u; }
            __syncthreads();
            for (uint32_t i = tid; i < kLazySegs; i += kLazyThreads) { const uint32_t j = ja[i]; jb[i] = j != 0xffffu ? ja[j] : (uint16_t)0xffffu; }
            __syncthreads();
        }
        for (uint32_t i = tid; i < kLazySegs; i += kLazyThreads) if (s_true[i] && s_next[i] != 0xffffu) s_merge[s_next[i]] = s_term[i];
        if (tid == 0) s_merge[0] = 0;
        __syncthreads();
        // replay of a true segment's piece [from, to) over the marks, four marks' steps asked for together
        uint32_t* chosen = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(links + (size_t)chunk * kLinkStride) + kChosenOffset);
        auto replay = [&](uint32_t from, uint32_t to, bool second, uint32_t base) -> uint32_t {
            uint32_t nf = from, nm = 0;
            for (uint32_t w = from >> 5; (w << 5) < to; w++) {
                uint32_t bits = s_U[w];
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
        constexpr uint32_t kPer = kLazySegs / kLazyThreads;          // consecutive segments a thread replays
        uint32_t my[kPer], mine = 0;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint32_t k = tid * kPer + j;
            const uint32_t from = s_merge[k], to = s_term[k];
            my[j] = (k < nsegs && s_true[k] && from != 0xffffu) ? replay(from, min(to, L), false, 0u) : 0u;
            mine += my[j];
        }
        const uint32_t incl = wave_scan_incl(mine);
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        uint32_t base = incl - mine;
        for (uint32_t i = 0; i < kLazyThreads / 64u; i++) base += i < wave ? s_wave[i] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint32_t k = tid * kPer + j;
            if (my[j]) replay(s_merge[k], min((uint32_t)s_term[k], L), true, base);
            base += my[j];
        }
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
        ZWZ_LSTAMP(6);
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
    const uint32_t G = a.n < cus ? a.n : cus;
    hipLaunchKernelGGL(lz_lazy_kernel, dim3(G), dim3(kLazyThreads), kLazyLdsBytes, s, a.in, a.in_off, a.in_len, a.dense_list, a.tickets, a.links,
                       reinterpret_cast<uint32_t*>(a.entries), a.sym, a.mst, a.info, a.links);
#if ZWZ_LAZY_EXP & 16
    if (getenv("ZWZ_LAZY_TIMES")) {
        uint32_t h[64];
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, a.tickets, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "ZWZ_LAZY_TIMES n=%u (cycles >> 8 summed over waves) stage=%u service=%u walk=%u compare=%u tile_tail=%u flush=%u end=%u ticket=%u | services=%u served_lanes=%u walks=%u\n",
                    a.n, h[16], h[17], h[18], h[19], h[20], h[21], h[22], h[23], h[24], h[25], h[26]);
    }
#endif
    return hipGetLastError();
}

}  // namespace zwz
