// zwz_band.hip -- match records of chain-heavy chunks (text: four positions in five have a chain predecessor) without a
// chain walk.  Replaces, for those chunks, what lz_match's walk did with two dependent LDS gathers per candidate
// (consumer()'s longest_match, compression.cpp:119-131); the arithmetic and its proof of equivalence are csrc/lz_band.h.
//
//   lz_dense_list  one wave / chunk     a sample of the chunk's trigrams says "chain-heavy" or not -> two lists
//   lz_sort        one WG / chunk       positions sorted by (hash bucket, position): histogram, scan, and a ranking pass by
//                                       ONE wave -- ds_add_rtn serves same-address lanes in lane order, the property lz_links
//                                       already stands on -- whose results go out as (bucket << 16 | position) words
//   lz_match_band  one WG / chunk       the chunk's bytes in LDS; the sorted array streams through in tiles; per tile:
//                                       eight comparison bytes beside every entry, candidates per entry, then the banded
//                                       first pass (lane = entry, step k = "the entry k places in front": consecutive
//                                       lanes read consecutive LDS words -- no gather, no conflict) and the sharers' pass
// Records and has128 bits come out exactly as lz_match writes them; lz_parse does not know the difference.
#include <hip/hip_runtime.h>

#include "lz_band.h"
#include "zwz_kernels.h"
#include "zwz_device.h"

#ifndef ZWZ_BAND_EXP
#define ZWZ_BAND_EXP 0     // timing experiments only (tools/gpu.sh times band ... <bits>; the records are then not zlib's): 1 = no second pass, 2 = no band loop, 4 = no record stores
#endif

// (ZWZ_BAND_EXP & 16: thread 0 of every workgroup adds the cycles it spent per phase, >> 8, to tickets[16 + phase]; launch_deflate
// prints them when ZWZ_BAND_TIMES is set)
#if ZWZ_BAND_EXP & 16
#define ZWZ_STAMP(ph) do { if (tid == 0) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&tickets[16 + (ph)], (uint32_t)((now_ - stamp_) >> 8)); stamp_ = now_; } } while (0)
#else
#define ZWZ_STAMP(ph) do { } while (0)
#endif

namespace zwz {

// ------------------------------------------------------------------------------------------------
// lz_dense_list: a wave per chunk hashes the chunk's first kDenseSample trigrams into two 32 Ki-bit sets (LDS) -- bucket taken, bucket
// taken twice -- and counts the ones that are at least the second resp. the third of their bucket (zwz_kernels.h: sample_is_dense).
// Chain-heavy chunks get kDenseMark in link_stat[], the others 0 (for lz_links to count in); lz_lists turns the marks into the two lists.
constexpr uint32_t kDenseThreads = 256;
__global__ __launch_bounds__(kDenseThreads) void lz_dense_list_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                                      const uint32_t* __restrict__ in_len, uint32_t n, uint32_t* __restrict__ link_stat,
                                                                      uint32_t force /* 2: every chunk is chain-heavy */) {
    __shared__ uint32_t s_set[kDenseThreads / 64][2048];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    uint32_t* set = s_set[wave];
    // persistent: a wave takes chunks wave-id, + all waves, ... (a workgroup per four chunks spent more time being dispatched than working)
    for (uint32_t c = blockIdx.x * (kDenseThreads / 64u) + wave; c < n; c += gridDim.x * (kDenseThreads / 64u)) {
        const uint32_t L = in_len[c];
        bool dense = force == 2u && L != 0u;
        if (!dense && L >= kMinMatch) {
            for (uint32_t i = lane; i < 2048u; i += 64u) set[i] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t* d32 = reinterpret_cast<const uint32_t*>(in + in_off[c]);       // 16-byte aligned; readable to L rounded up to 16
            const uint32_t sampled = min(L - (kMinMatch - 1u), kDenseSample), nd = ((L + 15u) & ~15u) >> 2;
            uint32_t repeats = 0, thirds = 0;
            static_assert(kDenseSample == 2048, "eight trips of 256 positions, their loads asked for together");
            uint32_t w0[8], w1[8];                                                      // four positions a lane a trip: a dword and its successor
#pragma unroll
            for (uint32_t t = 0; t < 8; t++) {
                const uint32_t i = 64u * t + lane;
                w0[t] = 256u * t < sampled && i < nd ? d32[i] : 0u; w1[t] = 256u * t < sampled && i + 1u < nd ? d32[i + 1u] : 0u;
            }
#pragma unroll
            for (uint32_t t = 0; t < 8; t++) {
                if (256u * t >= sampled) break;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t x = j ? __builtin_amdgcn_alignbyte(w1[t], w0[t], j) : w0[t];
                    const uint32_t h = hash3(x & 0xffu, (x >> 8) & 0xffu, (x >> 16) & 0xffu);
                    const bool live = 256u * t + 4u * lane + j < sampled;
                    uint32_t old = 0, old2 = 0;
                    if (live) old = atomicOr(&set[h >> 5], 1u << (h & 31u));            // (same-word lanes are served one after the other: each sees the earlier ones' bits)
                    const bool again = live && ((old >> (h & 31u)) & 1u);
                    if (again) old2 = atomicOr(&set[1024u + (h >> 5)], 1u << (h & 31u));
                    repeats += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(again));
                    thirds += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(again && ((old2 >> (h & 31u)) & 1u)));
                }
            }
            dense = sample_is_dense(repeats, thirds, sampled, L);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        if (lane == 0) link_stat[c] = dense ? kDenseMark : 0u;
    }
}

// The two lists, each in chunk order (one workgroup; appended by atomics their order was arbitrary, and lz_links' workgroups,
// which take list entries w, w + G, ..., ended up with uneven shares of config 2's full and 4-byte chunks).
__global__ __launch_bounds__(1024) void lz_lists_kernel(const uint32_t* __restrict__ in_len, const uint32_t* __restrict__ link_stat, uint32_t n,
                                                        uint32_t* __restrict__ dense_list, uint32_t* __restrict__ sparse_list, uint32_t* __restrict__ tickets) {
    __shared__ uint32_t s_d[16], s_s[16];
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    uint32_t nd = 0, ns = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += 1024u) {
        const uint32_t c = c0 + tid;
        const bool live = c < n && in_len[c] != 0u, dense = live && link_stat[c] == kDenseMark, sparse = live && !dense;
        const uint64_t md = __builtin_amdgcn_ballot_w64(dense), ms = __builtin_amdgcn_ballot_w64(sparse);
        if (lane == 0) { s_d[wave] = (uint32_t)__popcll(md); s_s[wave] = (uint32_t)__popcll(ms); }
        __syncthreads();
        uint32_t bd = nd, bs = ns, td = 0, ts = 0;
        for (uint32_t w = 0; w < 16u; w++) { bd += w < wave ? s_d[w] : 0u; bs += w < wave ? s_s[w] : 0u; td += s_d[w]; ts += s_s[w]; }
        if (dense) dense_list[bd + rank_in(md)] = c;
        if (sparse) sparse_list[bs + rank_in(ms)] = c;
        nd += td; ns += ts;
        __syncthreads();
    }
    if (tid == 0) { tickets[kTicketDenseCount] = nd; tickets[kTicketSparseCount] = ns; }
}

// ------------------------------------------------------------------------------------------------
// lz_sort.  The table holds two 16-bit counters a dword (bucket h in half h & 1 of dword h >> 1): 64 KiB, two workgroups a CU.
// A counter never carries into its neighbour: it ends at bucket start + bucket size <= 65 533.
// returning adds the ranking wave keeps in flight (a trip's worth): the step is the LDS's LATENCY for a returning atomic divided by what is in flight, and the
// kernel's two workgroups a CU leave a wave 256 registers -- 4: 7.81 ms for the links stage on text, 8 (round 3): 7.33, 16: 7.00, 24: 6.93
#ifndef ZWZ_SORT_DEPTH
#define ZWZ_SORT_DEPTH 16
#endif
constexpr uint32_t kSortDepth = ZWZ_SORT_DEPTH;
__global__ __launch_bounds__(kSortThreads) void lz_sort_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                               const uint32_t* __restrict__ in_len, const uint32_t* __restrict__ list,
                                                               uint32_t* __restrict__ tickets, uint32_t* __restrict__ sorted,
                                                               uint16_t* __restrict__ hbuf /* the chunks' dead link arrays */,
                                                               uint32_t* __restrict__ scratch /* the chunks' entries space (lz_lazy's bucket ends), or null */) {
    __shared__ __attribute__((aligned(16))) uint32_t tab[16384];
    __shared__ uint32_t s_wsum[kSortThreads / 64];
    __shared__ uint32_t s_chunk;
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_list = tickets[kTicketDenseCount];
    for (;;) {
        __syncthreads();                                           // (s_chunk and the table are free again)
        if (tid == 0) s_chunk = atomicAdd(&tickets[kTicketSortNext], 1u);
        __syncthreads();
        const uint32_t t = s_chunk;
        if (t >= n_list) break;
        const uint32_t chunk = list[t];
        const uint32_t L = in_len[chunk];
        const uint32_t n = L >= kMinMatch ? L - (kMinMatch - 1u) : 0u;     // positions with a trigram
        if (n == 0) continue;
        const uint32_t* d32 = reinterpret_cast<const uint32_t*>(in + in_off[chunk]);   // 16-byte aligned; readable to L rounded up to 16
        const uint32_t nd = ((L + 15u) & ~15u) >> 2;
        uint16_t* hb = hbuf + (size_t)chunk * kLinkStride;
        uint16_t* out = reinterpret_cast<uint16_t*>(sorted + (size_t)chunk * kSortedStride);   // dest[p]: p's index in (bucket, position) order
        {
            uint4* t4 = reinterpret_cast<uint4*>(tab);
            for (uint32_t i = tid; i < 4096u; i += kSortThreads) t4[i] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        // histogram: a thread takes sixteen positions a trip (a 16-byte vector and the dword behind it), eight trips' loads in flight
        // together (one at a time this loop was sixty-four HBM round trips: most of the kernel); the hashes go out to hbuf for the ranking
        {
            const uint4* d4 = reinterpret_cast<const uint4*>(d32);
            const uint32_t nv = nd >> 2;                                        // 16-byte vectors of the slot
#pragma unroll 1
            for (uint32_t i0 = tid; 16u * i0 < n; i0 += 8u * kSortThreads) {
                uint4 v[8]; uint32_t nx[8];
#pragma unroll
                for (uint32_t r = 0; r < 8; r++) {
                    const uint32_t i = i0 + r * kSortThreads;
                    v[r] = i < nv ? d4[i] : make_uint4(0, 0, 0, 0);
                    nx[r] = i + 1u < nv ? d32[4u * i + 4u] : 0u;
                }
#pragma unroll
                for (uint32_t r = 0; r < 8; r++) {
                    const uint32_t i = i0 + r * kSortThreads;
                    if (16u * i >= n) break;
                    const uint32_t w[5] = {v[r].x, v[r].y, v[r].z, v[r].w, nx[r]};
                    uint32_t h[16];
#pragma unroll
                    for (uint32_t j = 0; j < 16; j++) {
                        const uint32_t x = (j & 3u) ? __builtin_amdgcn_alignbyte(w[(j >> 2) + 1u], w[j >> 2], j & 3u) : w[j >> 2];
                        h[j] = hash3(x & 0xffu, (x >> 8) & 0xffu, (x >> 16) & 0xffu);
                        if (16u * i + j < n) atomicAdd(&tab[h[j] >> 1], (h[j] & 1u) ? 0x10000u : 1u);
                    }
                    // (what goes out is the counter's place as the ranking wave wants it: byte offset of the dword | which half -- a lone
                    // wave pays ~7 cycles an instruction, and this is done here by four)
#pragma unroll
                    for (uint32_t j = 0; j < 16; j++) h[j] = (h[j] >> 1) << 2 | (h[j] & 1u);
                    uint4* hb4 = reinterpret_cast<uint4*>(hb + 16u * i);
                    hb4[0] = make_uint4(h[0] | h[1] << 16, h[2] | h[3] << 16, h[4] | h[5] << 16, h[6] | h[7] << 16);
                    hb4[1] = make_uint4(h[8] | h[9] << 16, h[10] | h[11] << 16, h[12] | h[13] << 16, h[14] | h[15] << 16);
                }
            }
        }
        __syncthreads();
        // exclusive scan of the 32768 counters, in place: a wave owns 4096 dwords, 64 rows of 64
        {
            const uint32_t w_base = wave * 4096u;
            uint32_t tot = 0;
            for (uint32_t r = 0; r < 64u; r++) { const uint32_t v = tab[w_base + r * 64u + lane]; tot += (v & 0xffffu) + (v >> 16); }
            tot = wave_scan_incl(tot);
            if (lane == 63) s_wsum[wave] = tot;
            __syncthreads();
            uint32_t carry = 0;
            for (uint32_t i = 0; i < kSortThreads / 64u; i++) carry += i < wave ? s_wsum[i] : 0u;
            for (uint32_t r = 0; r < 64u; r++) {
                const uint32_t idx = w_base + r * 64u + lane;
                const uint32_t v = tab[idx], lo = v & 0xffffu, hi = v >> 16, own = lo + hi;
                const uint32_t incl = wave_scan_incl(own), ex = carry + incl - own;
                tab[idx] = ex | (ex + lo) << 16;
                carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        __syncthreads();
        // ranking: positions in order, 64 a step, by ONE wave: the returning add hands same-bucket lanes their slots in lane order
        // (= position order), steps follow one another in program order.
        // (Round 5, measured and dropped: ALL FOUR waves ranking, each the positions whose bucket lies in its quarter of the table -- buckets do not
        // care about each other, so it is exact (all oracle tests passed) -- on the idea that the step is latency over what is in flight: links stage
        // 7.1 -> 10.4 ms on text.  Every wave then loads all the hashes, and the per-lane guards around add and store are the exec-mask bookkeeping
        // the unguarded trips below got rid of.  Also measured and dropped: NO histogram -- the ranking's returning add counts from zero and hands a
        // position its rank inside its bucket, the scan comes afterwards, and a last pass of plain LDS reads adds the bucket's start to every rank
        // (on the idea that the LDS's atomics bound the kernel and the histogram is a third of them): exact, 7.1 -> 8.5 ms.  The histogram's adds
        // overlap the other workgroup's ranking; the last pass's two global round trips a position vector do not.)
        if (wave == 0) {
            // Whole trips of eight steps run without a single test (every lane has a position: guarded per lane, the loop was
            // mostly exec-mask bookkeeping); the next trip's hashes are on their way while this one ranks.
            constexpr uint32_t K = kSortDepth;                                      // steps a trip: that many returning adds in flight
            const uint32_t full = n / (64u * K);                                   // trips of 64 K positions
            // (running pointers, so that a trip's eight loads and eight stores are one address and constant offsets; hb holds byte
            // offset | half, see the histogram: per step the wave is left with and + two shifts + the add, and a shift + the store)
            auto bump = [&](uint32_t w) { return atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(tab) + (w & 0xfffcu)), 1u << ((w & 1u) << 4)); };
            const uint16_t* hin = hb + lane;
            uint16_t* o = out + lane;
            uint32_t hv[K];
            auto ask = [&](uint32_t* h8) {
#pragma unroll
                for (uint32_t j = 0; j < K; j++) h8[j] = hin[j * 64u];
                hin += 64u * K;
            };
            if (full) ask(hv);
            uint32_t hp[K], op[K];                                              // the previous trip's entries and what its adds returned
            for (uint32_t t = 0; t < full; t++) {
                uint32_t hn[K], old[K];
                if (t + 1u < full) ask(hn);
#pragma unroll
                for (uint32_t j = 0; j < K; j++) old[j] = bump(hv[j]);
                // (the previous trip's results are written out while this trip's adds are on their way through the LDS)
                if (t) {
#pragma unroll
                    for (uint32_t j = 0; j < K; j++) o[j * 64u] = (uint16_t)(op[j] >> ((hp[j] & 1u) << 4));
                    o += 64u * K;
                }
#pragma unroll
                for (uint32_t j = 0; j < K; j++) { hp[j] = hv[j]; op[j] = old[j]; hv[j] = hn[j]; }
            }
            if (full) {
#pragma unroll
                for (uint32_t j = 0; j < K; j++) o[j * 64u] = (uint16_t)(op[j] >> ((hp[j] & 1u) << 4));
            }
            for (uint32_t p = full * 64u * K + lane; p - lane < n; p += 64u) {  // the ragged end, a step at a time
                if (p < n) {
                    const uint32_t w = hb[p];
                    out[p] = (uint16_t)(bump(w) >> ((w & 1u) << 4));
                }
            }
        }
        // the table now holds every bucket's END in the sorted array (a counter ran from its bucket's start): lz_lazy's bucket bounds
        if (scratch) {
            __syncthreads();
            uint4* o4 = reinterpret_cast<uint4*>(scratch + (size_t)chunk * kLazyScratchWords + kLazyBendOff);
            const uint4* t4 = reinterpret_cast<const uint4*>(tab);
            for (uint32_t i = tid; i < 4096u; i += kSortThreads) o4[i] = t4[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// lz_place: dest[p] -> the sorted array itself, position of sorted index u at spos[u].  The inverse of a permutation is a
// scatter; done through HBM it costs a 64-line store per wave (lz_sort spent 5x its time on it), so the whole chunk's
// array is scattered in LDS (128 KiB of 16-bit positions) and leaves as whole lines.
__global__ __launch_bounds__(kPlaceThreads) void lz_place_kernel(const uint32_t* __restrict__ in_len, const uint32_t* __restrict__ list,
                                                                 uint32_t* __restrict__ tickets, uint32_t* __restrict__ sorted) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t* stage = reinterpret_cast<uint16_t*>(smem);
    __shared__ uint32_t s_chunk;
    const uint32_t tid = threadIdx.x;
    const uint32_t n_list = tickets[kTicketDenseCount];
    for (;;) {
        __syncthreads();
        if (tid == 0) s_chunk = atomicAdd(&tickets[kTicketPlaceNext], 1u);
        __syncthreads();
        const uint32_t t = s_chunk;
        if (t >= n_list) break;
        const uint32_t chunk = list[t];
        const uint32_t L = in_len[chunk];
        const uint32_t n = L >= kMinMatch ? L - (kMinMatch - 1u) : 0u;
        const uint4* d4 = reinterpret_cast<const uint4*>(sorted + (size_t)chunk * kSortedStride);            // dest[p], 16 bits each
        uint4* o4 = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(sorted + (size_t)chunk * kSortedStride) + 65536);   // spos[u]
        uint4 d[8];
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) if (8u * (tid + kPlaceThreads * r) < n) d[r] = d4[tid + kPlaceThreads * r];
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) {
            const uint32_t p0 = 8u * (tid + kPlaceThreads * r);
            const uint32_t dw[4] = {d[r].x, d[r].y, d[r].z, d[r].w};
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) if (p0 + j < n) stage[(dw[j >> 1] >> (16u * (j & 1u))) & 0xffffu] = (uint16_t)(p0 + j);
        }
        __syncthreads();
        for (uint32_t i = tid; 8u * i < n; i += kPlaceThreads) o4[i] = reinterpret_cast<const uint4*>(stage)[i];
    }
}

// ------------------------------------------------------------------------------------------------
// lz_match_band.  Per chunk: bytes -> LDS, dest[] -> registers (64 positions a thread, kept for all tiles).  Per tile of
// kBandTile sorted entries (+ the 128 in front of them):
//   scatter   every thread looks at its 64 dests; the ones inside the tile drop (bucket << 16 | position) into their slot
//   build     the 8-byte comparison word of every entry; "pure" (one trigram per bucket, so the word starts behind the trigram)
//             unless two neighbours of one bucket differ in their trigrams -- then the words are rebuilt from the trigram on
//   count     candidates per entry (binary search over the monotone band_valid, six entries a thread in flight together);
//             the chunk's last positions are finished here, byte by byte
//   order     runs of eight consecutive entries, sorted by their greatest count: a wave's 64 lanes are eight runs of similar
//             length (a trip lasts as long as its longest band: 0.66 of the lanes busy in array order, 0.89 this way), and
//             each run still reads 64 consecutive LDS bytes per step
//   pass 1    the banded keys (see the loop)
//   pass 2    the flagged entries, compacted into full waves, walk their sharers (csrc/lz_band.h band_deep)
constexpr uint32_t kBandArr = kBandTile + kBand;                    // a tile's arrays: 128 halo entries, then the tile's own
constexpr uint32_t kBandDataBytes = 65536 + 64;                     // the chunk + slack for comparisons that run past its end
constexpr uint32_t kBandOffS = kBandDataBytes, kBandOffE = kBandOffS + kBandArr * 4, kBandOffCk = kBandOffE + kBandArr * 8,
                   kBandOffHas = kBandOffCk + kBandTile * 2;
static_assert(kBandOffHas + 8192 == kBandLdsBytes, "lz_match_band LDS layout");
static_assert(kBandTile % 64 == 0 && kBandOffE % 8 == 0 && kBandOffS % 16 == 0 && kBandOffCk % 16 == 0, "lz_match_band LDS alignment");
constexpr uint32_t kBandRuns = kBandTile / 8;

#ifndef ZWZ_BAND_WAVES
#define ZWZ_BAND_WAVES 4
#endif
__global__ __launch_bounds__(kBandThreads, ZWZ_BAND_WAVES) void lz_match_band_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                                    const uint32_t* __restrict__ in_len, const uint32_t* __restrict__ list,
                                                                    uint32_t* __restrict__ tickets, const uint32_t* __restrict__ sorted,
                                                                    uint2* __restrict__ entries, uint64_t* __restrict__ has128) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* sdata = smem;
    uint32_t* S = reinterpret_cast<uint32_t*>(smem + kBandOffS);
    uint2* E = reinterpret_cast<uint2*>(smem + kBandOffE);
    uint16_t* ck = reinterpret_cast<uint16_t*>(smem + kBandOffCk);       // per own entry: candidates | k of the nearest sharer << 8
    uint32_t* hasb = reinterpret_cast<uint32_t*>(smem + kBandOffHas);    // has128 bits of the whole chunk
    uint16_t* flist = reinterpret_cast<uint16_t*>(E + kBand);            // pass 2's work list: over the own entries' words, dead by then
    __shared__ uint32_t s_chunk, s_grp[2], s_nflag, s_nrun;
    __shared__ uint16_t s_halo_link[kBand];                              // the links of a tile's last 128 entries, for the next tile's halo
    __shared__ uint32_t s_bin[132];
    __shared__ uint16_t s_order[kBandRuns];
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_list = tickets[kTicketDenseCount];
    for (uint32_t i = tid; i < 2048u; i += kBandThreads) hasb[i] = 0;
#if ZWZ_BAND_EXP & 16
    uint64_t stamp_ = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        __syncthreads();
        ZWZ_STAMP(7);
        if (tid == 0) s_chunk = atomicAdd(&tickets[kTicketBandNext], 1u);
        __syncthreads();
        const uint32_t tk = s_chunk;
        if (tk >= n_list) break;
        const uint32_t chunk = list[tk];
        const uint32_t L = in_len[chunk];
        const uint32_t n = L >= kMinMatch ? L - (kMinMatch - 1u) : 0u;
        uint2* ent = entries + (size_t)chunk * kEntryStride;
        uint64_t* hm = has128 + (size_t)chunk * kMaskWords;
        const uint4* spos4 = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(sorted + (size_t)chunk * kSortedStride) + 65536);   // lz_place: position of sorted index u
        copy_vec16(reinterpret_cast<uint4*>(sdata), reinterpret_cast<const uint4*>(in + in_off[chunk]), (L + 15u) >> 4);
        ZWZ_STAMP(0);
        bool prev_pure = false;
        // a tile's positions (eight a thread) are asked for one tile ahead: an HBM round trip per tile otherwise stands in the open
        auto spos_of = [&](uint32_t a_) { return 8u * tid < kBandArr && 8u * tid + a_ >= kBand && 8u * tid + a_ - kBand < n ? spos4[(8u * tid + a_ - kBand) >> 3] : make_uint4(0, 0, 0, 0); };
        static_assert(kBandArr <= 8u * kBandThreads, "a thread holds eight of a tile's positions");
        uint4 v_next = spos_of(0u);
        for (uint32_t a = 0; a < n; a += kBandTile) {
            const uint32_t b = min(a + kBandTile, n), m = b - a + kBand;      // array index i <-> sorted index a - 128 + i
            if (a < kBand) for (uint32_t i = tid; i < kBand - a; i += kBandThreads) S[i] = kBandHaloWord;   // (a = 0: the first tile)
            if (tid == 0) { s_grp[0] = 0; s_grp[1] = 0; s_nflag = 0; s_nrun = 0; }
            if (tid < 132u) s_bin[tid] = 0;
            __syncthreads();                                                  // (also: the chunk's bytes are in place)
            // ---- the tile's entries and the 128 in front of them: positions from lz_place's array; bucket AND the eight comparison bytes
            // out of ONE read of the four dwords that hold bytes p .. p + 10 (round 3 read the trigram here and then, in a second phase,
            // the trigram again, the previous entry's, and two unaligned words behind it: ten dword gathers an entry where four do, and
            // both phases were the LDS pipe's -- 12 % of the kernel)
            bool mixed = false;
            {
                const uint32_t pw[4] = {v_next.x, v_next.y, v_next.z, v_next.w};
                if (8u * tid + a >= kBand) {                                    // (before the array's start: halo words, set above)
                    uint32_t prev_w = 0, prev_t = 0;
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) {
                        const uint32_t p = (pw[j >> 1] >> (16u * (j & 1u))) & 0xffffu;
                        const uint32_t* d = reinterpret_cast<const uint32_t*>(sdata) + (p >> 2);
                        const uint32_t w0 = d[0], w1 = d[1], w2 = d[2], w3 = d[3];           // bytes (p & ~3) .. + 15: p + 10 <= (p & ~3) + 13
                        const uint32_t x = __builtin_amdgcn_alignbyte(w1, w0, p & 3u) & 0xffffffu;
                        const uint32_t word = band_word(hash3(x & 0xffu, (x >> 8) & 0xffu, x >> 16), p);
                        const bool up = (p & 3u) != 0u;                                       // bytes p + 3 .. start in w1 (p & 3 >= 1) or in w0
                        const uint32_t ea = up ? w1 : w0, eb = up ? w2 : w1, ec = up ? w3 : w2, sh = (p + 3u) & 3u;
                        if (8u * tid + j < m) {
                            S[8u * tid + j] = word;
                            E[8u * tid + j] = make_uint2(__builtin_amdgcn_alignbyte(eb, ea, sh), __builtin_amdgcn_alignbyte(ec, eb, sh));
                            if (j) mixed |= band_hash(word) == band_hash(prev_w) && x != prev_t;      // one bucket, two trigrams (j = 0: below)
                        }
                        prev_w = word; prev_t = x;
                    }
                }
                v_next = spos_of(a + kBandTile);                                // (a and 128 are multiples of 8)
            }
            __syncthreads();
            ZWZ_STAMP(1);
            // ---- "pure" (every bucket of the tile holds one trigram only: the words start behind the trigram) unless two neighbours of one
            // bucket differ in their trigrams -- then the words are rebuilt from the trigram on.  Neighbours inside a thread's eight were
            // compared above; here every thread looks across the seam in front of its first entry.
            bool pure;
            {
                const uint32_t i = 8u * tid;
                if (i && i < m) {
                    const uint32_t w = S[i], wp = S[i - 1u];
                    const uint32_t tg = load_u32(sdata, band_pos(w)), tp = load_u32(sdata, band_pos(wp));
                    mixed |= w != kBandHaloWord && wp != kBandHaloWord && band_hash(w) == band_hash(wp) && ((tg ^ tp) & 0xffffffu) != 0u;
                }
                pure = !__syncthreads_or((int)mixed);
                if (!pure) {
                    for (uint32_t i2 = tid; i2 < m; i2 += kBandThreads) { const uint32_t q = band_pos(S[i2]); E[i2] = make_uint2(load_u32(sdata, q), load_u32(sdata, q + 4u)); }
                }
            }
            const uint32_t off = pure ? 3u : 0u, deep = pure ? 11u : 8u;
            auto Sf = [&](uint32_t i) { return S[i]; };
            ZWZ_STAMP(2);
            // (Tried in round 4 and dropped: the bucket's start from a scan of head flags -- a max-scan over the wave plus the last head of the
            // two blocks in front -- and ONE gather to check that the farthest candidate so found is in reach, the binary search only for
            // the entries that fail the check.  On the text corpus nearly every wave holds such an entry (a rare trigram's candidates lie
            // further back than MAX_DIST for every position in the chunk's second half), so the search ran anyway, behind the scan and a
            // barrier: this phase 3.93 M -> 5.45 M cycles >> 8 per 10 000 chunks, 6.56 M with the fallback's searches advancing together.)
            // (Round 5, measured and dropped: a thread takes six CONSECUTIVE entries, searches the first and derives each next count from the one before
            // -- an entry behind one of its own bucket has that one as a candidate plus those of its candidates still in reach, so its count is
            // the previous + 1 less what fell out at the far end: one probe there and one more per entry that did, ~23 probes a thread where the
            // searches make 48.  Exact (all oracle tests); this phase 3.67 M -> 4.77 M, text match stage 68.5 -> 70.7 ms: the 23 probes hang on one
            // another, the six searches run side by side -- the phase is the LDS round trip times the length of the chain, not its probes.)
            // ---- count (thread <-> entries kBand + tid + 1024 j: the searches of a thread's entries advance together)
            {
                constexpr uint32_t kOwnPer = (kBandTile + kBandThreads - 1) / kBandThreads;
                uint32_t own[kOwnPer], k[kOwnPer];
#pragma unroll
                for (uint32_t j = 0; j < kOwnPer; j++) { const uint32_t i = kBand + tid + kBandThreads * j; own[j] = i < m ? S[i] : kBandHaloWord; k[j] = 0; }
                auto search_step = [&](uint32_t step) {
                    uint32_t c[kOwnPer];
#pragma unroll
                    for (uint32_t j = 0; j < kOwnPer; j++) { const uint32_t t = k[j] + step; c[j] = S[kBand + tid + kBandThreads * j - (t <= kBand ? t : 0u)]; }
#pragma unroll
                    for (uint32_t j = 0; j < kOwnPer; j++) { const uint32_t t = k[j] + step; if (t <= kBand && band_valid(own[j], c[j])) k[j] = t; }
                };
                search_step(kBand);
                bool open = false;                                              // inside a long bucket every entry has all 128: nothing to search
#pragma unroll
                for (uint32_t j = 0; j < kOwnPer; j++) open |= kBand + tid + kBandThreads * j < m && k[j] != kBand;
                if (__builtin_amdgcn_ballot_w64(open) != 0) {
#pragma unroll
                    for (uint32_t step = kBand / 2u; step >= 1u; step >>= 1) search_step(step);
                }
#pragma unroll
                for (uint32_t j = 0; j < kOwnPer; j++) {
                    const uint32_t i = kBand + tid + kBandThreads * j;
                    if (i < m) {
                        if (k[j] == 0u && band_first_at_max_dist(own[j], S[i - 1u])) k[j] = 1u;
                        ck[i - kBand] = (uint16_t)k[j];
                    }
                }
            }
            __syncthreads();
            ZWZ_STAMP(3);
            // The halo's entries were the previous tile's last: their links (to sharers under THAT tile's word format) carry over
            // when the format is the same; else a walk that reaches the halo starts again the slow way (band_deep).
            const bool halo_links = a == 0u || prev_pure == pure;
            if (tid < kBand) reinterpret_cast<uint16_t*>(S)[2u * tid + 1u] = a != 0u && halo_links ? s_halo_link[tid] : (uint16_t)kBandNoLink;
            prev_pure = pure;
            // ---- order: counting sort of the runs by their greatest count, longest first (runs without candidates drop out)
            const uint32_t n_run = (b - a + 7u) >> 3;
            uint32_t run_key = 0;
            if (tid < n_run) {
                const uint4 c8 = *reinterpret_cast<const uint4*>(ck + 8u * tid);   // (counts behind the tile's end are stale: masked below)
                const uint32_t cw[4] = {c8.x, c8.y, c8.z, c8.w};
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) if (8u * tid + j < b - a) run_key = max(run_key, (cw[j >> 1] >> (16u * (j & 1u))) & 0xffu);
                if (run_key) atomicAdd(&s_bin[128u - run_key], 1u);
            }
            __syncthreads();
            if (wave == 0) {                                                    // exclusive scan of the 128 bins
                const uint32_t v0 = s_bin[lane], v1 = s_bin[64u + lane];
                const uint32_t i0 = wave_scan_incl(v0), t0 = (uint32_t)__builtin_amdgcn_readlane((int)i0, 63);
                const uint32_t i1 = wave_scan_incl(v1);
                s_bin[lane] = i0 - v0; s_bin[64u + lane] = t0 + i1 - v1;
                if (lane == 63) s_nrun = t0 + i1;
            }
            __syncthreads();
            if (tid < n_run && run_key) s_order[atomicAdd(&s_bin[128u - run_key], 1u)] = (uint16_t)tid;
            __syncthreads();
            ZWZ_STAMP(8);
            const uint32_t n_act = s_nrun, n_grp = (n_act + 7u) >> 3;
            const uint32_t none = pure ? kBandKeyNonePure : kBandKeyNoneImpure;
            // ---- first pass
            for (;;) {
                uint32_t g = 0;
                if (lane == 0) g = atomicAdd(&s_grp[0], 1u);
                g = __builtin_amdgcn_readfirstlane(g);
                if (g >= n_grp) break;
                const uint32_t ri = 8u * g + (lane >> 3);
                const uint32_t i = kBand + 8u * (ri < n_act ? (uint32_t)s_order[ri] : 0u) + (lane & 7u);
                const bool active = ri < n_act && i < m;
                const uint32_t w = S[i], p = band_pos(w);                       // (i < kBandArr whatever the lane)
                const uint32_t cntb = active ? (uint32_t)ck[i - kBand] : 0u;
                const uint2 own = E[i];
                // The band, eight candidates a trip, nearest first.  A key costs eight vector instructions (two XORs, two find-first-bits,
                // mask / mask-and-offset, minimum, shift-or with the trip's 129 - k from a scalar register) and two keys join the
                // running maximum by one v_max3 -- picked by hand: the compiler's own choice was twelve a candidate.  Up to the
                // smallest count among the lanes that have candidates nothing needs masking; lanes without any are kept out of
                // it by a running maximum nothing can beat.
                const uint32_t kmax = (ZWZ_BAND_EXP & 2) ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_max_incl(cntb), 63);
                const uint32_t kmin = 128u - (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_max_incl(cntb ? 128u - cntb : 0u), 63);   // (128 if nobody has any)
                uint32_t best = cntb ? none : 0xffffffffu, snap = none;
                const uint2* Ei = E + i;                                        // (i - k >= 0 for every lane: k <= 128 <= i)
                auto key_of = [&](const uint2 c, uint32_t kc) {
                    uint32_t t0, t1, a1, key;
                    asm("v_ffbl_b32 %0, %1" : "=v"(t0) : "v"(own.x ^ c.x));   // 0xffffffff for 0
                    asm("v_ffbl_b32 %0, %1" : "=v"(t1) : "v"(own.y ^ c.y));
                    asm("v_and_or_b32 %0, %1, %2, 32" : "=v"(a1) : "v"(t1), "s"(0x78u));   // 32 + 8 * equal bytes of the upper word; 0x78 for all four
                    const uint32_t m8 = min(t0 & 0x78u, a1);
                    asm("v_lshl_or_b32 %0, %1, 5, %2" : "=v"(key) : "v"(m8), "s"(kc));
                    return key;
                };
                const uint32_t la = L - p;
                const bool tail = active && la < deep;                          // the word reaches past the data
                if (__builtin_amdgcn_ballot_w64(tail) == 0) {
                    uint32_t k0 = 0;
                    for (; k0 + 8u <= kmin; k0 += 8u) {
                        uint2 c[8];
#pragma unroll
                        for (uint32_t j = 0; j < 8; j++) c[j] = Ei[-(int)(k0 + j + 1u)];
#pragma unroll
                        for (uint32_t j = 0; j < 8; j += 2) {
                            const uint32_t ka = key_of(c[j], 128u - k0 - j), kb = key_of(c[j + 1], 127u - k0 - j);
                            asm("v_max3_u32 %0, %0, %1, %2" : "+v"(best) : "v"(ka), "v"(kb));
                        }
                        if (k0 + 8u == kShortChain) snap = best;
                    }
                    for (; k0 < kmax; k0 += 8u) {
                        uint2 c[8];
#pragma unroll
                        for (uint32_t j = 0; j < 8; j++) c[j] = Ei[-(int)(k0 + j + 1u)];
#pragma unroll
                        for (uint32_t j = 0; j < 8; j++) {
                            const uint32_t key = key_of(c[j], 128u - k0 - j);
                            best = max(best, k0 + j + 1u <= cntb ? key : 0u);
                        }
                        if (k0 + 8u == kShortChain) snap = best;
                    }
                } else {                                                        // a handful of groups per chunk: the plain loop, XORs masked to the bytes that exist
                    const uint32_t nb = tail ? band_tail_bytes(pure, la) : 8u, m_lo = band_tail_mask(nb, 0), m_hi = band_tail_mask(nb, 1);
                    for (uint32_t k = 1; k <= kmax; k++) {
                        const uint2 c = Ei[-(int)k];
                        const uint32_t key = band_key_masked(own.x, own.y, c.x, c.y, m_lo, m_hi, k);
                        best = max(best, k <= cntb ? key : 0u);
                        if (k == kShortChain) snap = best;
                    }
                }
                if (cntb == 0u) best = none;
                const uint32_t key32 = cntb > kShortChain ? snap : best;
                uint32_t e128 = 0, e32 = 0, k1 = 0;
                if (best != none) {
                    if (band_key_len(best) == 15u && !tail && !(ZWZ_BAND_EXP & 1)) k1 = band_key_k(best);
                    else e128 = band_record(best, pure, p, band_pos(S[i - band_key_k(best)]), la);
                }
                if (key32 != none && (tail || band_key_len(key32) != 15u)) e32 = band_record(key32, pure, p, band_pos(S[i - band_key_k(key32)]), la);
                if (active) {
                    if (k1) {
                        ck[i - kBand] = (uint16_t)(cntb | k1 << 8);
                        reinterpret_cast<uint16_t*>(S)[2u * i + 1u] = (uint16_t)(i - k1);   // the bucket field has done its work: now the link
                        if (k1 > kShortChain && !(ZWZ_BAND_EXP & 4)) reinterpret_cast<uint32_t*>(ent + p)[1] = e32;   // final; the second pass writes the rest
                        atomicOr(&hasb[p >> 5], 1u << (p & 31u));
                    } else {
                        reinterpret_cast<uint16_t*>(S)[2u * i + 1u] = (uint16_t)kBandNoLink;
                        if (e128) {
                            if (!(ZWZ_BAND_EXP & 4)) ent[p] = make_uint2(e128, e32); else asm volatile("" :: "v"(e128), "v"(e32), "v"(p));
                            atomicOr(&hasb[p >> 5], 1u << (p & 31u));
                        }
                    }
                }
            }
            // entries of runs that were left out (no candidates anywhere in the run) link nowhere either
            for (uint32_t i = kBand + tid; i < m; i += kBandThreads) if ((uint32_t)ck[i - kBand] == 0u) reinterpret_cast<uint16_t*>(S)[2u * i + 1u] = (uint16_t)kBandNoLink;
            __syncthreads();
            ZWZ_STAMP(4);
            // ---- second pass: the flagged entries, gathered into full waves, walk their sharers (csrc/lz_band.h, band_deep).  A hop is
            // ONE LDS round trip: an entry's word holds its position and its link, so the next entry's word and this one's
            // eight bytes behind the compared ones are asked for together, and those eight bytes settle all but the longest matches.
            for (uint32_t i = kBand + tid; i - tid < m; i += kBandThreads) {
                const bool f = i < m && ((uint32_t)ck[i - kBand] >> 8) != 0u;
                const uint64_t fm = __builtin_amdgcn_ballot_w64(f);
                if (fm) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&s_nflag, (uint32_t)__popcll(fm));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (f) flist[base + rank_in(fm)] = (uint16_t)i;
                }
            }
            __syncthreads();
            ZWZ_STAMP(9);
            // (Round 4, measured and dropped: the walks measured first -- the same chase through the links, nothing compared -- and then taken
            // longest first.  Taken 64 at a time in array order a third of the lanes are busy, sorted by length four fifths
            // (tools/exp/band_pass2_skew.py), and nothing the first pass leaves behind predicts the length.  The walk itself fell by 18 %
            // (6.99 M -> 5.75 M cycle units per 10 000 chunks), measuring and sorting cost 2.75 M: a hop's price is its lanes' LDS gathers,
            // not the trip.  Text match stage 70.4 -> 73.7 ms.)
            const uint32_t n_flag = s_nflag;
            for (;;) {
                uint32_t g = 0;
                if (lane == 0) g = atomicAdd(&s_grp[1], 1u);
                g = __builtin_amdgcn_readfirstlane(g);
                if (64u * g >= n_flag) break;
                const bool on = 64u * g + lane < n_flag;
                const uint32_t i = on ? (uint32_t)flist[64u * g + lane] : kBand;
                const uint32_t c = ck[i - kBand], cnt = c & 0xffu, k1 = c >> 8;
                const uint32_t p = band_pos(S[i]), la = L - p;
                const uint32_t max_len = la < kMaxMatch ? la : kMaxMatch, nice = la < kNiceLen ? la : kNiceLen;
                // (sixteen bytes behind the compared ones, not eight: with eight, some lane of the wave fell out into match_len_from's serial loop
                // at nearly every hop -- a third identical word in a row is a 2-3 % event per lane, i.e. an 85 % event per wave --
                // and a hop cost ~1 500 cycles; sixteen settle all but ~1 in 2 000 visits)
                const uint32_t own2_lo = load_u32(sdata, p + deep), own2_hi = load_u32(sdata, p + deep + 4u);
                const uint32_t own3_lo = load_u32(sdata, p + deep + 8u), own3_hi = load_u32(sdata, p + deep + 12u);
                uint32_t best = 0, best_pos = 0, snap = 0xffffffffu;
                uint32_t j = on ? i - k1 : kBand;
                uint32_t w = S[j];
                bool walking = on, slow = false;
                while (__builtin_amdgcn_ballot_w64(walking) != 0) {
                    if (walking && j < kBand && !halo_links) { slow = true; walking = false; }   // into a halo of another format: the plain walk, from the start
                    const uint32_t q = band_pos(w), link = w >> 16, k = i - j;
                    const bool more = link != kBandNoLink && i - link <= cnt;
                    const uint32_t wn = S[more ? link : kBand];
                    const uint32_t x0 = load_u32(sdata, q + deep) ^ own2_lo, x1 = load_u32(sdata, q + deep + 4u) ^ own2_hi;
                    const uint32_t x2 = load_u32(sdata, q + deep + 8u) ^ own3_lo, x3 = load_u32(sdata, q + deep + 12u) ^ own3_hi;
                    if (walking) {
                        if (k > kShortChain && snap == 0xffffffffu) snap = best ? entry_pack(best, p - best_pos) : 0u;
                        uint32_t len = (x0 | x1) ? deep + (band_ctz64(x0, x1) >> 3) : (x2 | x3) ? deep + 8u + (band_ctz64(x2, x3) >> 3) : match_len_from(sdata, q, p, deep + 16u, max_len);
                        len = len < max_len ? len : max_len;
                        if (len > best) { best = len; best_pos = q; }
                        walking = more && best < nice;
                        j = link; w = wn;
                    }
                }
                uint32_t e128 = entry_pack(best, p - best_pos), e32 = snap != 0xffffffffu ? snap : e128;
                if (slow) {
                    const uint64_t own = (uint64_t)load_u32(sdata, p + off) | (uint64_t)load_u32(sdata, p + off + 4u) << 32;
                    band_deep(sdata, Sf, [&](uint32_t jj) { return S[jj] >> 16; },
                              [&](uint32_t jj) { const uint2 e = E[jj]; return (uint64_t)e.x | (uint64_t)e.y << 32; },   // (halo entries only)
                              kBand, i, cnt, k1, deep, L, own, e128, e32);
                }
                if (on) {
                    if (ZWZ_BAND_EXP & 4) asm volatile("" :: "v"(e128), "v"(e32), "v"(p));
                    if (!(ZWZ_BAND_EXP & 4)) {
                        reinterpret_cast<uint32_t*>(ent + p)[0] = e128;
                        if (k1 <= kShortChain) reinterpret_cast<uint32_t*>(ent + p)[1] = e32;
                    }
                }
            }
            __syncthreads();
            if (tid < kBand && b < n) {                                       // (a further tile follows: this one is full, m = kBandArr)
                const uint32_t lk = S[m - kBand + tid] >> 16;
                s_halo_link[tid] = (uint16_t)(lk != kBandNoLink && lk >= m - kBand ? lk - (m - kBand) : kBandNoLink);
            }
            __syncthreads();                                                  // the next tile overwrites S, E, ck and the counters
            ZWZ_STAMP(5);
        }
        __syncthreads();
        for (uint32_t i = tid; i < ((L + 63u) >> 6); i += kBandThreads) {
            hm[i] = (uint64_t)hasb[2 * i] | ((uint64_t)hasb[2 * i + 1] << 32);
            hasb[2 * i] = 0; hasb[2 * i + 1] = 0;
        }
        ZWZ_STAMP(6);
    }
}

// ------------------------------------------------------------------------------------------------
uint32_t exp_flags_band() { return (uint32_t)(ZWZ_BAND_EXP); }

hipError_t configure_band_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lz_place_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPlaceLdsBytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(lz_match_band_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBandLdsBytes);
}

// which: 0 = by a sample of each chunk (production), 2 = every chunk is chain-heavy (tests).
hipError_t launch_dense_list(const DeflateArgs& a, hipStream_t s, uint32_t which) {
    const uint32_t cus = a.cu_count ? a.cu_count : 256u, want = (a.n + kDenseThreads / 64u - 1u) / (kDenseThreads / 64u);
    hipLaunchKernelGGL(lz_dense_list_kernel, dim3(want < 8u * cus ? want : 8u * cus), dim3(kDenseThreads), 0, s, a.in, a.in_off, a.in_len, a.n, a.link_stat, which);
    hipLaunchKernelGGL(lz_lists_kernel, dim3(1), dim3(1024), 0, s, a.in_len, a.link_stat, a.n, a.dense_list, a.sparse_list, a.tickets);
    return hipGetLastError();
}

hipError_t launch_sort(const DeflateArgs& a, hipStream_t s) {
    const uint32_t cus = a.cu_count ? a.cu_count : 256u;
    const uint32_t G = a.n < 2u * cus ? a.n : 2u * cus;
    hipLaunchKernelGGL(lz_sort_kernel, dim3(G), dim3(kSortThreads), 0, s, a.in, a.in_off, a.in_len, a.dense_list, a.tickets, a.sorted, a.links,
                       reinterpret_cast<uint32_t*>(a.entries));
    return hipGetLastError();
}

hipError_t launch_place(const DeflateArgs& a, hipStream_t s) {
    const uint32_t cus = a.cu_count ? a.cu_count : 256u;
    const uint32_t G = a.n < cus ? a.n : cus;
    hipLaunchKernelGGL(lz_place_kernel, dim3(G), dim3(kPlaceThreads), kPlaceLdsBytes, s, a.in_len, a.dense_list, a.tickets, a.sorted);
    return hipGetLastError();
}

hipError_t launch_match_band(const DeflateArgs& a, hipStream_t s) {
    const uint32_t cus = a.cu_count ? a.cu_count : 256u;
    const uint32_t G = a.n < cus ? a.n : cus;
    hipLaunchKernelGGL(lz_match_band_kernel, dim3(G), dim3(kBandThreads), kBandLdsBytes, s, a.in, a.in_off, a.in_len, a.dense_list, a.tickets, a.sorted,
                       a.entries, a.has128);
    return hipGetLastError();
}

}  // namespace zwz
