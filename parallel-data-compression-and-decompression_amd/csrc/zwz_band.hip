// zwz_band.hip -- match records of chain-heavy chunks (text: four positions in five have a chain predecessor) without a
// chain walk.  Replaces, for those chunks, what lz_match's walk did with two dependent LDS gathers per candidate
// (consumer()'s longest_match, compression.cpp:119-131); the arithmetic and its proof of equivalence are csrc/lz_band.h.
//
//   lz_dense_list  one thread / chunk   chunks whose lz_links count says "chain-heavy" -> a list
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

namespace zwz {

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lz_dense_list_kernel(const uint32_t* __restrict__ in_len, const uint32_t* __restrict__ link_stat, uint32_t n,
                                                            uint32_t* __restrict__ list, uint32_t* __restrict__ tickets, uint32_t force) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n) return;
    const uint32_t L = in_len[c];
    const bool dense = force == 2u ? L != 0u : (force == 1u ? false : (L != 0u && chunk_is_dense(link_stat[c], L)));
    const uint64_t m = __builtin_amdgcn_ballot_w64(dense);
    if (m == 0) return;
    uint32_t base = 0;
    if (lane_id() == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&tickets[kTicketDenseCount], (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
    if (dense) list[base + rank_in(m)] = c;
}

// ------------------------------------------------------------------------------------------------
// lz_sort.  The table holds two 16-bit counters a dword (bucket h in half h & 1 of dword h >> 1): 64 KiB, two workgroups a CU.
// A counter never carries into its neighbour: it ends at bucket start + bucket size <= 65 533.
__global__ __launch_bounds__(kSortThreads) void lz_sort_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                               const uint32_t* __restrict__ in_len, const uint32_t* __restrict__ list,
                                                               uint32_t* __restrict__ tickets, uint32_t* __restrict__ sorted,
                                                               uint16_t* __restrict__ hbuf /* the chunks' dead link arrays */) {
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
        uint32_t* out = sorted + (size_t)chunk * kSortedStride;
        {
            uint4* t4 = reinterpret_cast<uint4*>(tab);
            for (uint32_t i = tid; i < 4096u; i += kSortThreads) t4[i] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        // histogram: a thread takes four positions a trip (one dword and its successor); the hashes go out to hbuf for the ranking
        for (uint32_t i = tid; 4u * i < n; i += kSortThreads) {
            const uint32_t w0 = d32[i], w1 = i + 1u < nd ? d32[i + 1u] : 0u;
            uint32_t h[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t x = j ? __builtin_amdgcn_alignbyte(w1, w0, j) : w0;
                h[j] = hash3(x & 0xffu, (x >> 8) & 0xffu, (x >> 16) & 0xffu);
                if (4u * i + j < n) atomicAdd(&tab[h[j] >> 1], (h[j] & 1u) ? 0x10000u : 1u);
            }
            *reinterpret_cast<uint2*>(hb + 4u * i) = make_uint2(h[0] | h[1] << 16, h[2] | h[3] << 16);
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
        // (= position order), steps follow one another in program order.  Eight steps' hashes are asked for ahead.
        if (wave == 0) {
            const uint32_t steps = (n + 63u) >> 6;
            for (uint32_t s0 = 0; s0 < steps; s0 += 8u) {
                uint32_t hv[8];
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) { const uint32_t p = (s0 + j) * 64u + lane; hv[j] = p < n ? (uint32_t)hb[p] : 0xffffffffu; }
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) {
                    const uint32_t p = (s0 + j) * 64u + lane, h = hv[j];
                    if (h != 0xffffffffu) {
                        const uint32_t old = atomicAdd(&tab[h >> 1], (h & 1u) ? 0x10000u : 1u);
                        const uint32_t dest = (h & 1u) ? old >> 16 : old & 0xffffu;
                        out[dest] = band_word(h, p);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// lz_match_band.
constexpr uint32_t kBandArr = kBandTile + kBand;                    // a tile's arrays: 128 halo entries, then the tile's own
constexpr uint32_t kBandDataBytes = 65536 + 64;                     // the chunk + slack for comparisons that run past its end
constexpr uint32_t kBandOffS = kBandDataBytes, kBandOffE = kBandOffS + kBandArr * 4, kBandOffCk = kBandOffE + kBandArr * 8,
                   kBandOffHas = kBandOffCk + kBandTile * 2;
static_assert(kBandOffHas + 8192 == kBandLdsBytes, "lz_match_band LDS layout");
static_assert(kBandTile % 64 == 0 && kBandOffE % 8 == 0 && kBandOffS % 16 == 0, "lz_match_band LDS alignment");

__global__ __launch_bounds__(kBandThreads) void lz_match_band_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                                    const uint32_t* __restrict__ in_len, const uint32_t* __restrict__ list,
                                                                    uint32_t* __restrict__ tickets, const uint32_t* __restrict__ sorted,
                                                                    uint2* __restrict__ entries, uint64_t* __restrict__ has128) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* sdata = smem;
    uint32_t* S = reinterpret_cast<uint32_t*>(smem + kBandOffS);
    uint2* E = reinterpret_cast<uint2*>(smem + kBandOffE);
    uint16_t* ck = reinterpret_cast<uint16_t*>(smem + kBandOffCk);       // per own entry: candidates | k of the nearest sharer << 8
    uint32_t* hasb = reinterpret_cast<uint32_t*>(smem + kBandOffHas);    // has128 bits of the whole chunk
    __shared__ uint32_t s_chunk, s_grp[2];
    const uint32_t tid = threadIdx.x, lane = lane_id();
    const uint32_t n_list = tickets[kTicketDenseCount];
    for (uint32_t i = tid; i < 2048u; i += kBandThreads) hasb[i] = 0;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_chunk = atomicAdd(&tickets[kTicketBandNext], 1u);
        __syncthreads();
        const uint32_t tk = s_chunk;
        if (tk >= n_list) break;
        const uint32_t chunk = list[tk];
        const uint32_t L = in_len[chunk];
        const uint32_t n = L >= kMinMatch ? L - (kMinMatch - 1u) : 0u;
        const uint32_t* srt = sorted + (size_t)chunk * kSortedStride;
        uint2* ent = entries + (size_t)chunk * kEntryStride;
        uint64_t* hm = has128 + (size_t)chunk * kMaskWords;
        copy_vec16(reinterpret_cast<uint4*>(sdata), reinterpret_cast<const uint4*>(in + in_off[chunk]), (L + 15u) >> 4);
        for (uint32_t a = 0; a < n; a += kBandTile) {
            const uint32_t b = min(a + kBandTile, n), m = b - a + kBand;      // array index i <-> sorted index a - 128 + i
            for (uint32_t i = tid; i < m; i += kBandThreads) S[i] = a + i >= kBand ? srt[a + i - kBand] : kBandHaloWord;
            if (tid == 0) { s_grp[0] = 0; s_grp[1] = 0; }
            __syncthreads();                                                  // (also: the chunk's bytes are in place)
            // one trigram per bucket within the tile's reach?  (each entry against the one in front of it)
            bool mixed = false;
            for (uint32_t i = tid; i < m; i += kBandThreads) {
                const uint32_t w = S[i], wp = i ? S[i - 1u] : kBandHaloWord;
                if (w != kBandHaloWord && wp != kBandHaloWord && band_hash(w) == band_hash(wp))
                    mixed |= ((load_u32(sdata, band_pos(w)) ^ load_u32(sdata, band_pos(wp))) & 0xffffffu) != 0u;
            }
            const bool pure = !__syncthreads_or((int)mixed);
            const uint32_t off = pure ? 3u : 0u, deep = pure ? 11u : 8u;
            for (uint32_t i = tid; i < m; i += kBandThreads) {
                const uint32_t w = S[i], q = band_pos(w) + off;
                E[i] = make_uint2(load_u32(sdata, q), load_u32(sdata, q + 4u));   // (a halo word reads position 0's bytes: never compared)
            }
            auto Sf = [&](uint32_t i) { return S[i]; };
            for (uint32_t i = kBand + tid; i < m; i += kBandThreads) ck[i - kBand] = (uint16_t)band_count(Sf, i);
            __syncthreads();
            const uint32_t n_grp = (b - a + 63u) >> 6;
            const uint32_t none = pure ? kBandKeyNonePure : kBandKeyNoneImpure;
            // ---- first pass
            for (;;) {
                uint32_t g = 0;
                if (lane == 0) g = atomicAdd(&s_grp[0], 1u);
                g = __builtin_amdgcn_readfirstlane(g);
                if (g >= n_grp) break;
                const uint32_t i = kBand + 64u * g + lane;
                const bool active = i < m;
                const uint32_t w = S[active ? i : kBand], p = band_pos(w);
                const uint32_t cnt = active ? (uint32_t)ck[i - kBand] : 0u;
                const bool tail = L - p < kBandTailLook;
                const uint32_t cntb = tail ? 0u : cnt;                         // candidates the banded loop looks at
                const uint2 own = E[active ? i : kBand];
                const uint32_t kmax = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_max_incl(cntb), 63);
                uint32_t best = none, snap = none;
                const uint2* Ei = E + i;                                        // (i - k >= 0 for every lane: k <= 128 <= i)
                for (uint32_t k0 = 0; k0 < kmax; k0 += 8u) {
                    uint2 c[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) c[j] = Ei[-(int)(k0 + j + 1u)];
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) {
                        const uint32_t k = k0 + j + 1u;
                        const uint32_t key = band_key(own.x, own.y, c[j].x, c[j].y, k);
                        best = max(best, k <= cntb ? key : 0u);
                    }
                    if (k0 + 8u == kShortChain) snap = best;
                }
                const uint32_t key32 = cntb > kShortChain ? snap : best;
                uint32_t e128 = 0, e32 = 0, k1 = 0;
                if (tail) {
                    if (active) band_generic(sdata, Sf, i, cnt, L, e128, e32);
                } else {
                    if (best != none) {
                        if (band_key_len(best) == 15u) k1 = band_key_k(best);
                        else e128 = band_record(best, pure, p, band_pos(S[i - band_key_k(best)]));
                    }
                    if (key32 != none && band_key_len(key32) != 15u) e32 = band_record(key32, pure, p, band_pos(S[i - band_key_k(key32)]));
                }
                if (active) {
                    ck[i - kBand] = (uint16_t)(cnt | k1 << 8);
                    reinterpret_cast<uint16_t*>(S)[2u * i + 1u] = (uint16_t)(k1 ? i - k1 : kBandNoLink);   // the bucket field has done its work: now the link
                    if (k1) {
                        if (k1 > kShortChain) reinterpret_cast<uint32_t*>(ent + p)[1] = e32;   // final; the second pass writes the rest
                        atomicOr(&hasb[p >> 5], 1u << (p & 31u));
                    } else if (e128) {
                        ent[p] = make_uint2(e128, e32);
                        atomicOr(&hasb[p >> 5], 1u << (p & 31u));
                    }
                }
            }
            __syncthreads();
            // ---- second pass: positions whose nearest sharer agrees on all eight bytes
            for (;;) {
                uint32_t g = 0;
                if (lane == 0) g = atomicAdd(&s_grp[1], 1u);
                g = __builtin_amdgcn_readfirstlane(g);
                if (g >= n_grp) break;
                const uint32_t i = kBand + 64u * g + lane;
                const uint32_t c = i < m ? (uint32_t)ck[i - kBand] : 0u;
                const uint32_t k1 = c >> 8;
                if (k1) {
                    const uint32_t p = band_pos(S[i]);
                    uint32_t e128 = 0, e32 = 0;
                    band_deep(sdata, Sf, [&](uint32_t j) { return S[j] >> 16; },
                              [&](uint32_t j) { const uint2 e = E[j]; return (uint64_t)e.x | (uint64_t)e.y << 32; },
                              kBand, i, c & 0xffu, k1, deep, L, e128, e32);
                    reinterpret_cast<uint32_t*>(ent + p)[0] = e128;
                    if (k1 <= kShortChain) reinterpret_cast<uint32_t*>(ent + p)[1] = e32;
                }
            }
            __syncthreads();                                                  // the next tile overwrites S, E, ck and the counters
        }
        for (uint32_t i = tid; i < ((L + 63u) >> 6); i += kBandThreads) {
            hm[i] = (uint64_t)hasb[2 * i] | ((uint64_t)hasb[2 * i + 1] << 32);
            hasb[2 * i] = 0; hasb[2 * i + 1] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
hipError_t configure_band_kernels() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(lz_match_band_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBandLdsBytes);
}

// which: 0 = by lz_links' count (production), 1 = no chunk, 2 = every chunk (tests).
hipError_t launch_dense_list(const DeflateArgs& a, hipStream_t s, uint32_t which) {
    hipLaunchKernelGGL(lz_dense_list_kernel, dim3((a.n + 255u) / 256u), dim3(256), 0, s, a.in_len, a.link_stat, a.n, a.dense_list, a.tickets, which);
    return hipGetLastError();
}

hipError_t launch_sort(const DeflateArgs& a, hipStream_t s) {
    const uint32_t cus = a.cu_count ? a.cu_count : 256u;
    const uint32_t G = a.n < 2u * cus ? a.n : 2u * cus;
    hipLaunchKernelGGL(lz_sort_kernel, dim3(G), dim3(kSortThreads), 0, s, a.in, a.in_off, a.in_len, a.dense_list, a.tickets, a.sorted, a.links);
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
