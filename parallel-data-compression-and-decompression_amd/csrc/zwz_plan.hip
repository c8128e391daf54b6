// zwz_plan.hip -- the plan stage of the chunk compressor: what zlib decides when it flushes a block (_tr_flush_block under
// consumer(), compression.cpp:131) -- the three Huffman trees, the code-length header, stored / static / dynamic.
//
//   plan_probe   wave per block    histograms sorted, exact static cost               } the stored-block shortcut: incompressible
//   plan_cost    lane per block    optimal-Huffman-cost lower bound -> stored / static  } and tiny blocks never get trees; the
//                                  settled, or the block goes onto the open list       } rest go onto a list
//   plan_heap    LANE per tree     zlib's heap (heapify + merge loop, its tie-breaks are the heap's mechanics, so it is run as
//                                  it is): sixteen open blocks a wave, nothing but the heaps in LDS; the merges go to HBM
//   plan         wave per block    everything else of build_tree / scan_tree / send_tree in wave-parallel form: depths by
//                                  pointer jumping over the merges, lengths, length counts, codes (ranks by ordered LDS adds),
//                                  code-length runs a lane each, header bits by prefix sum; the 19-symbol tree on one lane
// (Until round 3 `plan` ran huff_core.h's plan_block on lane 0 of a wave per block: ~9 300 dependent LDS round trips a block,
// 4.0 ms per 40 000 text blocks and 89 ms per 370 000 image-like ones.  ZWZ_PLAN=serial still runs that kernel: same bytes.)
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "huff_core.h"
#include "zwz_kernels.h"
#include "zwz_device.h"

namespace zwz {

// a chunk's dead link space: its kMaxBlocks BlockProbes first, its chosen records from kChosenOffset on, the merge lists last
static __device__ __forceinline__ BlockProbe* probe_of(BlockProbe* probes_base, uint32_t g) {
    uint8_t* region = reinterpret_cast<uint8_t*>(probes_base) + (size_t)(g / kMaxBlocks) * (kLinkStride * sizeof(uint16_t));
    return reinterpret_cast<BlockProbe*>(region) + g % kMaxBlocks;
}
static __device__ __forceinline__ const BlockProbe* probe_of(const BlockProbe* probes_base, uint32_t g) {
    return probe_of(const_cast<BlockProbe*>(probes_base), g);
}
static __device__ __forceinline__ uint32_t* pairs_of(const BlockProbe* probes_base, uint32_t g) {
    uint8_t* region = reinterpret_cast<uint8_t*>(const_cast<BlockProbe*>(probes_base)) + (size_t)(g / kMaxBlocks) * (kLinkStride * sizeof(uint16_t)) + kPairsOffset;
    return reinterpret_cast<uint32_t*>(region) + (g % kMaxBlocks) * kPairWords;
}

// ------------------------------------------------------------------------------------------------
// plan, in three launches.
//   plan_probe (one wave per block): the stored-block shortcut's parallel half -- exact static_len /
//       extra bits / used codes, and both histograms sorted ascending (by counting when every count is
//       small, else by rank) -> BlockProbe.
//   plan_cost  (one LANE per block): the optimal Huffman cost of the sorted counts by two-queue
//       merge, 64 blocks per wave, then huff_core.h's shortcut_type.  The merge is a chain of
//       ~m dependent steps; a wave per block spends them on one lane (and on the CU's one scalar
//       issue slot per cycle when written with readlanes -- measured 560 cycles per step with 20
//       such waves on a CU), a lane per block runs 64 chains in each instruction.
//   plan       (one wave per block): blocks the shortcut did not settle get zlib's exact tree
//       construction on lane 0 with its scratch in LDS.
// Sorting: composite keys (count << 9 | symbol) make the order total, so a lane's rank is a plain
// count of smaller keys; the keys are read back four per LDS access and compared against all of the
// lane's (up to five) own keys at once.
constexpr uint32_t kSortKeys = 288;   // kLCodes rounded up to a multiple of 4
static __device__ __forceinline__ uint32_t wave_rank_sort(const uint16_t* freq, uint32_t n, uint32_t* keys, uint16_t* sorted) {
    // ascending order of the non-zero counts (ties by symbol; any order gives the same cost) -> sorted[0..m), returns m
    const uint32_t lane = lane_id();
    const uint32_t n4 = (n + 3u) & ~3u;
    uint32_t k[5], rank[5] = {0, 0, 0, 0, 0}, zeros = 0;
#pragma unroll
    for (uint32_t r = 0; r < 5; r++) {
        const uint32_t i = lane + 64u * r;
        k[r] = i < n ? ((uint32_t)freq[i] << 9) | i : 0xffffffffu;
        if (i < n4) keys[i] = k[r];
        zeros += (uint32_t)__popcll(__ballot(i < n && (k[r] >> 9) == 0));
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const uint4* k4 = reinterpret_cast<const uint4*>(keys);
    const uint32_t rounds = (n + 63u) / 64u;            // wave-uniform: own keys in use
#pragma unroll 2
    for (uint32_t j = 0; j < n4 / 4u; j++) {
        const uint4 q = k4[j];
#pragma unroll
        for (uint32_t r = 0; r < 5; r++)
            if (r < rounds) rank[r] += (uint32_t)(q.x < k[r]) + (uint32_t)(q.y < k[r]) + (uint32_t)(q.z < k[r]) + (uint32_t)(q.w < k[r]);
    }
#pragma unroll
    for (uint32_t r = 0; r < 5; r++) {
        const uint32_t i = lane + 64u * r;
        if (i < n && (k[r] >> 9) != 0) sorted[rank[r] - zeros] = (uint16_t)(k[r] >> 9);
    }
    return n - zeros;
}

// The same order by counting when every count is small (a 16 383-symbol block of incompressible bytes: all of them
// between ~30 and ~100): a 128-bin histogram of the counts by LDS atomics, a wave scan, and every lane writes out its
// two bins -- ~100 instructions against the rank sort's ~3 000.  Returns m, or 0xffffffff if some count is >= 128.
static __device__ __forceinline__ uint32_t wave_count_sort(const uint16_t* freq, uint32_t n, uint32_t* bins /* 128 */, uint16_t* sorted) {
    const uint32_t lane = lane_id();
    uint32_t f[5]; bool big = false;
#pragma unroll
    for (uint32_t r = 0; r < 5; r++) { const uint32_t i = lane + 64u * r; f[r] = i < n ? (uint32_t)freq[i] : 0u; big = big || f[r] >= 128u; }
    if (__ballot(big)) return 0xffffffffu;
    bins[lane] = 0; bins[lane + 64u] = 0;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (uint32_t r = 0; r < 5; r++) if (f[r]) atomicAdd(&bins[f[r]], 1u);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const uint32_t c0 = lane ? bins[2u * lane] : 0u, c1 = bins[2u * lane + 1u];     // bin 0 = unused symbols: not sorted
    uint32_t incl = c0 + c1;
    for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    uint32_t at = incl - c0 - c1;
    for (uint32_t k = 0; k < c0; k++) sorted[at++] = (uint16_t)(2u * lane);
    for (uint32_t k = 0; k < c1; k++) sorted[at++] = (uint16_t)(2u * lane + 1u);
    return __shfl(incl, 63);
}

// (Round 5, measured and dropped: one workgroup a CHUNK with its blocks one after the other, here and in plan_kernel -- a workgroup per (chunk, block
// slot) is five dispatches a chunk of which four usually find nothing to do, 264 000 a launch for 52 900 small files: plan stage 16.6 -> 19.1 ms on the
// small files, 1.05 -> 1.34 ms on text.  The empty workgroups cost less than the loop's barriers and the lost overlap between a chunk's blocks.)
__global__ __launch_bounds__(64) void plan_probe_kernel(const ChunkInfo* __restrict__ info, const BlockInfo* __restrict__ blocks,
                                                        BlockProbe* __restrict__ probes) {
    __shared__ uint16_t lf[kLCodes + 2], df[kDCodes + 2];
    __shared__ __attribute__((aligned(16))) uint32_t keys[kSortKeys];
    const uint32_t chunk = blockIdx.x / kMaxBlocks, b = blockIdx.x % kMaxBlocks;
    BlockProbe* pb = probe_of(probes, blockIdx.x);
    const BlockInfo* bi = blocks + blockIdx.x;
    if (b >= info[chunk].n_blocks) { if (threadIdx.x == 0) pb->state = kProbeNone; return; }
    const uint32_t stored_len = bi->end - bi->start;
    const bool stored_ok = !(bi->flush_pos >= kSlidePos && bi->start < kWSize);
    for (uint32_t i = threadIdx.x; i < kLCodes; i += 64) lf[i] = bi->lfreq[i];
    if (threadIdx.x < kDCodes) df[threadIdx.x] = bi->dfreq[threadIdx.x];
    __syncthreads();
    // exact static_len / extra bits / used codes: five symbols per lane, wave reduction
    StoredProbe pr{0, 0, 0};
    for (uint32_t n = threadIdx.x; n < kLCodes + kDCodes; n += 64) {
        const bool lit = n < kLCodes;
        const uint32_t f = lit ? lf[n] : df[n - kLCodes];
        if (!f) continue;
        const uint32_t x = lit ? (n >= 257u ? length_extra_bits(n - 257u) : 0u) : dist_extra_bits(n - kLCodes);
        pr.static_len += f * ((lit ? static_lit_len(n) : 5u) + x); pr.extra_bits += f * x; pr.used++;
    }
    for (uint32_t d = 32; d >= 1; d >>= 1) {
        pr.static_len += __shfl_xor(pr.static_len, d); pr.extra_bits += __shfl_xor(pr.extra_bits, d); pr.used += __shfl_xor(pr.used, d);
    }
    uint32_t m_l = wave_count_sort(lf, kLCodes, keys, pb->lit);
    if (m_l == 0xffffffffu) m_l = wave_rank_sort(lf, kLCodes, keys, pb->lit);
    __syncthreads();
    uint32_t m_d = wave_count_sort(df, kDCodes, keys, pb->dist);
    if (m_d == 0xffffffffu) m_d = wave_rank_sort(df, kDCodes, keys, pb->dist);
    if (threadIdx.x == 0) {
        pb->static_len = pr.static_len; pb->extra_bits = pr.extra_bits; pb->used = pr.used;
        pb->m_l = m_l; pb->m_d = m_d; pb->stored_len = stored_len; pb->stored_ok = stored_ok; pb->state = kProbeOpen;
    }
}

// Two-queue merge on one lane, in place: a[0..m) ascending leaves.  Internal node k is written to
// a[k]; that slot is always a leaf already consumed (after k merges 2k items are gone, of which at
// most k were internal nodes, so at least k leaves -- and a step reads its inputs before it writes).
// All lanes run the loop to `steps` = the wave's largest m; a lane is live while step < its own m.
static __device__ __forceinline__ uint32_t lane_huffman_cost(uint16_t* a, uint32_t m, uint32_t steps) {
    constexpr uint32_t kInf = 0x7fffffffu;
    uint32_t li = 0, qi = 0, qn = 0, cost = 0;
    for (uint32_t step = 1; step < steps; step++) {
        if (step < m) {
            const uint32_t l0 = li < m ? (uint32_t)a[li] : kInf, l1 = li + 1 < m ? (uint32_t)a[li + 1] : kInf;
            const uint32_t q0 = qi < qn ? (uint32_t)a[qi] : kInf, q1 = qi + 1 < qn ? (uint32_t)a[qi + 1] : kInf;
            // first = smaller head (leaf on ties), second = smaller of the heads left
            const uint32_t tl = l0 <= q0 ? 1u : 0u;
            const uint32_t first = tl ? l0 : q0, nl = tl ? l1 : l0, nq = tl ? q0 : q1;
            const uint32_t t2 = nl <= nq ? 1u : 0u;
            const uint32_t sum = first + (t2 ? nl : nq);
            li += tl + t2; qi += 2u - tl - t2;
            cost += sum;
            a[qn++] = (uint16_t)sum;                    // <= 16384: the block's symbol count
        }
    }
    return cost;
}

constexpr uint32_t kCostLaneWords = 161;   // 160 words of sorted counts per lane, odd stride against bank conflicts
constexpr uint32_t kCostLanes = 16;        // blocks per wave: a lane's row is 644 bytes and the merge is a chain of LDS round trips per lane -- with 64
                                           // rows a wave (41 KB) three waves a CU had nothing to hide them behind: 0.51 ms per 250 000 blocks,
                                           // 32 rows 0.33, 16 rows (fifteen waves a CU) 0.30
__global__ __launch_bounds__(64) void plan_cost_kernel(BlockProbe* __restrict__ probes, uint32_t n_blocks_total, uint32_t* __restrict__ open_list,
                                                       uint32_t* __restrict__ tickets) {
    __shared__ uint32_t arr[kCostLanes * kCostLaneWords];
    const uint32_t lane = threadIdx.x, g0 = blockIdx.x * kCostLanes;
    const uint32_t g = g0 + lane;
    const bool mine_ok = lane < kCostLanes && g < n_blocks_total;
    BlockProbe* mine = probe_of(probes, mine_ok ? g : 0u);
    const bool live = mine_ok && mine->state == kProbeOpen;
    uint32_t m_l = 0, m_d = 0;
    if (live) { m_l = mine->m_l; m_d = mine->m_d; }
    // stage the sorted counts of every probed block of this wave: lane-private rows, coalesced copy
    for (uint32_t j = 0; j < kCostLanes; j++) {
        const uint32_t ml = __shfl(m_l, j), md = __shfl(m_d, j);
        if (ml == 0) continue;                           // not a block (a block counts at least its end-of-block symbol)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(probe_of(probes, g0 + j)->lit);
        uint32_t* dst = arr + j * kCostLaneWords;
        for (uint32_t w = lane; w < (ml + 1u) / 2u; w += 64) dst[w] = src[w];
        if (lane < (md + 1u) / 2u) dst[144 + lane] = src[144 + lane];
    }
    __syncthreads();
    uint32_t max_l = m_l, max_d = m_d;
    for (uint32_t d = 32; d >= 1; d >>= 1) { max_l = max(max_l, (uint32_t)__shfl_xor(max_l, d)); max_d = max(max_d, (uint32_t)__shfl_xor(max_d, d)); }
    uint16_t* row = reinterpret_cast<uint16_t*>(arr + (lane < kCostLanes ? lane : 0u) * kCostLaneWords);   // (lanes without a row have m = 0: they never touch it)
    const uint32_t hl = lane_huffman_cost(row, m_l, max_l);
    const uint32_t hd = lane_huffman_cost(row + 288, m_d, max_d);
    bool open = false;
    if (live) {
        const StoredProbe pr{mine->static_len, mine->extra_bits, mine->used};
        const uint32_t t = shortcut_type(pr, hl, hd, mine->stored_len, mine->stored_ok != 0);
        if (t != kShortNone) mine->state = t == kShortStored ? kProbeStored : kProbeStatic;
        else open = true;
    }
    // the blocks that need zlib's trees, as a list: plan_heap gives every lane one of them
    const uint64_t om = __ballot(open);
    if (om) {
        const uint32_t first = (uint32_t)__builtin_ctzll(om);
        uint32_t base = 0;
        if (lane == first) base = atomicAdd(&tickets[kTicketOpenCount], (uint32_t)__popcll(om));
        base = __shfl(base, first);
        if (open) open_list[base + rank_in(om)] = g;
    }
}

__global__ __launch_bounds__(64) void plan_serial_kernel(const ChunkInfo* __restrict__ info, const BlockInfo* __restrict__ blocks,
                                                  const BlockProbe* __restrict__ probes, BlockOut* __restrict__ plans) {
    __shared__ TreeScratch ts;
    __shared__ BlockCodes bc;
    __shared__ uint32_t hdr[kHdrWords];
    __shared__ uint16_t lf[kLCodes], df[kDCodes];
    const uint32_t chunk = blockIdx.x / kMaxBlocks, b = blockIdx.x % kMaxBlocks;
    if (b >= info[chunk].n_blocks) return;
    const BlockInfo* bi = blocks + (size_t)chunk * kMaxBlocks + b;
    BlockOut* bo = plans + (size_t)chunk * kMaxBlocks + b;
    const uint32_t last = b + 1 == info[chunk].n_blocks;
    const BlockProbe* mine = probe_of(probes, blockIdx.x);
    const uint32_t settled = mine->state;
    if (settled == kProbeStored) {                      // codes are never read for stored blocks
        if (threadIdx.x == 0) { bo->type = kStored; bo->hdr_bits = 3; bo->body_bits = 0; bo->eob_len = 0; bo->eob_code = 0; bo->hdr[0] = last; }
        return;
    }
    if (settled == kProbeStatic) {                      // the static codes, written out so the encoder needs no special case
        if (threadIdx.x == 0) {
            bo->type = kStatic; bo->hdr_bits = 3; bo->body_bits = mine->static_len; bo->hdr[0] = (1u << 1) + last;
            bo->eob_len = static_lit_len(256); bo->eob_code = static_lit_code(256);
        }
        for (uint32_t i = threadIdx.x; i < kLCodes; i += 64) { bo->llen[i] = (uint8_t)static_lit_len(i); bo->lcode[i] = (uint16_t)static_lit_code(i); }
        if (threadIdx.x < kDCodes) { bo->dlen[threadIdx.x] = 5; bo->dcode[threadIdx.x] = (uint16_t)bit_reverse(threadIdx.x, 5); }
        return;
    }
    for (uint32_t i = threadIdx.x; i < kLCodes; i += 64) lf[i] = bi->lfreq[i];
    if (threadIdx.x < kDCodes) df[threadIdx.x] = bi->dfreq[threadIdx.x];
    for (uint32_t i = threadIdx.x; i < kHdrWords; i += 64) hdr[i] = 0;
    __syncthreads();
    const uint32_t stored_len = bi->end - bi->start;
    const bool stored_ok = !(bi->flush_pos >= kSlidePos && bi->start < kWSize);

    if (threadIdx.x == 0) {
        BlockPlan bp = plan_block(ts, lf, df, stored_len, stored_ok, last, bc, hdr);
        bo->type = bp.type; bo->hdr_bits = bp.hdr_bits; bo->body_bits = bp.body_bits;
        bo->eob_len = bc.llen[256]; bo->eob_code = bc.lcode[256];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kLCodes; i += 64) { bo->llen[i] = bc.llen[i]; bo->lcode[i] = bc.lcode[i]; }
    if (threadIdx.x < kDCodes) { bo->dlen[threadIdx.x] = bc.dlen[threadIdx.x]; bo->dcode[threadIdx.x] = bc.dcode[threadIdx.x]; }
    for (uint32_t i = threadIdx.x; i < kHdrWords; i += 64) bo->hdr[i] = hdr[i];
}


// ------------------------------------------------------------------------------------------------
// plan_heap: a lane per tree.  A one-wave workgroup takes 16 open blocks: it lays the blocks' heaps out in LDS (build_tree's
// first loop, wave-cooperative: used symbols in symbol order by ballot ranks; forced symbols behind), runs the 16 heaps, a
// lane each (huff_core.h: heap_merge_all), and copies the merges out.  Literal/length trees first, then the distance trees in
// the same space.  Rows are an odd number of words apart.
// What a launch costs is one heap's own chain -- ~3 800 sift levels for 256 used symbols, ~30 instructions and one LDS round trip
// each, at a lone wave's issue rate: 0.63 ms -- times the rounds it takes to seat every workgroup; the heaps a CU holds at once
// are bounded by the LDS either way (128), so rows per workgroup do not matter to the time (64 rows / 4 waves: 0.64 ms per 30 000
// image-like blocks; 16 rows / 1 wave: 0.63) -- sixteen keeps a wave's lanes closer in trip count.
constexpr uint32_t kHeapLanes = 16, kHeapStride = 289, kHeapThreads = 64;

template <uint32_t E>
static __device__ __forceinline__ uint32_t wave_heap_init(const uint32_t (&f)[(E + 63u) / 64u], uint32_t* h) {
    const uint32_t lane = lane_id();
    uint32_t m = 0; int maxc = -1;
#pragma unroll
    for (uint32_t r = 0; r < (E + 63u) / 64u; r++) {
        const uint64_t mask = __ballot(f[r] != 0u);
        if (f[r]) h[1u + m + rank_in(mask)] = f[r] << 16 | (lane + 64u * r);
        m += (uint32_t)__popcll(mask);
        if (mask) maxc = (int)(64u * r + 63u - (uint32_t)__builtin_clzll(mask));
    }
    uint32_t node[2] = {0, 0};
    const uint32_t m0 = m, nf = tree_forced_nodes(m, maxc, node);
    if (lane < nf) h[1u + m0 + lane] = 1u << 16 | (lane ? node[1] : node[0]);
    return m;
}
template <uint32_t E>
static __device__ __forceinline__ void load_freq(const uint16_t* __restrict__ g, uint32_t (&f)[(E + 63u) / 64u]) {
#pragma unroll
    for (uint32_t r = 0; r < (E + 63u) / 64u; r++) { const uint32_t i = lane_id() + 64u * r; f[r] = i < E ? (uint32_t)g[i] : 0u; }
}

template <uint32_t E, bool kLit>
static __device__ __forceinline__ void heap_pass(uint32_t* reg, const BlockInfo* __restrict__ blocks, const BlockProbe* __restrict__ probes,
                                                 const uint32_t* s_g, uint32_t* s_m, uint32_t cnt) {
    constexpr uint32_t T = (E + 63u) / 64u;
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto freq_of = [&](uint32_t j) { const BlockInfo* bi = blocks + s_g[j]; return kLit ? bi->lfreq : bi->dfreq; };
    {   // the heaps, a wave every fourth block; the next block's counts are asked for before this one's are used
        uint32_t cur[T], nxt[T];
        uint32_t j = wave;
        if (j < cnt) load_freq<E>(freq_of(j), cur);
        for (; j < cnt; j += kHeapThreads / 64u) {
            const bool more = j + kHeapThreads / 64u < cnt;
            if (more) load_freq<E>(freq_of(j + kHeapThreads / 64u), nxt);
            const uint32_t m = wave_heap_init<E>(cur, reg + j * kHeapStride);
            if (lane == 0) s_m[j] = m;
            if (more) {
#pragma unroll
                for (uint32_t r = 0; r < T; r++) cur[r] = nxt[r];
            }
        }
    }
    __syncthreads();
    if (tid < cnt) heap_merge_all(reg + tid * kHeapStride, s_m[tid], E);
    __syncthreads();
    for (uint32_t j = wave; j < cnt; j += kHeapThreads / 64u) {      // merge s of a heap of m: row[m - s]
        const uint32_t m = s_m[j];
        uint32_t* out = pairs_of(probes, s_g[j]) + (kLit ? 0u : kPairLitWords);
        const uint32_t* row = reg + j * kHeapStride;
        for (uint32_t s = lane; s + 1u < m; s += 64u) out[s] = row[m - s];
    }
    __syncthreads();
}

__global__ __launch_bounds__(kHeapThreads) void plan_heap_kernel(const BlockInfo* __restrict__ blocks, const BlockProbe* __restrict__ probes,
                                                                 const uint32_t* __restrict__ open_list, const uint32_t* __restrict__ tickets) {
    __shared__ uint32_t reg[kHeapLanes * kHeapStride];
    __shared__ uint32_t s_g[kHeapLanes], s_m[kHeapLanes];
    const uint32_t n_open = tickets[kTicketOpenCount], t0 = blockIdx.x * kHeapLanes;
    if (t0 >= n_open) return;
    const uint32_t cnt = min(kHeapLanes, n_open - t0);
    if (threadIdx.x < cnt) s_g[threadIdx.x] = open_list[t0 + threadIdx.x];
    __syncthreads();
    heap_pass<kLCodes, true>(reg, blocks, probes, s_g, s_m, cnt);
    heap_pass<kDCodes, false>(reg, blocks, probes, s_g, s_m, cnt);
}

// ------------------------------------------------------------------------------------------------
// plan: a wave per block.  Blocks the shortcut settled are written out as before; an open block's trees are finished from
// its merge lists.
struct PlanMem {
    uint32_t pairs[kPairLitWords];            // the merges of the tree at hand: n_s | m_s << 16, node kElems + s
    uint16_t par[kPairLitWords];              // pointer jumping: ancestor (as a merge index) ...
    uint8_t dep[kPairLitWords];               // ... and the distance to it
    uint8_t len[kLCodes + 2 + kDCodes + 2];   // code lengths: literal/length symbols, then distance symbols from kLCodes + 2 on
    uint32_t blc[16], nc[16];                 // bl_count, next_code of the tree at hand
    uint32_t blf[kBLCodes + 1];               // counts of the code-length symbols
    uint16_t blfreq[kBLCodes + 1];
    uint8_t bllen[kBLCodes + 1];
    uint16_t blcode[kBLCodes + 1];
    uint32_t hdr[kHdrWords];
    BlTreeScratch bts;
    uint32_t type, fixed_bits, opt_len;       // lane 0's decisions
};

static __device__ __forceinline__ void wave_sync() { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }   // one wave: LDS traffic settled

// gen_bitlen + gen_codes of one tree from its merges.  f[] = the lane's symbol counts (symbol lane + 64 r); lens -> LDS, lengths
// and bit-reversed codes -> the block's BlockOut.  Returns max_code.
template <uint32_t E, class ExtraFn, class StatFn>
static __device__ __forceinline__ int wave_tree_finish(PlanMem& mem, uint32_t (&f)[(E + 63u) / 64u], const uint32_t* __restrict__ gpairs, uint8_t* lens,
                                                       uint8_t* __restrict__ glen, uint16_t* __restrict__ gcode, ExtraFn extra_bits, StatFn static_len_of,
                                                       uint32_t& opt_len, uint32_t& static_len) {
    constexpr uint32_t T = (E + 63u) / 64u, kMaxLen = 15u;
    const uint32_t lane = lane_id();
    uint32_t m = 0; int maxc = -1;
#pragma unroll
    for (uint32_t r = 0; r < T; r++) {
        const uint64_t mask = __ballot(f[r] != 0u);
        m += (uint32_t)__popcll(mask);
        if (mask) maxc = (int)(64u * r + 63u - (uint32_t)__builtin_clzll(mask));
    }
    uint32_t node[2] = {0, 0};
    const uint32_t nf = tree_forced_nodes(m, maxc, node);
#pragma unroll
    for (uint32_t k = 0; k < 2; k++) {
        if (k < nf) {
            opt_len--; static_len -= static_len_of(node[k]);
#pragma unroll
            for (uint32_t r = 0; r < T; r++) if (node[k] == lane + 64u * r) f[r] = 1u;
        }
    }
    const uint32_t nm = __builtin_amdgcn_readfirstlane(m - 1u), root = nm - 1u;       // merges; the last one made the root
#pragma unroll
    for (uint32_t r = 0; r < T; r++) { const uint32_t i = lane + 64u * r; if (i < E + 2u) lens[i] = 0; }
    for (uint32_t s = lane; s < nm; s += 64u) { mem.pairs[s] = gpairs[s]; mem.par[s] = (uint16_t)s; }
    if (lane < 16u) mem.blc[lane] = 0;
    wave_sync();
    for (uint32_t s = lane; s < nm; s += 64u) {
        const uint32_t pr = mem.pairs[s], a = pr & 0xffffu, b = pr >> 16;
        if (a >= E) mem.par[a - E] = (uint16_t)s;
        if (b >= E) mem.par[b - E] = (uint16_t)s;
    }
    wave_sync();
    // depth of every internal node: pointer jumping towards the root (a tree over <= 16 384 symbols is < 64 deep: six rounds)
    uint32_t A[T], D[T];
#pragma unroll
    for (uint32_t r = 0; r < T; r++) { const uint32_t s = lane + 64u * r; A[r] = s < nm ? (uint32_t)mem.par[s] : root; D[r] = s < nm && s != root ? 1u : 0u; }
    for (uint32_t round = 0; round < 6u; round++) {
        bool far = false;
#pragma unroll
        for (uint32_t r = 0; r < T; r++) { const uint32_t s = lane + 64u * r; if (s < nm) { mem.dep[s] = (uint8_t)D[r]; mem.par[s] = (uint16_t)A[r]; far = far || A[r] != root; } }
        if (!__ballot(far)) break;
        wave_sync();
#pragma unroll
        for (uint32_t r = 0; r < T; r++) { const uint32_t s = lane + 64u * r; if (s < nm) { const uint32_t a = A[r]; D[r] += mem.dep[a]; A[r] = mem.par[a]; } }
        wave_sync();
    }
    // lengths of the leaves, capped; bl_count; overflow = nodes of either kind deeper than the cap
    uint32_t over = 0;
#pragma unroll
    for (uint32_t r = 0; r < T; r++) {
        const uint32_t s = lane + 64u * r;
        const bool valid = s < nm;
        const uint32_t pr = valid ? mem.pairs[s] : 0u, a = pr & 0xffffu, b = pr >> 16, bits = D[r] + 1u, capped = bits < kMaxLen ? bits : kMaxLen;
        if (valid && a < E) { lens[a] = (uint8_t)capped; atomicAdd(&mem.blc[capped], 1u); }
        if (valid && b < E) { lens[b] = (uint8_t)capped; atomicAdd(&mem.blc[capped], 1u); }
        over += (uint32_t)__popcll(__ballot(valid && s != root && D[r] > kMaxLen)) + (uint32_t)__popcll(__ballot(valid && a < E && bits > kMaxLen)) +
                (uint32_t)__popcll(__ballot(valid && b < E && bits > kMaxLen));
    }
    wave_sync();
    if (over) {                                          // rare (a tree deeper than 15): zlib's repair, on one lane
        if (lane == 0) tree_fix_overflow(mem.blc, (int)kMaxLen, (int)over, [&](uint32_t s) { return mem.pairs[s]; }, nm, maxc, lens);
        wave_sync();
    }
    if (lane == 0) {
        uint32_t code = 0;
        mem.nc[0] = 0;
        for (uint32_t b = 1; b <= 15u; b++) { code = (code + mem.blc[b - 1u]) << 1; mem.nc[b] = code; }
    }
    wave_sync();
    // cost under these lengths; codes: next_code[len]++ in symbol order = an LDS add per symbol (lanes of one instruction are
    // served in lane order, trips in program order -- the property zwz_ctx_create checks for lz_sort)
    uint32_t o = 0, st = 0;
#pragma unroll
    for (uint32_t r = 0; r < T; r++) {
        const uint32_t i = lane + 64u * r;
        const uint32_t l = i < E ? (uint32_t)lens[i] : 0u;
        uint32_t code = 0;
        if (l) code = atomicAdd(&mem.nc[l], 1u);
        if (i < E) { glen[i] = (uint8_t)l; gcode[i] = l ? (uint16_t)bit_reverse(code, l) : (uint16_t)0; }
        if (f[r]) { const uint32_t x = extra_bits(i); o += f[r] * (l + x); st += f[r] * (static_len_of(i) + x); }
    }
    for (uint32_t d = 32; d >= 1; d >>= 1) { o += __shfl_xor(o, d); st += __shfl_xor(st, d); }
    opt_len += o; static_len += st;
    return maxc;
}

// The maximal runs of equal lengths in lens[0..maxc]: c[r] = length of the run that starts at symbol lane + 64 r, 0 if none does.
template <uint32_t E>
static __device__ __forceinline__ void wave_runs(const uint8_t* lens, int maxc, uint32_t (&v)[(E + 63u) / 64u], uint32_t (&c)[(E + 63u) / 64u]) {
    const uint32_t lane = lane_id();
    uint32_t next_after = (uint32_t)(maxc + 1);
#pragma unroll
    for (int r = (int)((E + 63u) / 64u) - 1; r >= 0; r--) {
        const uint32_t i = lane + 64u * (uint32_t)r;
        const bool valid = (int)i <= maxc;
        v[r] = valid ? (uint32_t)lens[i] : 0xffu;
        const uint32_t pv = valid && i ? (uint32_t)lens[i - 1u] : 0xfeu;
        const bool start = valid && v[r] != pv;
        const uint64_t mask = __ballot(start), rest = lane == 63u ? 0ull : mask >> (lane + 1u);
        const uint32_t nxt = rest ? i + 1u + (uint32_t)__builtin_ctzll(rest) : next_after;
        c[r] = start ? nxt - i : 0u;
        if (mask) next_after = 64u * (uint32_t)r + (uint32_t)__builtin_ctzll(mask);
    }
}

static __device__ __forceinline__ void lds_or_bits32(uint32_t* words, uint32_t pos, uint32_t v, uint32_t n) {   // n <= 14
    if (n == 0) return;
    const uint32_t w = pos >> 5, o = pos & 31u;
    atomicOr(&words[w], v << o);
    if (o + n > 32u) atomicOr(&words[w + 1u], v >> (32u - o));
}

__global__ __launch_bounds__(64) void plan_kernel(const ChunkInfo* __restrict__ info, const BlockInfo* __restrict__ blocks,
                                                  const BlockProbe* __restrict__ probes, BlockOut* __restrict__ plans) {
    __shared__ PlanMem mem;
    const uint32_t chunk = blockIdx.x / kMaxBlocks, b = blockIdx.x % kMaxBlocks, lane = threadIdx.x;
    if (b >= info[chunk].n_blocks) return;
    const BlockInfo* bi = blocks + (size_t)chunk * kMaxBlocks + b;
    BlockOut* bo = plans + (size_t)chunk * kMaxBlocks + b;
    const uint32_t last = b + 1 == info[chunk].n_blocks;
    const BlockProbe* mine = probe_of(probes, blockIdx.x);
    const uint32_t settled = mine->state;
    if (settled == kProbeStored) {                      // codes are never read for stored blocks
        if (lane == 0) { bo->type = kStored; bo->hdr_bits = 3; bo->body_bits = 0; bo->eob_len = 0; bo->eob_code = 0; bo->hdr[0] = last; }
        return;
    }
    auto write_static_codes = [&]() {                  // the static codes, written out so the encoder needs no special case
        for (uint32_t i = lane; i < kLCodes; i += 64) { bo->llen[i] = (uint8_t)static_lit_len(i); bo->lcode[i] = (uint16_t)static_lit_code(i); }
        if (lane < kDCodes) { bo->dlen[lane] = 5; bo->dcode[lane] = (uint16_t)bit_reverse(lane, 5); }
    };
    if (settled == kProbeStatic) {
        if (lane == 0) {
            bo->type = kStatic; bo->hdr_bits = 3; bo->body_bits = mine->static_len; bo->hdr[0] = (1u << 1) + last;
            bo->eob_len = static_lit_len(256); bo->eob_code = static_lit_code(256);
        }
        write_static_codes();
        return;
    }
    const uint32_t stored_len = bi->end - bi->start;
    const bool stored_ok = !(bi->flush_pos >= kSlidePos && bi->start < kWSize);
    uint32_t lf[(kLCodes + 63u) / 64u], df[1];
    load_freq<kLCodes>(bi->lfreq, lf);
    load_freq<kDCodes>(bi->dfreq, df);
    for (uint32_t i = lane; i < kHdrWords; i += 64) mem.hdr[i] = 0;
    if (lane <= kBLCodes) mem.blf[lane] = 0;
    uint32_t opt_len = 0, static_len = 0;
    const uint32_t* gp = pairs_of(probes, blockIdx.x);
    uint8_t* llen = mem.len; uint8_t* dlen = mem.len + kLCodes + 2u;
    const int l_max = wave_tree_finish<kLCodes>(mem, lf, gp, llen, bo->llen, bo->lcode, [](uint32_t n) { return n >= 257u ? length_extra_bits(n - 257u) : 0u; },
                                                [](uint32_t n) { return static_lit_len(n); }, opt_len, static_len);
    wave_sync();
    const int d_max = wave_tree_finish<kDCodes>(mem, df, gp + kPairLitWords, dlen, bo->dlen, bo->dcode, [](uint32_t n) { return dist_extra_bits(n); },
                                                [](uint32_t) { return 5u; }, opt_len, static_len);
    wave_sync();
    // scan_tree over both length arrays: a lane per run
    uint32_t lv[(kLCodes + 63u) / 64u], lc[(kLCodes + 63u) / 64u], dv[1], dc[1];
    wave_runs<kLCodes>(llen, l_max, lv, lc);
    wave_runs<kDCodes>(dlen, d_max, dv, dc);
    auto tally = [&](uint32_t sym, uint32_t, uint32_t) { atomicAdd(&mem.blf[sym], 1u); };
#pragma unroll
    for (uint32_t r = 0; r < (kLCodes + 63u) / 64u; r++) if (lc[r]) rle_run(lv[r], lc[r], tally);
    if (dc[0]) rle_run(dv[0], dc[0], tally);
    wave_sync();
    if (lane == 0) {                                    // the code-length tree (19 symbols) and the decision, as plan_block makes it
        for (uint32_t i = 0; i < kBLCodes; i++) mem.blfreq[i] = (uint16_t)mem.blf[i];
        uint32_t dummy_static = 0;
        build_tree(mem.bts, mem.blfreq, (int)kBLCodes, 7, [](uint32_t n) { return n < 16u ? 0u : n == 16u ? 2u : n == 17u ? 3u : 7u; },
                   [](uint32_t) { return 0u; }, false, mem.bllen, mem.blcode, opt_len, dummy_static);
        int mbi;
        for (mbi = (int)kBLCodes - 1; mbi >= 3; mbi--)
            if (mem.bllen[bl_order((uint32_t)mbi)] != 0) break;
        opt_len += 3u * ((uint32_t)mbi + 1u) + 5u + 5u + 4u;
        uint32_t opt_lenb = (opt_len + 3u + 7u) >> 3;
        const uint32_t static_lenb = (static_len + 3u + 7u) >> 3;
        if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
        BitSink sink{mem.hdr, 0};
        uint32_t type;
        if (stored_len + 4u <= opt_lenb && stored_ok) { type = kStored; sink.put((0u << 1) + last, 3); }
        else if (static_lenb == opt_lenb) { type = kStatic; sink.put((1u << 1) + last, 3); }
        else {
            type = kDynamic;
            sink.put((2u << 1) + last, 3);
            sink.put((uint32_t)(l_max + 1 - 257), 5);
            sink.put((uint32_t)(d_max + 1 - 1), 5);
            sink.put((uint32_t)(mbi + 1 - 4), 4);
            for (int r = 0; r <= mbi; r++) sink.put(mem.bllen[bl_order((uint32_t)r)], 3);
        }
        mem.type = type; mem.fixed_bits = sink.nbits; mem.opt_len = opt_len;
    }
    wave_sync();
    const uint32_t type = mem.type;
    uint32_t hdr_bits = mem.fixed_bits;
    if (type == kDynamic) {                             // send_tree: every run's bits at the prefix sum of the runs before it
        auto run_bits = [&](uint32_t v, uint32_t c) { uint32_t n = 0; if (c) rle_run(v, c, [&](uint32_t sym, uint32_t, uint32_t xn) { n += (uint32_t)mem.bllen[sym] + xn; }); return n; };
        auto run_send = [&](uint32_t v, uint32_t c, uint32_t pos) {
            if (c) rle_run(v, c, [&](uint32_t sym, uint32_t xv, uint32_t xn) {
                const uint32_t bl = mem.bllen[sym];
                lds_or_bits32(mem.hdr, pos, (uint32_t)mem.blcode[sym] | xv << bl, bl + xn);
                pos += bl + xn;
            });
        };
#pragma unroll
        for (uint32_t r = 0; r < (kLCodes + 63u) / 64u; r++) {
            const uint32_t nb = run_bits(lv[r], lc[r]), incl = wave_scan_incl(nb);
            run_send(lv[r], lc[r], hdr_bits + incl - nb);
            hdr_bits += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        const uint32_t nb = run_bits(dv[0], dc[0]), incl = wave_scan_incl(nb);
        run_send(dv[0], dc[0], hdr_bits + incl - nb);
        hdr_bits += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    wave_sync();
    if (type == kStatic) write_static_codes();
    if (lane == 0) {
        bo->type = type; bo->hdr_bits = hdr_bits;
        bo->body_bits = type == kStored ? 0u : type == kStatic ? static_len : mem.opt_len - (hdr_bits - 3u);   // opt_len counts header + symbols + EOB
        if (type == kStatic) { bo->eob_len = static_lit_len(256); bo->eob_code = static_lit_code(256); }
        else { bo->eob_len = mem.len[256]; bo->eob_code = bo->lcode[256]; }    // (this lane wrote lcode[256] itself)
    }
    for (uint32_t i = lane; i < kHdrWords; i += 64) bo->hdr[i] = mem.hdr[i];
}

hipError_t launch_plan(const DeflateArgs& a, hipStream_t s) {
    const bool serial = a.plan_serial != 0u;          // the context's option (zwz_ctx_set_option; ZWZ_PLAN is read once, at zwz_ctx_create)
    uint32_t* open_list = reinterpret_cast<uint32_t*>(a.perm) + a.n;          // lz_match's work-order array is dead by now; its first n words are encode's list
    const uint32_t n_blocks = a.n * kMaxBlocks;
    hipLaunchKernelGGL(plan_probe_kernel, dim3(n_blocks), dim3(64), 0, s, a.info, a.blocks, a.probes);
    hipLaunchKernelGGL(plan_cost_kernel, dim3((n_blocks + kCostLanes - 1u) / kCostLanes), dim3(64), 0, s, a.probes, n_blocks, open_list, a.tickets);
    if (serial) hipLaunchKernelGGL(plan_serial_kernel, dim3(n_blocks), dim3(64), 0, s, a.info, a.blocks, a.probes, a.plans);
    else {
        hipLaunchKernelGGL(plan_heap_kernel, dim3((n_blocks + kHeapLanes - 1u) / kHeapLanes), dim3(kHeapThreads), 0, s, a.blocks, a.probes, open_list, a.tickets);
        hipLaunchKernelGGL(plan_kernel, dim3(n_blocks), dim3(64), 0, s, a.info, a.blocks, a.probes, a.plans);
    }
    return hipGetLastError();
}

}  // namespace zwz
