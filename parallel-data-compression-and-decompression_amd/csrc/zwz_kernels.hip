// zwz_kernels.hip -- gfx950 kernels of the chunk codec (the reference's zlib call pair,
// compression.cpp:119-134 and decompression.cpp:16-36, rebuilt as a batch pipeline over HBM).
//
// Compress, per batch of chunks (stage -> intermediate in HBM -> next stage):
//   lz_dense_list / lz_lists         (zwz_band.hip) which chunks are chain-heavy: those go lz_sort -> lz_place -> lz_match_band
//   lz_links    persistent, 1 WG / CU 15-bit hash + newest-first chain links (zlib's head/prev) of the other chunks: eight feeder
//                                     waves hash and stream, an inserter wave walks the LDS head table, one exchange a step
//   lz_match    one WG / chunk        per-position best-of-32 / best-of-128 match records of those chunks; the 32 KiB history
//                                     window (bytes + links) lives in LDS and slides tile by tile
//   lz_parse    one wave / chunk      lazy-evaluation walk over the records, block-parallel -> symbol bit masks
//   blockify    one WG / chunk        symbol ranks (popcount prefix), 16383-symbol block cuts,
//                                     per-block histograms (LDS atomics)
//   plan_probe / plan_cost / plan     stored / static settled by an optimal-Huffman-cost lower bound (wave per
//                                     block sorts, lane per block merges); the rest get zlib-exact trees
//   encode_stored / encode            all-stored chunks copied straight through (one WG / chunk); the rest, persistent, two WGs / CU:
//                                     code lengths -> prefix scan -> bit offsets -> LDS bit packing; Adler-32, 65535-byte truncation
// Decompress:
//   inflate_order / inflate           chunks by payload length, longest first; one wave / chunk: window-parallel Huffman decode,
//                                     byte-parallel copies
// Integrity:
//   md5_files   one lane / file       RFC 1321 over a file's chunk slots (input slots on compress, output slots on decompress)
//
// All stages are integer/byte work bounded by LDS latency and HBM traffic; no MFMA.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "huff_core.h"
#include "inflate_core.h"
#include "lz_core.h"
#include "zwz_md5.h"
#include "zwz_kernels.h"
#include "zwz_device.h"

// (ZWZ_MATCH_EXP & 16, experiment builds only -- tools/match_times.sh: thread 0 of every lz_match workgroup sums the cycles it spent
// per phase -- in registers: an atomic per stamp would be a VMEM operation the next vmcnt(0) waits for, which is what the first
// version of this measured -- and adds them, >> 8, to tickets[40 + phase] at the end; launch_deflate prints them when
// ZWZ_MATCH_TIMES is set.  Thread 0 is in the oldest wave of its SIMD, which the SIMD favours: the other waves' share of a phase shows
// up as its "wait".)
// (ZWZ_MATCH_EXP & 2: a timing experiment whose records are NOT zlib's -- the screening pass's middle-byte gather replaced by a conflict-free read: the
// upper bound for "the inserter's middle byte in the bucket word", measured in round 4; & 1 was "no slide", the bound that led to the ring: DESIGN.md section 8)
#ifndef ZWZ_MATCH_EXP
#define ZWZ_MATCH_EXP 0
#endif
#if ZWZ_MATCH_EXP & 16
#define ZWZ_MSTAMP(ph) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); acc_[(ph)] += (uint32_t)(now_ - stamp_); stamp_ = now_; } while (0)
#else
#define ZWZ_MSTAMP(ph) do { } while (0)
#endif

namespace zwz {

// a chunk's dead link space: its kMaxBlocks BlockProbes first (zwz_plan.hip), its chosen records from kChosenOffset on, the plan stage's merge lists at the end
static __device__ __forceinline__ uint32_t* chosen_of(const uint16_t* links, uint32_t chunk) {
    return reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(const_cast<uint16_t*>(links + (size_t)chunk * kLinkStride)) + kChosenOffset);
}

// ------------------------------------------------------------------------------------------------
// lz_links: link[p] = newest position q < p with hash3(q) == hash3(p), 0 if none (zlib NIL).
// The chain insert is sequential in the position: one wave (the inserter) walks the chunk 64 positions a step with the
// head table in LDS.  A step is ONE LDS operation: with 32-bit buckets, ds_wrxchg_rtn_b32 stores the 64 positions into
// their buckets and hands each lane its predecessor -- lanes that name the same bucket are served in ascending lane
// order (each gets what the nearest lower one wrote, the lowest the old bucket value, the highest lane's position stays),
// which is zlib's insert-in-position-order: no read-back, no collision repair, the same cost on text as on random bytes.
// The ISA manual does not promise that order; zwz_ctx_create checks it on the device (exchange_order_probe_kernel) and
// refuses to create a context where it does not hold.  (The round's earlier kernel -- 16-bit buckets, head read / write /
// read-back per step with collision repair -- took 5.8 ms per 50 000 chunks on random bytes and 35.6 ms on text, where
// most steps hold colliding hashes.)
// A lone wave issues an instruction every ~6-9 cycles, so everything but the exchange is kept off the inserter: eight feeder
// waves (four positions a lane a block) turn input bytes into the bucket ADDRESSES of the next 2048 positions in an LDS buffer and move the finished links
// of the previous 2048 positions out of that buffer to HBM, both as 16-byte vectors; the two roles meet at a barrier
// every 2048 positions (two buffers alternate).  Positions past the last trigram are sent to bucket 0: they are the last
// positions of the chunk, nothing of this chunk reads the table after them (the next chunk's epoch disowns them, below), and
// the feeders zero their links on the way out -- so the inserter needs no validity logic.  The 128 KiB table leaves room for
// one workgroup per CU.
// The compiler's own wait insertion drains to 0 at loop headers and right behind loads it schedules early, which is why the
// inserter's loop and the feeders' input loads are inline asm with counted waits; what keeps that sound:
//   * every step / hand-over issues exactly the same operations in the same order, whatever its lanes hold;
//   * no in-flight register is read, copied or merged before the wait that covers it;
//   * the loops are entered with nothing in flight (inserter) or a known number of operations in flight (feeders).
constexpr uint32_t kLinksBlock = 2048;             // positions per hand-over between the inserter and the feeders
constexpr uint32_t kLinksNoHash = 0;               // bucket of a position without a trigram (any bucket will do, see above)

// Persistent: workgroup w of G (one per CU) links chunks w, w + G, w + 2G, ... as ONE stream of blocks.  The table is cleared
// once: a bucket holds (epoch << 16) | position with epoch = 1 + the chunk's index in this workgroup's sequence, and a
// predecessor that carries another epoch is a leftover of an earlier chunk, i.e. NIL -- the feeders check that on the way
// out, the inserter never knows where a chunk ends.  So the feeders' three cursors (links out: one block behind the
// inserter; bucket addresses: one ahead; input requests: four ahead) simply run on into the next chunk.  (A workgroup per
// chunk paid ~3.5 us per chunk -- dispatch, a 128 KiB clear, an HBM round trip nothing covered, a drained pipeline: a sixth
// of the kernel.)
__global__ __launch_bounds__(kLinksThreads) void lz_links_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                                 const uint32_t* __restrict__ in_len, uint16_t* __restrict__ links,
                                                                 uint32_t* __restrict__ link_stat /* zeroed by the launcher */, uint32_t n_all,
                                                                 const uint32_t* __restrict__ list /* null: every chunk; else the chunks to link ... */,
                                                                 const uint32_t* __restrict__ list_n /* ... and how many there are */) {
    // typed LDS arrays: a generic/volatile pointer here turns every access into a flat_* op with a
    // vmcnt(0) wait behind it (measured: 870 cycles per 64-position step)
    extern __shared__ __attribute__((aligned(16))) uint16_t head[];          // 32768 32-bit buckets + 16 spare bytes
    constexpr uint32_t kHeadBytes = 131072u;
    uint16_t* hbuf = head + (kHeadBytes + 16u) / 2u;                        // 2 x kLinksBlock 32-bit entries: bucket addresses in, links out
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t G = gridDim.x;
    // A cursor: block k of the chunk c = blockIdx.x + j G, the g-th block of this workgroup's stream (the input register sets
    // rotate with g mod 3, the two hand-over buffers alternate with g).  The stream ends on a multiple of three hand-overs;
    // blocks past its end hash to the no-trigram bucket and are never inserted or written out.
    // All of it is wave-uniform (scalar loads, scalar registers).
    struct Cursor { uint32_t c, j, k, g, L, n_blocks; const uint8_t* data; uint16_t* lk; };
    // (with a list -- lz_dense_list's sparse chunks -- the workgroups' stream runs over list entries: c is an index into it)
    const uint32_t n = list ? *list_n : n_all;
    if (blockIdx.x >= n) return;                                             // (more workgroups than chunks: the whole workgroup leaves)
    auto cid = [&](uint32_t c) { return list ? list[c] : c; };
    const uint8_t* const data0 = in + in_off[cid(blockIdx.x)];               // any readable 16 bytes, for requests past the last chunk
    uint16_t* const lk_spare = links + (size_t)cid(blockIdx.x) * kLinkStride + (kLinkStride - 1u);   // position 65 535 does not exist: nobody's link
    auto enter = [&](Cursor& q) {                                            // settle on chunk q.c or the next one with any bytes
        for (;;) {
            if (q.c >= n) { q.L = 0; q.n_blocks = 0xffffffffu; q.data = data0; q.lk = nullptr; return; }   // past the end: stays here
            const uint32_t ch = cid(q.c);
            q.L = in_len[ch];
            q.n_blocks = (q.L + kLinksBlock - 1u) / kLinksBlock;
            q.data = in + in_off[ch];                                        // 16-byte aligned (API contract)
            q.lk = links + (size_t)ch * kLinkStride;
            if (q.n_blocks) return;
            q.c += G; q.j++;
        }
    };
    auto advance = [&](Cursor& q) { q.k++; q.g++; if (q.k >= q.n_blocks) { q.c += G; q.j++; q.k = 0; enter(q); } };

    // ---- feeder: input bytes -> registers (asked for hand-overs ahead, see the main loop) -> bucket addresses in LDS;
    // finished links LDS -> HBM
    constexpr uint32_t kFeeders = kLinksThreads / 64u - 1u;    // feeder waves, each with its own share of every block
    constexpr uint32_t kPer = 32u / kFeeders;         // positions per feeder lane per block
    // Input register sets.  A block's bytes are asked for three hand-overs before their use, by inline asm: left to the
    // compiler, the wait in front of the first use was vmcnt(0/1) -- everything but the link store just issued, including
    // the request made one hand-over ago -- and each hand-over again lasted one HBM round trip.  VMEM operations complete
    // in order, so "the set asked for three trips ago has landed" is a counted wait, provided every trip issues the same
    // operations: load_block always issues its loads (a part past the readable end is not needed -- its positions have no
    // trigram -- and is read from the chunk's start instead), and where a trip has no links to write out it stores zeros
    // over links that are written later (pad_stores).  The sets must also stay where they are while a load is in flight:
    // they only ever pass through unconditional asm statements ("+v"), in loops of their own.
    constexpr uint32_t kAhead = 3;
    static_assert(kPer == 4u, "the feeder below is written for four positions a lane: two 4-byte loads, one 8-byte link store");
    struct InSet { uint32_t c, t; };                   // input bytes [o, o + 4) and [o + 4, o + 8) of a lane's share
    InSet in_ring[kAhead];
#pragma unroll
    for (uint32_t r = 0; r < kAhead; r++) { in_ring[r].c = 0; in_ring[r].t = 0; }
    constexpr uint32_t kLoadOps = 2u;                                           // VMEM operations of a load_block ...
    constexpr uint32_t kStoreOps = 1u;                                          // ... and of a flush_block
    constexpr uint32_t kWait = 2u * (kLoadOps + kStoreOps) + kStoreOps;         // operations younger than the awaited set: two whole trips and this trip's store
    const uint32_t fpos = (wave - 1u) * (kLinksBlock / kFeeders) + lane * kPer;   // this lane's share within a block (feeder waves)
    auto load_block = [&](const Cursor& q, auto slot) { // input bytes [2048 k + fpos, + 8) of the cursor's block -> register set `slot` (= g mod kAhead)
        InSet& S = in_ring[decltype(slot)::value];
        const uint32_t o = q.k * kLinksBlock + fpos;
        const uint32_t Lr = (q.L + 15u) & ~15u;                              // the slot is readable this far
        const uint8_t* pa = q.data + (o + 4u <= Lr ? o : 0u);
        const uint8_t* pt = q.data + (o + 8u <= Lr ? o + 4u : 0u);
        // (the trailing comments put the operands' physical registers into the ISA text: tests/test_generated.py reads them there)
        asm volatile("global_load_dword %0, %2, off ; zwz-feeder load\n\tglobal_load_dword %1, %3, off ; zwz-feeder load" : "+v"(S.c), "+v"(S.t) : "v"(pa), "v"(pt) : "memory");
    };
    auto pad_stores = [&](uint32_t cnt) {          // stores with no effect, where a trip has no links to write out (the counted wait below wants every trip's)
        for (uint32_t i = 0; i < cnt; i++) asm volatile("global_store_short %0, %1, off" :: "v"(lk_spare), "v"(0u) : "memory");
    };
    typedef __attribute__((address_space(3))) uint8_t* lds_byte_ptr;
    const uint32_t head_base = (uint32_t)(uintptr_t)(lds_byte_ptr) reinterpret_cast<uint8_t*>(head);   // 0: this kernel has no static LDS
    uint32_t* hbuf32 = reinterpret_cast<uint32_t*>(hbuf);                       // 32-bit entries: bucket address in, link out
    auto hash_block = [&](const Cursor& q, auto slot) { // bucket addresses of positions [2048 k + fpos, + 4) from the loaded input
        InSet& S = in_ring[decltype(slot)::value];
        asm volatile("s_waitcnt vmcnt(%2) ; zwz-feeder covers %0 %1" : "+v"(S.c), "+v"(S.t) : "n"(kWait) : "memory");
        const uint32_t in_w[2] = {S.c, S.t};
        const uint32_t o = q.k * kLinksBlock + fpos, L = q.L;
        const uint32_t n_ok = L >= o + kMinMatch ? min(kPer, L - o - (kMinMatch - 1u)) : 0u;   // this lane's positions with a trigram (none past the last chunk)
        uint32_t e[kPer];
#pragma unroll
        for (uint32_t i = 0; i < kPer; i++) {
            const uint32_t x = i ? __builtin_amdgcn_alignbyte(in_w[1], in_w[0], i) : in_w[0];
            uint32_t h = hash3(x & 0xffu, (x >> 8) & 0xffu, (x >> 16) & 0xffu);
            if (i >= n_ok) h = kLinksNoHash;
            e[i] = 4u * h + head_base;                 // the bucket's LDS byte address
        }
        *reinterpret_cast<uint4*>(hbuf32 + (q.g & 1u) * kLinksBlock + fpos) = make_uint4(e[0], e[1], e[2], e[3]);
    };
    uint32_t linked = 0;                               // feeder: positions of this lane with a chain predecessor
    auto flush_block = [&](const Cursor& q) {      // links of the cursor's block: LDS -> HBM, zero where there is no trigram
        const uint32_t o = q.k * kLinksBlock + fpos, L = q.L;
        const uint32_t n_ok = L >= o + kMinMatch ? min(kPer, L - o - (kMinMatch - 1u)) : 0u;
        uint4 w = *reinterpret_cast<const uint4*>(hbuf32 + (q.g & 1u) * kLinksBlock + fpos);   // what each exchange returned: (epoch << 16) | predecessor
        const uint32_t ep = q.j + 1u;
        w.x = (w.x >> 16) == ep ? w.x : 0u; w.y = (w.y >> 16) == ep ? w.y : 0u;                    // another chunk's leftovers: NIL
        w.z = (w.z >> 16) == ep ? w.z : 0u; w.w = (w.w >> 16) == ep ? w.w : 0u;
        uint2 v = make_uint2((w.x & 0xffffu) | w.y << 16, (w.z & 0xffffu) | w.w << 16);
        if (n_ok < kPer) {                             // rare: the chunk's last positions
            if (n_ok == 0u) v.x = 0; else if (n_ok == 1u) v.x &= 0xffffu;
            if (n_ok <= 2u) v.y = 0; else if (n_ok == 3u) v.y &= 0xffffu;
        }
        *reinterpret_cast<uint2*>(q.lk + o) = v;
        linked += (uint32_t)((v.x & 0xffffu) != 0) + (uint32_t)((v.x >> 16) != 0) + (uint32_t)((v.y & 0xffffu) != 0) + (uint32_t)((v.y >> 16) != 0);
    };

    // ---- inserter
    typedef __attribute__((address_space(3))) uint8_t* lds_ptr;
    const uint32_t head_a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr) reinterpret_cast<uint8_t*>(head));   // LDS byte addresses
    const uint32_t hbuf_a = head_a + kHeadBytes + 16u;
    // one exchange per step (see the kernel's header)
    auto insert_block = [&](uint32_t k, uint32_t L, uint32_t tag /* epoch << 16 */, uint32_t parity) {
        const uint32_t first = k * kLinksBlock;
        const uint32_t n_steps = __builtin_amdgcn_readfirstlane((min(L, first + kLinksBlock) - first + 63u) / 64u);      // >= 1
        uint32_t base = hbuf_a + 4u * (parity * kLinksBlock + lane);                      // this lane's buffer entry of step s
        uint32_t p0 = tag + first + lane;                                                 // what a bucket holds: epoch and position
        auto step_now = [&](uint32_t slot, uint32_t p) {                                  // one step, start to finish
            uint32_t a, prev;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(slot) : "memory");
            asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(prev) : "v"(a), "v"(p) : "memory");
            asm volatile("ds_write_b32 %0, %1" :: "v"(slot), "v"(prev) : "memory");
        };
        uint32_t s = 0;
        if (n_steps == kLinksBlock / 64u) {
            // A whole block (all but the last of a short chunk), straight-line, steps in PAIRS -- one ds_read2 for two steps' bucket
            // addresses, two exchanges, one ds_write2 for two steps' links -- no loop, no address arithmetic (a pair's slots are a
            // base register plus constant offsets), named registers (a ds_read2 fills a register pair whose halves the exchanges use
            // one by one, which operand constraints cannot express).  Software-pipelined: per pair i, read the addresses of pair
            // i + 2, write the links of pair i - 3, exchange pair i -- three pairs of exchanges in flight (two: 3.09 ms per 50 000 chunks; three:
            // 2.99; four: 2.99).  LDS operations complete in order, so ONE counted wait in front of the write
            // says both "the addresses of pair i are here" and "the exchanges of pair i - 3 are back"; the text is generated,
            // waits included, by tools/gen_links_block.py (which simulates the queue).
            asm volatile(
                "v_add_u32 v116, 0x400, %[base]\n\t"
                "v_add_u32 v117, 0x800, %[base]\n\t"
                "v_add_u32 v118, 0xc00, %[base]\n\t"
                "v_add_u32 v119, 0x1000, %[base]\n\t"
                "v_add_u32 v120, 0x1400, %[base]\n\t"
                "v_add_u32 v121, 0x1800, %[base]\n\t"
                "v_add_u32 v122, 0x1c00, %[base]\n\t"
                "v_mov_b32 v114, %[p0]\n\t"
                "v_add_u32 v115, 0x40, %[p0]\n\t"
                "ds_read2_b32 v[100:101], %[base] offset1:64\n\t"
                "ds_read2_b32 v[102:103], %[base] offset0:128 offset1:192\n\t"
                "ds_read2_b32 v[104:105], v116 offset1:64\n\t"
                "s_waitcnt lgkmcnt(2)\n\t"
                "ds_wrxchg_rtn_b32 v106, v100, v114\n\t"
                "ds_wrxchg_rtn_b32 v107, v101, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[100:101], v116 offset0:128 offset1:192\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "ds_wrxchg_rtn_b32 v108, v102, v114\n\t"
                "ds_wrxchg_rtn_b32 v109, v103, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[102:103], v117 offset1:64\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "ds_wrxchg_rtn_b32 v110, v104, v114\n\t"
                "ds_wrxchg_rtn_b32 v111, v105, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[104:105], v117 offset0:128 offset1:192\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "ds_write2_b32 %[base], v106, v107 offset1:64\n\t"
                "ds_wrxchg_rtn_b32 v112, v100, v114\n\t"
                "ds_wrxchg_rtn_b32 v113, v101, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[100:101], v118 offset1:64\n\t"
                "s_waitcnt lgkmcnt(7)\n\t"
                "ds_write2_b32 %[base], v108, v109 offset0:128 offset1:192\n\t"
                "ds_wrxchg_rtn_b32 v106, v102, v114\n\t"
                "ds_wrxchg_rtn_b32 v107, v103, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[102:103], v118 offset0:128 offset1:192\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "ds_write2_b32 v116, v110, v111 offset1:64\n\t"
                "ds_wrxchg_rtn_b32 v108, v104, v114\n\t"
                "ds_wrxchg_rtn_b32 v109, v105, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[104:105], v119 offset1:64\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "ds_write2_b32 v116, v112, v113 offset0:128 offset1:192\n\t"
                "ds_wrxchg_rtn_b32 v110, v100, v114\n\t"
                "ds_wrxchg_rtn_b32 v111, v101, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[100:101], v119 offset0:128 offset1:192\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "ds_write2_b32 v117, v106, v107 offset1:64\n\t"
                "ds_wrxchg_rtn_b32 v112, v102, v114\n\t"
                "ds_wrxchg_rtn_b32 v113, v103, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[102:103], v120 offset1:64\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "ds_write2_b32 v117, v108, v109 offset0:128 offset1:192\n\t"
                "ds_wrxchg_rtn_b32 v106, v104, v114\n\t"
                "ds_wrxchg_rtn_b32 v107, v105, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[104:105], v120 offset0:128 offset1:192\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "ds_write2_b32 v118, v110, v111 offset1:64\n\t"
                "ds_wrxchg_rtn_b32 v108, v100, v114\n\t"
                "ds_wrxchg_rtn_b32 v109, v101, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[100:101], v121 offset1:64\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "ds_write2_b32 v118, v112, v113 offset0:128 offset1:192\n\t"
                "ds_wrxchg_rtn_b32 v110, v102, v114\n\t"
                "ds_wrxchg_rtn_b32 v111, v103, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[102:103], v121 offset0:128 offset1:192\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "ds_write2_b32 v119, v106, v107 offset1:64\n\t"
                "ds_wrxchg_rtn_b32 v112, v104, v114\n\t"
                "ds_wrxchg_rtn_b32 v113, v105, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[104:105], v122 offset1:64\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "ds_write2_b32 v119, v108, v109 offset0:128 offset1:192\n\t"
                "ds_wrxchg_rtn_b32 v106, v100, v114\n\t"
                "ds_wrxchg_rtn_b32 v107, v101, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "ds_read2_b32 v[100:101], v122 offset0:128 offset1:192\n\t"
                "s_waitcnt lgkmcnt(8)\n\t"
                "ds_write2_b32 v120, v110, v111 offset1:64\n\t"
                "ds_wrxchg_rtn_b32 v108, v102, v114\n\t"
                "ds_wrxchg_rtn_b32 v109, v103, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "s_waitcnt lgkmcnt(7)\n\t"
                "ds_write2_b32 v120, v112, v113 offset0:128 offset1:192\n\t"
                "ds_wrxchg_rtn_b32 v110, v104, v114\n\t"
                "ds_wrxchg_rtn_b32 v111, v105, v115\n\t"
                "v_add_u32 v114, 0x80, v114\n\t"
                "v_add_u32 v115, 0x80, v115\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "ds_write2_b32 v121, v106, v107 offset1:64\n\t"
                "ds_wrxchg_rtn_b32 v112, v100, v114\n\t"
                "ds_wrxchg_rtn_b32 v113, v101, v115\n\t"
                "s_waitcnt lgkmcnt(6)\n\t"
                "ds_write2_b32 v121, v108, v109 offset0:128 offset1:192\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "ds_write2_b32 v122, v110, v111 offset1:64\n\t"
                "s_waitcnt lgkmcnt(2)\n\t"
                "ds_write2_b32 v122, v112, v113 offset0:128 offset1:192\n\t"
                "s_nop 0"

                :
                : [base] "v"(base), [p0] "v"(p0)
                : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122");
            s = n_steps;
        } else if (n_steps >= 6u) {
            // steps s, s+1 issued; bucket addresses of s+2, s+3 at hand; then three steps a trip: read the address of step t+2,
            // exchange for step t, write the link of step t-2 into its slot.  Three LDS operations a step, in order, so "the
            // exchange of t-2 is back" (and with it the address of t, which is older) is lgkmcnt(5).  The wave issues an
            // instruction every ~9 cycles and that is the step's time, so the loop is unrolled to six steps a trip.
            uint32_t g0, g1, g2, q0, q1, q2, p1 = p0 + 64u, p2 = p0 + 128u, t0, t1;
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:768\n\t"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(t0), "=&v"(t1), "=&v"(g2), "=&v"(g0) : "v"(base) : "memory");
            asm volatile("ds_wrxchg_rtn_b32 %0, %2, %4\n\tds_wrxchg_rtn_b32 %1, %3, %5\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1) : "v"(t0), "v"(t1), "v"(p0), "v"(p1) : "memory");
            uint32_t left = __builtin_amdgcn_readfirstlane(n_steps - 2u);
            g1 = 0; q2 = 0;
#define ZWZ_XSTEP(HN, HN_OFF, HC, QC, PPOS, QR, RPOS, ROFF)                                          \
                "ds_read_b32 %[" HN "], %[base] offset:" HN_OFF "\n\t"                                 \
                "s_waitcnt lgkmcnt(5)\n\t"                                                           \
                "ds_wrxchg_rtn_b32 %[" QC "], %[" HC "], %[" PPOS "]\n\t"                            \
                "ds_write_b32 %[base], %[" QR "] offset:" ROFF "\n\t"                                \
                "v_add_u32 %[" RPOS "], 0xc0, %[" RPOS "]\n\t"
            asm volatile(
                "s_cmp_lt_u32 %[left], 6\n\t"
                "s_cbranch_scc1 2f\n\t"
                "1:\n\t"                                                       /* six steps a trip while six are left ... */
                ZWZ_XSTEP("g1", "1024", "g2", "q2", "p2", "q0", "p0", "0")       /* issue s+2, retire s */
                ZWZ_XSTEP("g2", "1280", "g0", "q0", "p0", "q1", "p1", "256")     /* issue s+3, retire s+1 */
                ZWZ_XSTEP("g0", "1536", "g1", "q1", "p1", "q2", "p2", "512")     /* issue s+4, retire s+2 */
                ZWZ_XSTEP("g1", "1792", "g2", "q2", "p2", "q0", "p0", "768")
                ZWZ_XSTEP("g2", "2048", "g0", "q0", "p0", "q1", "p1", "1024")
                ZWZ_XSTEP("g0", "2304", "g1", "q1", "p1", "q2", "p2", "1280")
                "v_add_u32 %[base], 0x600, %[base]\n\t"
                "s_sub_u32 %[left], %[left], 6\n\t"
                "s_cmp_gt_u32 %[left], 5\n\t"
                "s_cbranch_scc1 1b\n\t"
                "2:\n\t"                                                       /* ... then three (the stream of operations stays uniform) */
                "s_cmp_lt_u32 %[left], 3\n\t"
                "s_cbranch_scc1 3f\n\t"
                ZWZ_XSTEP("g1", "1024", "g2", "q2", "p2", "q0", "p0", "0")
                ZWZ_XSTEP("g2", "1280", "g0", "q0", "p0", "q1", "p1", "256")
                ZWZ_XSTEP("g0", "1536", "g1", "q1", "p1", "q2", "p2", "512")
                "v_add_u32 %[base], 0x300, %[base]\n\t"
                "s_sub_u32 %[left], %[left], 3\n\t"
                "3:\n\t"
                "s_waitcnt lgkmcnt(0)"
                : [g0] "+v"(g0), [g1] "+v"(g1), [g2] "+v"(g2), [q0] "+v"(q0), [q1] "+v"(q1), [q2] "+v"(q2),
                  [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [base] "+v"(base), [left] "+s"(left)
                :
                : "scc", "memory");
#undef ZWZ_XSTEP
            // steps s, s+1 (sets 0, 1) are complete and not yet written out
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:256" :: "v"(base), "v"(q0), "v"(q1) : "memory");
            s = n_steps - left;                                                           // first step not issued yet
            base += 512u; p0 = p2;                                                        // p2 = position of step s (two past set 0's)
        }
        for (; s < n_steps; s++) { step_now(base, p0); base += 256u; p0 += 64u; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the feeder reads these links right after the barrier
    };

    // The feeder's input comes from HBM, ~1.5 us away under load, and a hand-over lasts half of that: with the bytes of
    // block k + 2 asked for one hand-over ahead of their use every hand-over waited for them (4.8 of the kernel's 5.4 ms
    // with the inserter switched off).  Three register sets take turns, so a request has three hand-overs to land.
    using Slot0 = std::integral_constant<uint32_t, 0>; using Slot1 = std::integral_constant<uint32_t, 1>; using Slot2 = std::integral_constant<uint32_t, 2>;
    Cursor ld{blockIdx.x, 0, 0, 0, 0, 0, nullptr, nullptr}, hs = ld, fl = ld;   // feeders: input requests, bucket addresses, links out
    if (wave >= 1) {                                // the first requests go out before the table is cleared: one round trip less
        enter(ld); hs = ld; fl = ld;
        load_block(ld, Slot0{}); advance(ld);
        load_block(ld, Slot1{}); advance(ld);
        load_block(ld, Slot2{}); advance(ld);
        pad_stores(3u * kStoreOps);                 // (the counted wait assumes two earlier trips)
    }
    {
        uint4* h4 = reinterpret_cast<uint4*>(head);
        for (uint32_t i = tid; i < (kHeadBytes + 16u) / 16u; i += kLinksThreads) h4[i] = make_uint4(0, 0, 0, 0);   // epoch 0: nobody's
    }
    if (wave >= 1) {
        hash_block(hs, Slot0{}); advance(hs);
        load_block(ld, Slot0{}); advance(ld);
    }
    __syncthreads();
    // One hand-over: the inserter links block g; the feeders write out block g - 1, hash block g + 1 (register set NEXT_) and
    // ask for block g + 4 into the set that has just become free.  It concerns LDS only: __syncthreads() would also drain
    // the feeders' input loads and link stores.  The two roles run their own loops over the same number of hand-overs.
#define ZWZ_LINKS_HANDOVER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);              // the inserter shares its SIMD with two feeder waves and is the critical path
        uint32_t g = 0;
        for (uint32_t c = blockIdx.x, j = 0; c < n; c += G, j++) {
            const uint32_t L = in_len[cid(c)];
            const uint32_t n_blocks = (L + kLinksBlock - 1u) / kLinksBlock;
            for (uint32_t k = 0; k < n_blocks; k++, g++) {
                insert_block(k, L, (j + 1u) << 16, g & 1u);
                ZWZ_LINKS_HANDOVER();
            }
        }
        for (; g % 3u; g++) ZWZ_LINKS_HANDOVER();   // the feeders go three hand-overs a turn
    } else {
        auto chunk_done = [&](const Cursor& q) {    // the feeder waves add up a chunk's linked positions: lz_match picks its work order by it
            for (uint32_t d = 32; d >= 1; d >>= 1) linked += __shfl_xor(linked, d);
            if (lane == 0 && linked) atomicAdd(&link_stat[cid(q.c)], linked);
            linked = 0;
        };
        bool first_trip = true;                     // fl sits on the block the inserter has just finished, except before the first trip
        bool more = fl.c < n;                       // the block the inserter takes up at this hand-over exists
#define ZWZ_LINKS_TRIP(NEXT_)                                                                      \
        if (first_trip) { pad_stores(kStoreOps); first_trip = false; }                             \
        else if (fl.c < n) {                                                                       \
            flush_block(fl);                                                                       \
            if (fl.k + 1u >= fl.n_blocks) chunk_done(fl);                                          \
            advance(fl);                                                                           \
        } else pad_stores(kStoreOps);                                                              \
        more = hs.c < n;                            /* the next hand-over's block */               \
        hash_block(hs, NEXT_{}); advance(hs);                                                      \
        load_block(ld, NEXT_{}); advance(ld);                                                      \
        ZWZ_LINKS_HANDOVER();
        while (more) {                              // three hand-overs a turn, as the inserter counts them; the last of a stream may be idle
            ZWZ_LINKS_TRIP(Slot1)
            ZWZ_LINKS_TRIP(Slot2)
            ZWZ_LINKS_TRIP(Slot0)
        }
#undef ZWZ_LINKS_TRIP
        // The last requests (for blocks past the stream's end) are still in flight, into registers the compiler considers dead from
        // here on: they must land before anything else is computed in them (the flush below built a store's data, or address, in one).
        asm volatile("s_waitcnt vmcnt(0) ; zwz-feeder drains %0 %1 %2 %3 %4 %5" : "+v"(in_ring[0].c), "+v"(in_ring[0].t), "+v"(in_ring[1].c), "+v"(in_ring[1].t), "+v"(in_ring[2].c), "+v"(in_ring[2].t) :: "memory");
        if (!first_trip && fl.c < n) {              // the stream's last block, if its length is a multiple of three (fl is one behind)
            flush_block(fl);
            chunk_done(fl);
        }
    }
#undef ZWZ_LINKS_HANDOVER
}

// lz_core.h's lz_search, restated for a whole wave: same candidates, same order, same records.
// The chain walk is the framework's hottest loop.  Written per lane it compiles to ~32 instructions
// per candidate, a third of them exec-mask bookkeeping for the per-lane exits; a predicated C++
// version fared no better (every wave-uniform test became a select-and-compare pair).  So the
// common path -- fetch the candidate's filter word and its link, compare, count, advance -- is a
// loop of inline asm that runs with EXEC = the lanes still walking and keeps its lane sets in
// SGPR masks; it falls out to C++ (the full comparison) only when enough lanes have a matching
// filter word, or nobody walks any more (see "The walk" below).
// (The filter word is two adjacent aligned dwords -- ONE ds_read2_b32 into a named register pair, which costs the LDS
// ~13 cycles under random addresses against 2 x 10 for two ds_read_b32 (tools/exp/gather_rate.hip) -- and an alignbyte:
// one unaligned ds_read_b32 is legal on gfx950 but is replayed in the LDS pipeline -- it doubled this kernel's time.
// v_alignbyte_b32 looks at the low two bits of its shift operand only, so the byte address itself is the shift.)
// (data / link: lz_match's rings, csrc/zwz_kernels.h; position x sits at index match_ring(x))
static __device__ __forceinline__ uint32_t match_ring(uint32_t x) { return min(x, x - kMatchRing); }   // x < 2 * kMatchRing: x - ring wraps to a huge value below the ring's size
static __device__ __forceinline__ void lz_search_wave(const uint8_t* data, const uint16_t* link, uint32_t p, uint32_t L,
                                                      bool active, uint32_t& e128, uint32_t& e32) {
    typedef __attribute__((address_space(3))) uint8_t* lds_ptr;
    e128 = 0; e32 = 0;
    const uint32_t lane = lane_id();
    const uint32_t pp = match_ring(p);                             // inactive lanes pass any position of the tile
    uint32_t cur = link[pp];
    const bool start = active && p + kMinMatch <= L                // lookahead < 3: not inserted, not searched
                       && cur != 0 && p - cur <= kMaxDist          // first candidate: distance <= MAX_DIST
                       && !(p >= kSlidePos && cur <= kWSize);      // zlib's window has slid: <= 32768 reads as NIL
    uint64_t alive = __builtin_amdgcn_ballot_w64(start);
    if (alive == 0) return;
    if (!start) cur = p;                                           // a readable stand-in
    const uint32_t lookahead = L - p;
    const uint32_t max_len = lookahead < kMaxMatch ? lookahead : kMaxMatch;
    const uint32_t nice = lookahead < kNiceLen ? lookahead : kNiceLen;
    const uint32_t limit = p > kMaxDist ? p - kMaxDist : 0;
    constexpr uint32_t kNone = 0xffffffffu;
    uint32_t best = kMinMatch - 1, best_pos = 0, snap = kNone;
    uint32_t f_off = 0, f_mask = 0xffffffu;                        // see lz_search: the filter word
    const uint32_t scan0 = load_u32(data, pp), scan1 = load_u32(data, pp + 4u);   // the scan's first eight bytes: most full
    uint32_t scan_w = scan0 & f_mask;                                             // comparisons end inside them, without a loop
    const uint32_t data_a = (uint32_t)(uintptr_t)(lds_ptr)const_cast<uint8_t*>(data);                          // LDS byte addresses
    const uint32_t link_a = (uint32_t)(uintptr_t)(lds_ptr) reinterpret_cast<uint8_t*>(const_cast<uint16_t*>(link));
    const uint32_t lbias = __builtin_amdgcn_readfirstlane(link_a);
    uint32_t dbias = data_a + f_off;                               // filter word of candidate c: LDS byte ring(cur) + dbias (f_off <= 255: inside the mirror)
    // The walk.  A lane examines its candidates at its own pace and count (n_l): lanes whose filter word matches are PARKED --
    // they stop walking, candidate and link at hand -- and the full comparison is done for all parked lanes at once when
    // kParkLanes of them are waiting or nobody walks any more.  (Leaving the loop for every single hit -- with a wave-wide
    // candidate count that is what exactness seemed to ask for -- the divergent comparison ran once per 3.7 wave-steps on text for
    // one or two lanes, ~1100 cycles each time: more than half of the walk.)  The loop runs with EXEC = the walking lanes,
    // so parked and finished lanes cost the LDS nothing.  Per lane the sequence of candidates, the counts and every
    // decision are lz_search's: 128 candidates at most, and zlib's short-chain answer (the best after 32, if a 33rd is
    // in range) is taken lazily -- `best` only changes in the comparison, so the first comparison behind the 32nd candidate
    // (or the end) still sees it.
    constexpr uint32_t kParkLanes = 12;
    uint32_t n_l = 0, nxt = cur;                                   // candidates examined by this lane; the link of `cur`
    uint64_t walk = alive, park = 0;
    // (one asm statement: with the SGPR operands captured by reference in a lambda the backend fails with "illegal VGPR to SGPR copy")
    while (walk) {
        uint64_t sv, cont; uint32_t ta, tb, w0, tl, cnt;
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
            "s_and_b64 %[walk], %[walk], %[cont]\n\t"
            "v_cndmask_b32 %[cur], %[cur], %[nxt], %[walk]\n\t"    /* walking lanes step on; parked ones keep their candidate */
            "s_bcnt1_i32_b64 %[cnt], %[park]\n\t"
            "s_cmp_ge_u32 %[cnt], %[npark]\n\t"
            "s_cbranch_scc1 2f\n\t"
            "s_cmp_lg_u64 %[walk], 0\n\t"
            "s_cbranch_scc1 1b\n\t"
            "2:\n\t"
            "s_mov_b64 exec, %[sv]"
            : [cur] "+v"(cur), [nxt] "+v"(nxt), [nl] "+v"(n_l), [walk] "+s"(walk), [park] "+s"(park), [sv] "=&s"(sv), [cont] "=&s"(cont),
              [cnt] "=&s"(cnt), [a] "=&v"(ta), [b] "=&v"(tb), [w0] "=&v"(w0), [l] "=&v"(tl)
            : [dbias] "v"(dbias), [lbias] "s"(lbias), [fmask] "v"(f_mask), [scan] "v"(scan_w), [limit] "v"(limit), [bound] "s"(kMaxChain),
              [npark] "s"(kParkLanes), [ring] "s"(kMatchRing)
            : "vcc", "scc", "memory", "v90", "v91");
        if (park == 0) break;                                      // nobody hit, nobody walks
        bool resume = false;
        if ((park >> lane) & 1ull) {                               // cur = the candidate whose filter word matched, nxt = its link, n_l counts it
            if (n_l > kShortChain && snap == kNone) snap = best >= kMinMatch ? entry_pack(best, p - best_pos) : 0u;   // what zlib's short chain returned
            const uint32_t c_ = match_ring(cur);
            uint32_t x0 = load_u32(data, c_) ^ scan0, x1 = load_u32(data, c_ + 4u) ^ scan1;
            asm volatile("" : "+v"(x0), "+v"(x1));   /* both words now: left alone, the second read is sunk behind a branch on the first */
            const uint32_t l0 = (uint32_t)__builtin_ctz(x0 | 0x80000000u) >> 3, l1 = 4u + ((uint32_t)__builtin_ctz(x1 | 0x80000000u) >> 3);
            uint32_t len = x0 ? l0 : x1 ? l1 : 8u;
            if (len == 8u) len = match_len_from(data, c_, pp, 8u, max_len);
            len = len < max_len ? len : max_len;
            resume = nxt > limit && n_l < kMaxChain;
            if (len > best) {
                best = len; best_pos = cur;
                if (len >= nice) resume = false;
                else { f_off = best - 3u; f_mask = 0xffffffffu; scan_w = load_u32(data, pp + f_off); dbias = data_a + f_off; }
            }
            cur = nxt;
        }
        walk |= __builtin_amdgcn_ballot_w64(resume);
        park = 0;
    }
    if (n_l > kShortChain && snap == kNone) snap = best >= kMinMatch ? entry_pack(best, p - best_pos) : 0u;   // a 33rd candidate was examined: the short chain's answer
    e128 = best >= kMinMatch ? entry_pack(best, p - best_pos) : 0u;
    e32 = snap != kNone ? snap : e128;
    // TOO_FAR: a minimum-length match further than 4096 back is dropped (deflate_slow)
    if (entry_len(e128) == kMinMatch && entry_dist(e128) > kTooFar) e128 = 0;
    if (entry_len(e32) == kMinMatch && entry_dist(e32) > kTooFar) e32 = 0;
}

// ------------------------------------------------------------------------------------------------
// lz_match: per-position match records.  One workgroup walks one chunk tile by tile (16 Ki positions
// per tile) with the history zlib may look at -- bytes and links of the last 32506 positions --
// resident in LDS, as a ring (csrc/zwz_kernels.h: kMatchRing): only the next tile's own 16 KiB of
// bytes + 32 KiB of links come from HBM, fetched into registers while the current tile is being
// searched, and they go where the oldest tile's were (a tile-per-workgroup version re-read the whole
// 147 KB window per tile and spent 58% of its wave-cycles waiting on it; rounds 1-3 slid a linear
// window down after every tile).
__global__ __launch_bounds__(kMatchThreads) void lz_match_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                                 const uint32_t* __restrict__ in_len, const uint16_t* __restrict__ links,
                                                                 uint2* __restrict__ entries, uint64_t* __restrict__ has128,
                                                                 uint16_t* __restrict__ perms, const uint32_t* __restrict__ link_stat,
                                                                 uint32_t band /* 0: every chunk is this kernel's, 1: chain-heavy chunks are lz_match_band's, 2: all are */,
                                                                 uint32_t* __restrict__ mtimes /* experiment builds: phase cycles */) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t chunk = blockIdx.x, tid = threadIdx.x;
    const uint32_t L = in_len[chunk];
    if (L == 0 || band == 2u || (band == 1u && link_stat[chunk] == kDenseMark)) return;
    uint8_t* sdata = smem;
    uint16_t* slink = reinterpret_cast<uint16_t*>(smem + kMatchDataBytes);
    uint32_t* s_has = reinterpret_cast<uint32_t*>(smem + kMatchDataBytes + kMatchLinkBytes);      // 2 KiB: has128 bits of a tile
    uint16_t* s_cnt = reinterpret_cast<uint16_t*>(smem + kMatchDataBytes + kMatchLinkBytes + 2048);   // 4 KiB: bucket counts
    __shared__ uint32_t s_wtot[kMatchThreads / 64];
    __shared__ uint32_t s_grp;
    uint16_t* perm = perms + (size_t)chunk * kTile;
    uint4* sd4 = reinterpret_cast<uint4*>(sdata);
    uint4* sl4 = reinterpret_cast<uint4*>(slink);
    const uint4* gd4 = reinterpret_cast<const uint4*>(in + in_off[chunk]);                    // 16-byte aligned (API contract)
    const uint4* gl4 = reinterpret_cast<const uint4*>(links + (size_t)chunk * kLinkStride);
    uint2* ent = entries + (size_t)chunk * kEntryStride;
    uint64_t* hm = has128 + (size_t)chunk * kMaskWords;
    const uint32_t ntiles = (L + kTile - 1) / kTile;
    const uint32_t dvec_total = (L + 15u) >> 4;     // the slot is readable to L rounded up to 16; bytes past L
                                                    // never influence a result (compares are capped at the lookahead)
    // vectors [dlo, dhi) of data and [llo, lhi) of links are what tile t adds to the window
#define ZWZ_TILE_RANGE(t)                                                                          \
    const uint32_t te_ = min(((t) + 1) * kTile, L), pe_ = (t) ? min((t) * kTile, L) : 0u;          \
    const uint32_t dlo = (t) ? min((pe_ + kMaxMatch + 8u + 15u) >> 4, dvec_total) : 0u;            \
    const uint32_t dhi = min((te_ + kMaxMatch + 8u + 15u) >> 4, dvec_total);                       \
    const uint32_t llo = (pe_ + 7u) >> 3, lhi = (te_ + 7u) >> 3;
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    uint4 pd0 = z4, pd1 = z4, pl0 = z4, pl1 = z4;   // next tile's bytes / links in flight
#define ZWZ_PREFETCH(t)                                                                            \
    {                                                                                              \
        ZWZ_TILE_RANGE(t)                                                                          \
        if (dlo + tid < dhi) pd0 = gd4[dlo + tid];                                                 \
        if (dlo + tid + kMatchThreads < dhi) pd1 = gd4[dlo + tid + kMatchThreads];                 \
        if (llo + tid < lhi) pl0 = gl4[llo + tid];                                                 \
        if (llo + tid + kMatchThreads < lhi) pl1 = gl4[llo + tid + kMatchThreads];                 \
    }
    // chain-heavy data (four of five positions have a chain predecessor: text) takes the sorted work order, sparse
    // data (random bytes: 57 %) the screening pass; lz_links counted while it wrote the links out
    const bool sorted_order = band == 0u && chunk_is_dense(link_stat[chunk], L);  // workgroup-uniform (with the band kernels about, what gets here is sparse)
    for (uint32_t i = tid; i < 512u; i += kMatchThreads) s_has[i] = 0;   // has128 bits of a tile (32-bit words); cleared again as they are written out
#if ZWZ_MATCH_EXP & 16
    uint64_t stamp_ = __builtin_amdgcn_s_memtime();
    uint32_t acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    ZWZ_PREFETCH(0u)
    for (uint32_t t = 0; t < ntiles; t++) {
        const uint32_t ts = t * kTile, te = min(ts + kTile, L);
        {   // registers -> the rings (vector v of the chunk's bytes at ring vector v mod kMatchRing / 16; the ring's first vectors repeated behind its end)
            ZWZ_TILE_RANGE(t)
            auto put_d = [&](uint32_t v, const uint4& x) {
                const uint32_t rv = min(v, v - kMatchRing / 16u);
                sd4[rv] = x;
                if (rv < kMatchMirror / 16u) sd4[kMatchRing / 16u + rv] = x;
            };
            if (dlo + tid < dhi) put_d(dlo + tid, pd0);
            if (dlo + tid + kMatchThreads < dhi) put_d(dlo + tid + kMatchThreads, pd1);
            if (llo + tid < lhi) { const uint32_t u = llo + tid; sl4[min(u, u - kMatchRing / 8u)] = pl0; }
            if (llo + tid + kMatchThreads < lhi) { const uint32_t u = llo + tid + kMatchThreads; sl4[min(u, u - kMatchRing / 8u)] = pl1; }
        }
        __syncthreads();
        ZWZ_MSTAMP(t ? 6 : 0);                      // waiting for the tile's bytes and links, registers -> LDS (first tile: nothing hides the load)
        if (t + 1 < ntiles) ZWZ_PREFETCH(t + 1)     // in flight during the search below

        // Order of work inside the tile.  A group of 64 searches lasts as long as its longest chain, and chain lengths run
        // from 1 to 128 among neighbouring positions (21-29 % lane utilisation on text in natural order).  Chain-heavy chunks
        // sort the tile's positions by a prediction of the chain length and hand out groups of 64 longest first (below);
        // chunks with sparse chains (lz_links' count, above) take the screening pass instead.
        const uint32_t npos = te - ts;
        const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = lane_id();   // (the compiler does not see tid >> 6 as wave-uniform)
        uint32_t nlist = npos;                                     // sorted order: positions on the work list (wave-uniform after the scan)
        if (sorted_order) {
            // The key is a prediction of how many candidates the walk will visit: the chain is followed for up to kKeyDepth
            // links -- a chain that ends before that has exactly that many candidates; otherwise their density over the
            // distance to the kKeyDepth-th one, extended over the window, estimates the rest (capped at zlib's 128) --
            // quantised to 15 half-octave buckets.  A position whose first candidate is out of play (the test
            // lz_search_wave starts with) has no record and is left off the list altogether: key 0.
            // (Offline, on the text corpus: the second-predecessor key of round 1 gave 0.62 lane utilisation in the
            // walk, this one 0.84, the true chain length 0.99; tools/exp/chain_keys.py.)
            constexpr uint32_t kKeyDepth = 8;
            uint64_t mykeys = 0, myrank_lo = 0, myrank_hi = 0;            // 16 positions a thread: 4-bit keys, 8-bit ranks
            auto keys_of4 = [&](uint32_t q0, uint32_t key[4]) {       // positions ts + q0 + j * kMatchThreads: four chains in flight
                uint32_t p[4], cur[4], cnt[4], lim[4], live[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    p[j] = ts + q0 + j * kMatchThreads;
                    const bool ok = p[j] < te && p[j] + kMinMatch <= L;
                    const uint32_t l1 = slink[ok ? match_ring(p[j]) : 0u];
                    live[j] = (uint32_t)(ok && l1 != 0 && p[j] - l1 <= kMaxDist && !(p[j] >= kSlidePos && l1 <= kWSize));
                    cur[j] = live[j] ? l1 : (ok ? p[j] : 0u);          // (out of play: any readable position)
                    cnt[j] = live[j];
                    lim[j] = p[j] > kMaxDist ? p[j] - kMaxDist : 0u;
                }
#pragma unroll
                for (uint32_t st = 1; st < kKeyDepth; st++) {
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++) {
                        const uint32_t nx = slink[match_ring(cur[j])];
                        const uint32_t adv = live[j] & (uint32_t)(nx > lim[j]);
                        cur[j] = adv ? nx : cur[j]; cnt[j] += adv; live[j] = adv;
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    float est = (float)cnt[j];
                    if (cnt[j] == kKeyDepth) {
                        const uint32_t win = p[j] < kMaxDist ? p[j] : kMaxDist;
                        est = fminf(128.0f, (float)(win * kKeyDepth) * __frcp_rn((float)(p[j] - cur[j])));
                        est = fmaxf(est, (float)kKeyDepth);
                    }
                    const uint32_t k = 1u + (uint32_t)(__log2f(fmaxf(est, 1.0f)) * 2.0f);
                    key[j] = cnt[j] ? (k > 15u ? 15u : k) : 0u;
                }
            };
            // One-hot byte per key in four dwords, add-scanned over the wave: a lane's own field of the scan is its rank among
            // the wave's lanes with that key, lane 63's fields are the wave's counts (<= 64: a byte holds them).
            auto wave_key_rank = [&](uint32_t key, uint32_t& rank, uint32_t& count_for_lane) {
                const uint32_t one = key ? 1u << (8u * (key & 3u)) : 0u, word = key >> 2;
                uint32_t inc[4];
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) inc[i] = wave_scan_incl(word == i ? one : 0u);
                const uint32_t mine = word == 0 ? inc[0] : word == 1 ? inc[1] : word == 2 ? inc[2] : inc[3];
                rank = ((mine >> (8u * (key & 3u))) & 0xffu) - 1u;            // (meaningless for key 0: never used)
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)inc[0], 63), t1 = (uint32_t)__builtin_amdgcn_readlane((int)inc[1], 63);
                const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)inc[2], 63), t3 = (uint32_t)__builtin_amdgcn_readlane((int)inc[3], 63);
                const uint32_t lw = (lane >> 2) & 3u;
                const uint32_t tw = lw == 0 ? t0 : lw == 1 ? t1 : lw == 2 ? t2 : t3;
                count_for_lane = (tw >> (8u * (lane & 3u))) & 0xffu;         // lanes 0..15: the wave's count of key `lane`
            };
            // counts[key][trip][wave] -> exclusive scan -> destination of every listed position
            // (rolled loops and packed state on purpose: unrolled, this phase's registers pushed the walk's into scratch and
            // cost config 2 -- which never runs this code -- 3 ms)
#pragma unroll 1
            for (uint32_t kk0 = 0; kk0 < 16u; kk0 += 4u) {
                uint32_t key[4];
                keys_of4(tid + kk0 * kMatchThreads, key);
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t kk = kk0 + j;
                    uint32_t rank, cnt_lane;
                    wave_key_rank(key[j], rank, cnt_lane);
                    mykeys |= (uint64_t)key[j] << (4u * kk);
                    const uint64_t r8 = (uint64_t)(rank & 0xffu) << (8u * (kk & 7u));
                    if (kk0 < 8u) myrank_lo |= r8; else myrank_hi |= r8;
                    if (lane < 16u) s_cnt[(lane * 16u + kk) * 16u + wave] = (uint16_t)cnt_lane;
                }
            }
            __syncthreads();
            {   // exclusive scan of the 4096 counts, four per thread (key 0's are zero)
                const uint2 c4 = *reinterpret_cast<const uint2*>(s_cnt + 4 * tid);
                const uint32_t a = c4.x & 0xffffu, b = c4.x >> 16, c = c4.y & 0xffffu, d = c4.y >> 16;
                const uint32_t own = a + b + c + d;
                const uint32_t incl = wave_scan_incl(own);
                if (lane == 63) s_wtot[wave] = incl;
                __syncthreads();
                uint32_t wbase = 0, total = 0;
                for (uint32_t i = 0; i < kMatchThreads / 64; i++) { const uint32_t w = s_wtot[i]; wbase += i < wave ? w : 0u; total += w; }
                nlist = __builtin_amdgcn_readfirstlane(total);
                if (tid == 0) s_grp = (total + 63u) >> 6;                    // groups of the work list still to be taken (the walk below)
                const uint32_t ex = wbase + incl - own;
                *reinterpret_cast<uint2*>(s_cnt + 4 * tid) = make_uint2(ex | (ex + a) << 16, (ex + a + b) | (ex + a + b + c) << 16);
            }
            __syncthreads();
#pragma unroll 1
            for (uint32_t kk = 0; kk < 16u; kk++) {
                const uint32_t key = (uint32_t)(mykeys >> (4u * kk)) & 15u, rank = (uint32_t)((kk < 8u ? myrank_lo : myrank_hi) >> (8u * (kk & 7u))) & 0xffu;
                if (key) perm[((uint32_t)s_cnt[(key * 16u + kk) * 16u + wave] + rank) & (kTile - 1u)] = (uint16_t)(tid + kk * kMatchThreads);   // (< nlist by construction; masked all the same)
            }
            __syncthreads();
        }
        auto search_and_store = [&](uint32_t p, bool active) {
            uint32_t e128 = 0, e32 = 0;
            lz_search_wave(sdata, slink, p, L, active, e128, e32);
            if (e128) {
                ent[p] = make_uint2(e128, e32);   // e128 == 0 implies e32 == 0; readers gate on has128
                atomicOr(&s_has[(p - ts) >> 5], 1u << (p & 31u));
            }
        };
        if (sorted_order) {
            // Waves take groups of 64 list entries off a counter, longest predicted chains first: dealt out in a fixed order
            // (16 groups a wave) the waves finished up to a fifth of the walk apart -- a SIMD favours its oldest wave.
            for (;;) {
                uint32_t g = 0;
                if (lane == 0) g = atomicSub(&s_grp, 1u);
                g = __builtin_amdgcn_readfirstlane(g);
                if ((int32_t)g <= 0) break;
                const uint32_t idx = (g - 1u) * 64u + lane;
                search_and_store(idx < nlist ? ts + (uint32_t)perm[idx] : ts, idx < nlist);
            }
        } else {
            // Sparse tile (incompressible data: a position has one predecessor in range on average, most
            // have none or one, a few have five).  In natural order a wave's trip lasts as long as its
            // longest chain and most lanes idle (12% VALU lane utilisation on random bytes).  So a wave
            // first screens its 1024 positions, four trips' reads in flight at a time:
            //   no candidate in range              -> no record, done;
            //   one candidate, trigram differs     -> no record, done (the walk would end on it);
            //   anything else                      -> onto the wave's work list in LDS,
            // and then runs the full search over the list, 64 entries a trip.  The list is bounded
            // (kListCap entries): it is drained early whenever the next four trips might not fit.
            constexpr uint32_t kListCap = kMatchListBytes / 2u / (kMatchThreads / 64u);      // 432
            uint16_t* wl = s_cnt + wave * kListCap;
            uint32_t nl = 0;                                                               // wave-uniform
            auto drain = [&](bool all) {                                 // all: the tile is over; else whole trips only, the rest waits
                ZWZ_MSTAMP(1);                                           // screening so far
                // The list is refined in place before it is searched, 64 entries a trip with every lane busy: an entry whose
                // first two candidates both differ from it in the trigram's middle byte and have no third behind them cannot match (the
                // walk would end on them with nothing found) -- on random bytes that is three entries in four, at a sixth
                // of the cost of a search trip.  Survivors are packed to the front (a trip reads its entries before it writes).
                const uint32_t n_proc = all ? nl : nl & ~63u;               // (a ragged last trip ran a quarter of its lanes)
                uint32_t ns = 0;
                for (uint32_t i = lane; i - lane < n_proc; i += 64u) {
                    const bool valid = i < n_proc;
                    const uint32_t q = valid ? (uint32_t)wl[i] : 0u, p = ts + q, pi = match_ring(p);
                    const uint32_t floor1 = max(p + 1u, kMaxDist + 1u) - kMaxDist;               // a link >= this is a candidate in range
                    const uint32_t scan = sdata[pi + 1u];                                         // (one byte again: see the screening pass)
                    const uint32_t l1 = slink[pi];
                    const uint32_t li1 = valid ? match_ring(l1) : pi;                             // listed: its first candidate is in range
                    const uint32_t l2 = slink[li1], cw1 = sdata[li1 + 1u];
                    const bool in2 = l2 >= floor1;
                    const uint32_t li2 = in2 ? match_ring(l2) : pi;
                    const uint32_t l3 = slink[li2], cw2 = sdata[li2 + 1u];
                    uint32_t v = l3 >= floor1 ? scan : cw2;
                    v = in2 ? v : cw1;
                    v = cw1 == scan ? scan : v;
                    const bool keep = valid & (v == scan);
                    const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
                    if (keep) wl[ns + rank_in(m)] = (uint16_t)q;
                    ns += (uint32_t)__popcll(m);
                }
                ZWZ_MSTAMP(2);                                           // the list's refinement
                for (uint32_t i = lane; i - lane < ns; i += 64u) search_and_store(i < ns ? ts + (uint32_t)wl[i] : ts, i < ns);
                ZWZ_MSTAMP(3);                                           // the searches
                const uint32_t rem = nl - n_proc;                            // < 64 entries go to the front and wait for company
                const uint32_t keepq = lane < rem ? (uint32_t)wl[n_proc + lane] : 0u;
                if (lane < rem) wl[lane] = (uint16_t)keepq;
                nl = rem;
            };
            // A lane screens four consecutive positions a trip: their links are one 8-byte read, their trigrams come out of
            // two words.  Tests are selects on compare masks -- no flag words, no short-circuit logic (as bool tests and
            // guarded reads this pass was mostly exec-mask bookkeeping on the CU's one scalar unit) -- and "no candidate"
            // in all its forms (NIL, too far, no trigram, behind zlib's slid window) is one comparison: link >= floor, with
            // floor(p) = max(1, p - MAX_DIST); a second candidate is in range iff link2 > limit(p), i.e. link2 >= floor(p + 1).
            const uint32_t last_ok = L >= kMinMatch ? min(te, L - (kMinMatch - 1u)) : 0u;
            const uint32_t n_ok = __builtin_amdgcn_readfirstlane(last_ok > ts ? last_ok - ts : 0u);   // the tile's positions with a trigram
            for (uint32_t t4 = 0; t4 < 4u; t4++) {
                const uint32_t qb = t4 * 4096u + wave * 256u;                              // wave-uniform; the waves interleave: later positions have more history
                if (qb >= n_ok) break;
                if (nl + 256u > kListCap) drain(false);
                const uint32_t q0 = qb + lane * 4u, p0 = ts + q0, wi = match_ring(p0);      // wi: ring index, a multiple of 4 (the ring's size is one of 16: the four positions do not straddle its end)
                const uint2 lk2 = *reinterpret_cast<const uint2*>(slink + wi);
                const uint32_t w0 = reinterpret_cast<const uint32_t*>(sdata + wi)[0], w1 = reinterpret_cast<const uint32_t*>(sdata + wi)[1];
                uint32_t l1[4] = {lk2.x & 0xffffu, lk2.x >> 16, lk2.y & 0xffffu, lk2.y >> 16};
                if (qb + 256u > n_ok) {                                                   // the chunk's last positions
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++) if (q0 + j >= n_ok) l1[j] = 0;
                }
                if (ts + qb + 256u > kSlidePos) {                                         // zlib's window has slid: <= 32768 reads as NIL
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++) if (p0 + j >= kSlidePos && l1[j] <= kWSize) l1[j] = 0;
                }
                // (the kernel's time is the LDS pipe's: a gather costs ~8 cycles of bank conflicts per dword, so the lone
                // candidate is tested on ONE byte, the trigram's middle one -- with equal hashes it passes 1 in 64 candidates
                // that differ, and what it lets through the list's refinement and the search look at in full)
                uint32_t floor_[5], mid[4], l2[4], cb[4], okm[4];
#pragma unroll
                for (uint32_t j = 0; j < 5; j++) floor_[j] = max(p0 + j, kMaxDist + 1u) - kMaxDist;
                mid[0] = (w0 >> 8) & 0xffu; mid[1] = (w0 >> 16) & 0xffu; mid[2] = w0 >> 24; mid[3] = w1 & 0xffu;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    okm[j] = l1[j] >= floor_[j] ? 0xffffffffu : 0u;
                    const uint32_t li = okm[j] ? match_ring(l1[j]) : wi + j;              // a readable stand-in for positions out of play
                    l2[j] = slink[li];
                    cb[j] = (ZWZ_MATCH_EXP & 2) ? sdata[wi + j + 65u] : sdata[li + 1u];     // (& 2, timing only: the lone candidate's middle byte without its gather)
                }
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    uint32_t v = l2[j] >= floor_[j + 1] ? mid[j] : cb[j];                 // a second candidate, or the only one may share the trigram
                    v = okm[j] ? v : 0xffffffffu;
                    const bool push = v == mid[j];
                    const uint64_t m = __builtin_amdgcn_ballot_w64(push);
                    if (push) wl[nl + rank_in(m)] = (uint16_t)(q0 + j);
                    nl += (uint32_t)__popcll(m);
                }
            }
            drain(true);
        }
        ZWZ_MSTAMP(1);
        __syncthreads();
        ZWZ_MSTAMP(4);                              // waiting for the other waves
        for (uint32_t i = tid; i < ((npos + 63u) >> 6); i += kMatchThreads) {
            hm[(ts >> 6) + i] = (uint64_t)s_has[2 * i] | ((uint64_t)s_has[2 * i + 1] << 32);
            s_has[2 * i] = 0; s_has[2 * i + 1] = 0;                        // ready for the next tile (whoever read a word clears it)
        }
        if (t + 1 == ntiles) break;

        // (no slide: the next tile's bytes and links go where the oldest tile's were -- at the top of the loop, behind the barrier above)
        ZWZ_MSTAMP(5);                              // has128 out
    }
#undef ZWZ_TILE_RANGE
#undef ZWZ_PREFETCH
#if ZWZ_MATCH_EXP & 16
    if (tid == 0) for (uint32_t ph = 0; ph < 8; ph++) atomicAdd(&mtimes[ph], acc_[ph] >> 8);
#endif
}

// ------------------------------------------------------------------------------------------------
// lz_parse: the lazy-match walk, one wave per chunk, 64 positions per step, no sequential walk:
//   1. every lane computes the transition of "fresh at my position" (lz_core.h fresh_step);
//   2. the walk's nodes inside the block are the orbit of the block's entry position under
//      "next fresh": pointer doubling over 64 lanes, at most 6 rounds, early exit;
//   3. the fresh lanes scatter bits into 8-word LDS rings (a jump reaches at most 5 blocks ahead):
//      entry bit of a later block, match start, e32 selector, first interior position of the match;
//   4. a position is covered by a match iff the latest event at or before it is an interior start
//      (events: interior starts and fresh positions); symbols = the rest.
// (A scalar walk cost ~150 vector instructions per symbol: 81 ms for 50k text chunks.)
// (ZWZ_PARSE_EXP & 16, experiment builds only -- tools/parse_times.sh: every wave sums the cycles per phase in registers)
#ifndef ZWZ_PARSE_EXP
#define ZWZ_PARSE_EXP 0
#endif
#if ZWZ_PARSE_EXP & 16
__device__ unsigned long long g_parse_times[8];
#define ZWZ_PSTAMP(ph) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); pacc_[(ph)] += (uint32_t)(now_ - pstamp_); pstamp_ = now_; } while (0)
#else
#define ZWZ_PSTAMP(ph) do { } while (0)
#endif
struct ParseWaveMem {
    uint2 win[128];            // records of positions & 127 (this block and the next)
    uint32_t ring_r[16], ring_s[16], ring_m[16], ring_m32[16];   // 8 x 64-bit words each, as halves
    uint8_t flag[64];
};

__global__ __launch_bounds__(kParseThreads) void lz_parse_kernel(const uint32_t* __restrict__ in_len, uint32_t n,
                                                                const uint2* __restrict__ entries, const uint64_t* __restrict__ has128,
                                                                uint64_t* __restrict__ sym, uint64_t* __restrict__ mst,
                                                                ChunkInfo* __restrict__ info, uint16_t* __restrict__ links,
                                                                const uint32_t* __restrict__ parsed_marks /* link_stat: kDenseMark = lz_lazy has parsed this chunk; or null */) {
    __shared__ ParseWaveMem s_mem[kParseThreads / 64];
    const uint32_t chunk = blockIdx.x * (kParseThreads / 64) + (threadIdx.x >> 6);
    if (chunk >= n) return;
    if (parsed_marks && parsed_marks[chunk] == kDenseMark) return;
    ParseWaveMem& m = s_mem[threadIdx.x >> 6];
    const uint32_t lane = lane_id();
    const uint2* ent = entries + (size_t)chunk * kEntryStride;
    const uint64_t* hm = has128 + (size_t)chunk * kMaskWords;
    uint64_t* gsym = sym + (size_t)chunk * kMaskWords;
    uint64_t* gmst = mst + (size_t)chunk * kMaskWords;
    const uint32_t L = in_len[chunk];
    const uint32_t nwords = (L + 63) >> 6;

    if (lane < 16) { m.ring_r[lane] = lane == 0 ? 1u : 0u; m.ring_s[lane] = 0; m.ring_m[lane] = 0; m.ring_m32[lane] = 0; }
    // records exist only where lz_match found something: gate every lane on its has128 bit
    uint64_t hw_pre = 0;                             // has128 word of the window in `pre`
    // the has128 words of 64 consecutive blocks sit one per lane and are handed out by v_readlane: a record fetch is
    // then ONE global round trip (reading the block's mask word first made it two dependent ones, and with one block of
    // lookahead the second was not covered by a block's work: 5.9 k cycles per block on text)
    uint32_t hm_base = 0;
    uint64_t hm_l = lane < nwords ? hm[lane] : 0ull;
    auto hm_word = [&](uint32_t w) -> uint64_t {     // w wave-uniform, < nwords
        w = __builtin_amdgcn_readfirstlane(w);
        if (w - hm_base >= 64u) { hm_base = w & ~63u; hm_l = hm_base + lane < nwords ? hm[hm_base + lane] : 0ull; }
        const uint32_t i = w - hm_base;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)hm_l, (int)i);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(hm_l >> 32), (int)i);
        return (uint64_t)lo | ((uint64_t)hi << 32);
    };
    auto fetch = [&](uint32_t w) -> uint2 {          // window w (64 records), zeros past the chunk
        uint2 e = make_uint2(0u, 0u);
        hw_pre = 0;
        if (w < nwords) { hw_pre = hm_word(w); if ((hw_pre >> lane) & 1ull) e = ent[(w << 6) + lane]; }
        return e;
    };
    // (A second block of lookahead -- records asked for two blocks before they go into the window -- changes nothing as long as
    // the compiler's wait in front of the window write is vmcnt(0): it waits for the younger request as well.  The wait for the
    // records is 21 % of the kernel on text and 46 % on 7 KB chunks, ZWZ_PARSE_EXP; counted waits would need every block to issue
    // the same number of loads and stores.)
    uint2 pre = fetch(0);
    m.win[lane] = pre;
    uint64_t hw_cur = hw_pre;                        // has128 word of the block being processed
    pre = fetch(1);
    uint32_t carry_open = 0, n_sym = 0, last_is_match = 0;
    uint32_t* chosen = chosen_of(links, chunk);      // the chosen record of every match, compact, in stream order
    uint32_t n_match = 0;                            // wave-uniform
    auto lookup = [&](uint32_t p, uint32_t sel) -> uint32_t { const uint2 e = m.win[p & 127u]; return sel ? e.y : e.x; };

#if ZWZ_PARSE_EXP & 16
    uint64_t pstamp_ = __builtin_amdgcn_s_memtime();
    uint32_t pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (uint32_t blk = 0; blk < nwords; blk++) {
        const uint32_t base = blk << 6, q = base + lane;
        ZWZ_PSTAMP(5);                                   // the tail of the block before: outputs, ring reset
        {   // run of literal-only blocks entered at a block start with nothing pending: settle up to 64
            // blocks at once (incompressible data is almost all such runs)
            const uint32_t slot = 2 * (blk & 7u);
            const uint32_t quiet = (m.ring_r[slot] == 1u) & ((m.ring_r[slot + 1] | m.ring_s[slot] | m.ring_s[slot + 1] |
                                                              m.ring_m[slot] | m.ring_m[slot + 1]) == 0u);
            if (hw_cur == 0 && carry_open == 0 && __builtin_amdgcn_readfirstlane(quiet)) {
                const uint32_t w = blk + lane;
                const uint64_t word = w < nwords ? hm[w] : ~0ull;
                const uint64_t zeros = __ballot(word == 0 && ((w + 1u) << 6) <= L);
                const uint32_t run = zeros == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~zeros);
                if (run >= 2u) {
                    if (lane < run) { gsym[w] = ~0ull; gmst[w] = 0; }
                    n_sym += 64u * run;
                    const uint32_t nb = blk + run;
                    if (lane == 0) { m.ring_r[slot] = 0; if ((nb << 6) < L) m.ring_r[2 * (nb & 7u)] |= 1u; }
                    pre = fetch(nb);                 // re-prime the window pipeline at block nb
                    m.win[((nb << 6) + lane) & 127u] = pre;
                    hw_cur = hw_pre;
                    pre = fetch(nb + 1u);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    blk = nb - 1u;
                    continue;
                }
            }
        }
        m.win[(q + 64u) & 127u] = pre;               // window blk + 1 -> LDS, window blk + 2 in flight
        const uint64_t hw_blk = hw_cur;
        hw_cur = hw_pre;
        pre = fetch(blk + 2);
        const bool valid = q < L;
        {   // literal-only block entered at its first position with nothing pending: no walk to do
            const uint32_t slot = 2 * (blk & 7u);
            const uint32_t quiet = (m.ring_r[slot] == 1u) & ((m.ring_r[slot + 1] | m.ring_s[slot] | m.ring_s[slot + 1] |
                                                              m.ring_m[slot] | m.ring_m[slot + 1]) == 0u);
            if (hw_blk == 0 && carry_open == 0 && __builtin_amdgcn_readfirstlane(quiet)) {
                const uint64_t vm = __ballot(valid);
                n_sym += (uint32_t)__popcll(vm);
                if (lane == 0) {
                    gsym[blk] = vm; gmst[blk] = 0;
                    m.ring_r[slot] = 0;
                    if (base + 64u < L) m.ring_r[2 * ((blk + 1u) & 7u)] |= 1u;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                continue;
            }
        }
        m.flag[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        ZWZ_PSTAMP(0);                                   // quiet-block tests, the window's next records (fetch issued, previous written to LDS)
        FreshStep st{q + 1, q, 0u, 1u};
        if (valid) st = fresh_step(lookup, q, L);
        ZWZ_PSTAMP(1);                                   // every lane's step
        uint32_t succ = st.next >= base + 64u ? 64u : st.next - base;      // 64 = leaves the block
        const uint64_t entry = ((uint64_t)m.ring_r[2 * (blk & 7u)] | ((uint64_t)m.ring_r[2 * (blk & 7u) + 1] << 32));
        const uint64_t validm = __ballot(valid);
        uint64_t marks = entry & validm;
        if (marks) {
            const uint64_t nonlit = __ballot(valid && !st.is_lit);
            const uint32_t e0 = (uint32_t)__builtin_ctzll(marks);
            if ((nonlit >> e0) == 0) {
                marks = validm & ~((1ull << e0) - 1ull);                 // only literals from the entry on
            } else {
                // (Round 5, measured and dropped: the orbit by scalar code that hops only at MATCHES -- from where it stands the walk takes every
                // position up to the next lane that is not a literal (find-first-bit on the ballot, the run marked with two shifts) and goes on
                // where that lane's match ends (v_readlane): five jumps a block on text, ~14 scalar instructions each, nothing for the vector
                // unit.  Exact (all oracle tests); text 13.7 -> 17.8 ms, 370 000 image-like files 23.0 -> 31.7 ms: a jump is a chain of dependent
                // scalar issues, each a turn among the SIMD's waves, and the doubling's rounds are fewer turns than the jumps'.)
                // (Round 4, measured and dropped again: the orbit by a scalar hop -- v_readlane + a few scalar instructions a symbol, as in
                // inflate -- for the chunks lz_dense_list marks chain-heavy: text 15.3 -> 16.9 ms, 7 KB image-like files 6.7 -> 11.4 ms per
                // 100 000 chunks: a block of mostly literals is sixty hops, and six rounds of doubling cost less than fifteen hops.)
                for (uint32_t round = 0; round < 6; round++) {
                    if (((marks >> lane) & 1ull) && succ < 64u) m.flag[succ] = 1;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const uint64_t now = marks | __ballot(m.flag[lane] != 0);
                    const uint32_t s2 = __shfl(succ, succ & 63u);
                    succ = succ < 64u ? s2 : 64u;
                    if (now == marks) break;                             // orbit complete
                    marks = now;
                }
                marks &= validm;
            }
        }
        ZWZ_PSTAMP(2);                                   // the orbit of the block's entry
        const bool fresh = (marks >> lane) & 1ull;
        if (fresh) {
            const uint32_t nx = st.next;
            if (nx >= base + 64u && nx < L) atomicOr(&m.ring_r[((nx >> 5) & 15u)], 1u << (nx & 31u));
            if (!st.is_lit) {
                const uint32_t mp = st.mpos, sp = st.mpos + 1u;
                atomicOr(&m.ring_m[(mp >> 5) & 15u], 1u << (mp & 31u));
                if (st.sel) atomicOr(&m.ring_m32[(mp >> 5) & 15u], 1u << (mp & 31u));
                atomicOr(&m.ring_s[(sp >> 5) & 15u], 1u << (sp & 31u));
            }
        }
        last_is_match |= (uint32_t)(__ballot(fresh && !st.is_lit && st.next == L) != 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        ZWZ_PSTAMP(3);                                   // events into the rings
        const uint32_t slot = 2 * (blk & 7u);
        const uint64_t S = (uint64_t)m.ring_s[slot] | ((uint64_t)m.ring_s[slot + 1] << 32);
        const uint64_t M = (uint64_t)m.ring_m[slot] | ((uint64_t)m.ring_m[slot + 1] << 32);
        const uint64_t M32 = (uint64_t)m.ring_m32[slot] | ((uint64_t)m.ring_m32[slot + 1] << 32);
        const uint64_t X = S | marks;
        const uint64_t upto = X & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        const bool open = upto ? ((S >> (63u - (uint32_t)__builtin_clzll(upto))) & 1ull) : (carry_open != 0);
        const uint64_t cover = __ballot(open);
        carry_open = (uint32_t)(cover >> 63);
        const uint64_t sym_w = ~cover & validm;
        n_sym += (uint32_t)__popcll(sym_w);
        if (lane == 0) { gsym[blk] = sym_w; gmst[blk] = M; }
        ZWZ_PSTAMP(4);                                   // cover, symbol word
        if (M) {                                     // this block's match starts are final: their records go out, selected and compact
            if ((M >> lane) & 1ull) {
                const uint2 e = m.win[q & 127u];
                // (a chunk has fewer than kChosenCap matches -- each covers three positions -- but the index is record-derived:
                // clamped, so that no input, however wrong upstream, can reach past the chunk's own link space)
                chosen[min(n_match + rank_in(M), kChosenCap - 1u)] = ((M32 >> lane) & 1ull) ? e.y : e.x;
            }
            n_match += (uint32_t)__popcll(M);
        }
        if (lane < 2) { m.ring_r[slot + lane] = 0; m.ring_s[slot + lane] = 0; m.ring_m[slot + lane] = 0; m.ring_m32[slot + lane] = 0; }
    }
    if (lane == 0) {
        ChunkInfo ci;
        ci.n_sym = n_sym;
        const uint32_t s_in = (n_sym > 0 && !last_is_match) ? n_sym - 1 : n_sym;
        ci.n_blocks = s_in / kSymsPerBlock + 1;
        info[chunk] = ci;
    }
#if ZWZ_PARSE_EXP & 16
    if (lane == 0) for (uint32_t ph = 0; ph < 8; ph++) atomicAdd(&g_parse_times[ph], (unsigned long long)(pacc_[ph] >> 8));
#endif
}

// ------------------------------------------------------------------------------------------------
// blockify: symbol ranks, block cuts every 16383 symbols, per-block histograms.
static __device__ __forceinline__ uint32_t select_bit(uint64_t w, uint32_t k) {  // position of k-th (0-based) set bit
    for (uint32_t i = 0; i < k; i++) w &= w - 1;
    return (uint32_t)__builtin_ctzll(w);
}

__global__ __launch_bounds__(kBlockifyThreads) void blockify_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                                   const uint32_t* __restrict__ in_len, const uint2* __restrict__ entries,
                                                                   const uint64_t* __restrict__ sym, const uint64_t* __restrict__ mst,
                                                                   const ChunkInfo* __restrict__ info,
                                                                   BlockInfo* __restrict__ blocks, const uint16_t* __restrict__ links) {
    __shared__ uint64_t s_sym[kMaskWords];
    __shared__ uint32_t s_rank[kMaskWords + 1];   // symbols before word w
    __shared__ uint16_t s_mrank[kMaskWords];      // matches before word w (< 21 846)
    __shared__ uint32_t s_msum[kBlockifyThreads / 64];
    __shared__ uint32_t s_hist[kMaxBlocks][kLCodes + kDCodes + 4];
    __shared__ uint32_t s_wsum[kBlockifyThreads / 64];
    __shared__ uint32_t s_start[kMaxBlocks + 1], s_flush[kMaxBlocks];
    const uint32_t chunk = blockIdx.x, tid = threadIdx.x;
    const uint32_t L = in_len[chunk];
    const uint32_t nwords = (L + 63) >> 6;
    const ChunkInfo ci = info[chunk];
    const uint64_t* gsym = sym + (size_t)chunk * kMaskWords;
    const uint64_t* gmst = mst + (size_t)chunk * kMaskWords;

    for (uint32_t i = tid; i < kMaxBlocks * (kLCodes + kDCodes + 4); i += kBlockifyThreads) (&s_hist[0][0])[i] = 0;
    if (tid <= kMaxBlocks) s_start[tid] = L;
    if (tid < kMaxBlocks) s_flush[tid] = L;
    // popcount prefix over the 1024 mask words: 4 words per thread
    constexpr uint32_t kPer = kMaskWords / kBlockifyThreads;
    uint64_t w[kPer]; uint32_t c[kPer], tsum = 0;
    for (uint32_t k = 0; k < kPer; k++) {
        uint32_t wi = tid * kPer + k;
        w[k] = wi < nwords ? gsym[wi] : 0ull;
        s_sym[wi] = w[k];
        c[k] = (uint32_t)__popcll(w[k]); tsum += c[k];
    }
    uint32_t cm[kPer], msum = 0;                  // the same prefix over the match-start mask: a match's index into `chosen`
    for (uint32_t k = 0; k < kPer; k++) { const uint32_t wi = tid * kPer + k; cm[k] = wi < nwords ? (uint32_t)__popcll(gmst[wi]) : 0u; msum += cm[k]; }
    const uint32_t incl = wave_scan_incl(tsum), mincl = wave_scan_incl(msum);
    if (lane_id() == 63) { s_wsum[tid >> 6] = incl; s_msum[tid >> 6] = mincl; }
    __syncthreads();
    uint32_t wbase = 0, mbase = 0;
    for (uint32_t i = 0; i < (tid >> 6); i++) { wbase += s_wsum[i]; mbase += s_msum[i]; }
    uint32_t run = wbase + incl - tsum, mrun = mbase + mincl - msum;
    for (uint32_t k = 0; k < kPer; k++) { s_rank[tid * kPer + k] = run; run += c[k]; s_mrank[tid * kPer + k] = (uint16_t)mrun; mrun += cm[k]; }
    if (tid == kBlockifyThreads - 1) s_rank[kMaskWords] = run;
    __syncthreads();

    // block cuts: symbol #16383*b starts block b; the flush of block b-1 follows the loop-top
    // after its last symbol's start position
    for (uint32_t k = 0; k < kPer; k++) {
        const uint32_t wi = tid * kPer + k, r0 = s_rank[wi], r1 = r0 + c[k];
        for (uint32_t b = 1; b < ci.n_blocks; b++) {
            const uint32_t first = b * kSymsPerBlock, lastprev = first - 1;
            if (first >= r0 && first < r1) s_start[b] = (wi << 6) + select_bit(w[k], first - r0);
            if (lastprev >= r0 && lastprev < r1) s_flush[b - 1] = (wi << 6) + select_bit(w[k], lastprev - r0) + 1;
        }
    }
    if (tid == 0) s_start[0] = 0;
    __syncthreads();

    // histograms: a thread takes 16 consecutive positions at a time -- one 16-byte load of input and
    // a 16-bit slice of each mask -- four such groups per trip with every load of the trip issued
    // before the first use.  (One byte per thread per load left the loop waiting out ~200 global
    // round trips per chunk.)  A position's block is the number of block starts at or before it.
    const uint8_t* data = in + in_off[chunk];
    const uint32_t* chosen = chosen_of(links, chunk);
    const uint4* d4 = reinterpret_cast<const uint4*>(data);          // readable up to L rounded up to 16 (API contract)
    const uint32_t c1 = ci.n_blocks > 1 ? s_start[1] : 0xffffffffu, c2 = ci.n_blocks > 2 ? s_start[2] : 0xffffffffu;
    const uint32_t c3 = ci.n_blocks > 3 ? s_start[3] : 0xffffffffu, c4 = ci.n_blocks > 4 ? s_start[4] : 0xffffffffu;
    auto block_of = [&](uint32_t p) { return (uint32_t)(p >= c1) + (uint32_t)(p >= c2) + (uint32_t)(p >= c3) + (uint32_t)(p >= c4); };
    const uint32_t ngroups = (L + 15u) >> 4;
    for (uint32_t g0 = tid; g0 < ngroups; g0 += 4 * kBlockifyThreads) {
        uint4 bytes[4]; uint32_t lits[4], mats[4]; uint64_t mword[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t g = g0 + u * kBlockifyThreads;
            const bool in = g < ngroups;
            const uint32_t wi = in ? g >> 2 : 0u, sh = (g & 3u) * 16u;
            bytes[u] = in ? d4[g] : make_uint4(0, 0, 0, 0);
            const uint32_t sy = in ? (uint32_t)(s_sym[wi] >> sh) & 0xffffu : 0u;
            mword[u] = in ? gmst[wi] : 0ull;
            const uint32_t mt = (uint32_t)(mword[u] >> sh) & 0xffffu;
            mats[u] = sy & mt; lits[u] = sy & ~mt;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t pg = (g0 + u * kBlockifyThreads) << 4;
            const uint32_t b_lo = block_of(pg), b_hi = block_of(pg + 15u);
            const uint32_t wd[4] = {bytes[u].x, bytes[u].y, bytes[u].z, bytes[u].w};
            if (__ballot(lits[u] != 0xffffu || b_lo != b_hi) == 0) {
                // every lane: sixteen literals of one block (incompressible data, nearly always) -- no per-position tests,
                // which are sixteen rounds of exec-mask bookkeeping on the scalar unit
                uint32_t* h = s_hist[b_lo];
#pragma unroll
                for (uint32_t j = 0; j < 16; j++) atomicAdd(&h[(wd[j >> 2] >> ((j & 3u) * 8u)) & 0xffu], 1u);
            } else if (lits[u]) {
#pragma unroll
                for (uint32_t j = 0; j < 16; j++) {
                    if (!((lits[u] >> j) & 1u)) continue;
                    const uint32_t bk = b_lo == b_hi ? b_lo : block_of(pg + j);
                    atomicAdd(&s_hist[bk][(wd[j >> 2] >> ((j & 3u) * 8u)) & 0xffu], 1u);
                }
            }
            uint32_t mm = mats[u];
            while (mm) {                                 // chosen records, four loads in flight
                uint32_t q[4], e4[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) { q[k] = mm ? pg + (uint32_t)__builtin_ctz(mm) : 0xffffffffu; mm &= mm - 1u; }
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t qq = q[k] != 0xffffffffu ? q[k] : pg;
                    e4[k] = chosen[(uint32_t)s_mrank[qq >> 6] + (uint32_t)__popcll(mword[u] & ((1ull << (qq & 63u)) - 1ull))];
                }
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    if (q[k] == 0xffffffffu) continue;
                    const uint32_t e = e4[k];
                    const uint32_t bk = b_lo == b_hi ? b_lo : block_of(q[k]);
                    atomicAdd(&s_hist[bk][257u + length_code(entry_len(e) - kMinMatch)], 1u);
                    atomicAdd(&s_hist[bk][kLCodes + dist_code(entry_dist(e) - 1u)], 1u);
                }
            }
        }
    }
    __syncthreads();
    BlockInfo* bi = blocks + (size_t)chunk * kMaxBlocks;
    for (uint32_t b = 0; b < ci.n_blocks; b++) {
        for (uint32_t i = tid; i < kLCodes; i += kBlockifyThreads) bi[b].lfreq[i] = (uint16_t)(i == 256 ? 1u : s_hist[b][i]);
        if (tid < kDCodes) bi[b].dfreq[tid] = (uint16_t)s_hist[b][kLCodes + tid];
        if (tid == 0) {
            bi[b].start = s_start[b]; bi[b].end = b + 1 < ci.n_blocks ? s_start[b + 1] : L;
            bi[b].flush_pos = b + 1 < ci.n_blocks ? s_flush[b] : L;
            bi[b].first_sym = b * kSymsPerBlock;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// encode: bit-pack one chunk's zlib stream into LDS, then stream the first <= 65535 bytes out.
struct EncBlock { uint32_t type, hdr_pos, body_pos, body_bits, start, end, first_sym, eob_len, eob_code, sym_bits_before, data_byte; };

template <uint32_t kCapWords = kOutWords>
static __device__ __forceinline__ void lds_or_bits(uint32_t* out, uint64_t bitpos, uint64_t v, uint32_t n) {
    // OR n (<= 48) bits of v at absolute bit position; anything at or past word kOutWords is dropped
    if (n == 0) return;
    const uint32_t w = (uint32_t)(bitpos >> 5), o = (uint32_t)(bitpos & 31u);
    const uint64_t lo = v << o;
    const uint32_t hi = o ? (uint32_t)(v >> (64u - o)) : 0u;
    if (w < kCapWords && (uint32_t)lo) atomicOr(&out[w], (uint32_t)lo);
    if (w + 1 < kCapWords && (uint32_t)(lo >> 32)) atomicOr(&out[w + 1], (uint32_t)(lo >> 32));
    if (w + 2 < kCapWords && hi) atomicOr(&out[w + 2], hi);
}

// encode's private slots (one Huffman block, one pass): a wave packs its wpw words of positions at slot0 + wave * slot_bits first.  Twice a
// wave's share of the block's bits (plan knows them), at most nine bits a position, or what the staging buffer has when that is less (a full chunk: 7.98 -- a Huffman chunk is smaller than its bytes as a whole, a
// wave's segment need not be: then the chunk takes the two passes).  (Seven bits a position, the first form, sent most image-like chunks --
// literals at 7+ bits -- through the attempt AND the two passes: configs[3]-shaped files 22.2 -> 33.7 ms.)
template <uint32_t kWaves, uint32_t kCapWords>
static __device__ __forceinline__ uint32_t enc_slot_bits(uint32_t wpw, uint32_t body_pos, uint32_t body_bits) {
    const uint32_t slot0 = (body_pos + 63u) & ~31u, nine = wpw * 64u * 9u, twice = 2u * (body_bits / kWaves) + 1024u;
    const uint32_t want = nine < twice ? nine : twice;                      // (the buffer is zeroed as far as the slots reach: no wider than need be)
    const uint32_t have = kCapWords * 32u > slot0 + 64u ? (kCapWords * 32u - 64u - slot0) / kWaves : 0u;
    return (want < have ? want : have) & ~31u;
}

// An incompressible chunk: every block is stored, so the stream is the input with a 2-byte zlib
// header, a 5-byte header in front of each block and the Adler-32 behind.  No staging: one pass
// over the input for the checksum (v_dot4 sums four bytes per instruction; skipped when the cap cuts
// the checksum off), one pass that writes whole output dwords straight to HBM, each from one unaligned
// 4-byte read of the input except the handful of dwords that touch a header.
// (lane b of every wave holds block b's byte range: the caller read it alongside the chunk's other metadata, so that the
// copy is two HBM round trips -- metadata, data -- and not five dependent ones; one workgroup per CU has nothing to hide them behind)
template <uint32_t T>                                   // threads of the workgroup; 16 384 / T output vectors a thread
static __device__ __forceinline__ void encode_stored_chunk(const uint8_t* __restrict__ data, uint32_t L, uint32_t n_blocks,
                                                           uint32_t lane_start, uint32_t lane_end, uint32_t* __restrict__ gout,
                                                           uint32_t* __restrict__ out_len_slot) {
    __shared__ uint32_t s_hb[kMaxBlocks + 1], s_st[kMaxBlocks + 1];   // stream offset of each block header; its first input byte
    __shared__ uint32_t s_a[T / 64], s_adler_be;
    __shared__ unsigned long long s_b[T / 64];
    constexpr uint32_t kRounds = 1024u / T;
    const uint32_t tid = threadIdx.x;
    static_assert(kMaxBlocks == 5, "block ranges are passed in lanes 0..4");
    const uint32_t bs[kMaxBlocks] = {(uint32_t)__builtin_amdgcn_readlane((int)lane_start, 0), (uint32_t)__builtin_amdgcn_readlane((int)lane_start, 1),
                                     (uint32_t)__builtin_amdgcn_readlane((int)lane_start, 2), (uint32_t)__builtin_amdgcn_readlane((int)lane_start, 3),
                                     (uint32_t)__builtin_amdgcn_readlane((int)lane_start, 4)};
    const uint32_t be[kMaxBlocks] = {(uint32_t)__builtin_amdgcn_readlane((int)lane_end, 0), (uint32_t)__builtin_amdgcn_readlane((int)lane_end, 1),
                                     (uint32_t)__builtin_amdgcn_readlane((int)lane_end, 2), (uint32_t)__builtin_amdgcn_readlane((int)lane_end, 3),
                                     (uint32_t)__builtin_amdgcn_readlane((int)lane_end, 4)};
    if (tid == 0) {
        uint32_t hb = 2;
#pragma unroll
        for (uint32_t b = 0; b < kMaxBlocks; b++) {
            if (b < n_blocks) { s_hb[b] = hb; s_st[b] = bs[b]; hb += 5u + (be[b] - bs[b]); }
            else { s_hb[b] = hb; s_st[b] = L; }
        }
        s_hb[kMaxBlocks] = hb; s_st[kMaxBlocks] = L;                   // s_hb[n_blocks] = where the Adler-32 goes
    }
    const uint32_t* d32 = reinterpret_cast<const uint32_t*>(data);
    __syncthreads();
    // Adler-32: a = 1 + sum d_i, b = L + sum (L - i) d_i  (mod 65521); 16 dwords per thread, all in flight.
    // Not for a chunk whose stream reaches the 65 535-byte cap before the trailer starts (every full incompressible
    // chunk: 65 535 + 5 headers): the reference cuts the checksum off, so the input is read once, by the copy below.
    if (s_hb[n_blocks] < kChunk) {                            // workgroup-uniform
        uint32_t a_sum = 0; unsigned long long b_sum = 0;
        for (uint32_t r = 0; r < kRounds; r++) {
            uint32_t w[16];
#pragma unroll
            for (uint32_t u = 0; u < 16; u++) {
                const uint32_t i = (tid + (u + 16u * r) * T) * 4u;
                w[u] = i < L ? d32[i >> 2] : 0u;              // slot readable to L rounded up to 16
            }
#pragma unroll
            for (uint32_t u = 0; u < 16; u++) {
                const uint32_t i = (tid + (u + 16u * r) * T) * 4u;
                uint32_t x = w[u];
                if (i < L && i + 4u > L) x &= (1u << (8u * (L - i))) - 1u;
                const uint32_t sum = __builtin_amdgcn_udot4(x, 0x01010101u, 0u, false), ramp = __builtin_amdgcn_udot4(x, 0x03020100u, 0u, false);
                a_sum += sum;
                b_sum += (i < L ? (L - i) * sum - ramp : 0u);    // < 2^27 per dword
            }
        }
        for (uint32_t d = 32; d >= 1; d >>= 1) { a_sum += __shfl_down(a_sum, d); b_sum += __shfl_down(b_sum, d); }
        if (lane_id() == 0) { s_a[tid >> 6] = a_sum; s_b[tid >> 6] = b_sum; }
        __syncthreads();
        if (tid == 0) {
            unsigned long long a = 1, b = L;
            for (uint32_t i = 0; i < T / 64; i++) { a += s_a[i]; b += s_b[i]; }
            s_adler_be = __builtin_bswap32((uint32_t)((b % 65521ull) << 16) | (uint32_t)(a % 65521ull));
        }
        __syncthreads();
    }
    if (tid == 0) { const uint32_t total = s_hb[n_blocks] + 4u; *out_len_slot = total < kChunk ? total : kChunk; }
    const uint32_t total = s_hb[n_blocks] + 4u, n_out = total < kChunk ? total : kChunk;
    const uint32_t h1 = s_hb[1], h2 = s_hb[2], h3 = s_hb[3], h4 = s_hb[4], h_end = s_hb[n_blocks], adler_be = h_end < kChunk ? s_adler_be : 0u;
    auto block_at = [&](uint32_t x) { return (uint32_t)(x >= h1) + (uint32_t)(x >= h2) + (uint32_t)(x >= h3) + (uint32_t)(x >= h4); };   // unused slots hold h_end
    auto byte_at = [&](uint32_t x) -> uint32_t {
        if (x < 2u) return x ? 0x9cu : 0x78u;
        if (x >= h_end) return x - h_end < 4u ? (adler_be >> (8u * (x - h_end))) & 0xffu : 0u;
        uint32_t b = block_at(x);
        if (b >= n_blocks) b = n_blocks - 1u;
        const uint32_t off = x - s_hb[b], len = s_hb[b + 1] - s_hb[b] - 5u;
        if (off == 0u) return b + 1u == n_blocks ? 1u : 0u;
        if (off < 5u) { const uint32_t f = (len & 0xffffu) | ((~len & 0xffffu) << 16); return (f >> (8u * (off - 1u))) & 0xffu; }
        return data[s_st[b] + off - 5u];
    };
    // Four 16-byte output vectors per thread, each from one 16-byte and one 4-byte read of the input (any dword alignment)
    // unless it touches a header or the trailer; all reads in flight before the first is used.  (Dword by dword -- two
    // reads and a store per four bytes -- the copy ran at 3.5 TB/s of traffic; the memory pipeline counts instructions.)
    const uint32_t n_v = (n_out + 15u) >> 4;
  for (uint32_t r = 0; r < kRounds; r++) {
    if ((4u * r) * T >= n_v) break;
    uint4 qa[4]; uint32_t qb[4], sh[4]; bool fast[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; u++) {
        const uint32_t o = (tid + (u + 4u * r) * T) * 16u;
        uint32_t b = block_at(o);
        if (b >= n_blocks) b = n_blocks - 1u;
        const uint32_t lo = s_hb[b] + 5u, hi = s_hb[b + 1];
        fast[u] = o + 16u <= n_out && o >= lo && o + 16u <= hi && hi <= h_end;
        const uint32_t p = fast[u] ? s_st[b] + (o - lo) : 0u;
        sh[u] = p & 3u;
        const uint32_t* src = d32 + (p >> 2);
        qa[u] = fast[u] ? make_uint4(src[0], src[1], src[2], src[3]) : make_uint4(0, 0, 0, 0);
        qb[u] = fast[u] && sh[u] ? src[4] : 0u;          // (the slot is readable to L rounded up to 16: only touched when needed)
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; u++) {
        const uint32_t ov = tid + (u + 4u * r) * T, o = ov * 16u;
        if (ov >= n_v) continue;
        uint4 v;
        if (fast[u]) {
            v = make_uint4(__builtin_amdgcn_alignbyte(qa[u].y, qa[u].x, sh[u]), __builtin_amdgcn_alignbyte(qa[u].z, qa[u].y, sh[u]),
                           __builtin_amdgcn_alignbyte(qa[u].w, qa[u].z, sh[u]), __builtin_amdgcn_alignbyte(qb[u], qa[u].w, sh[u]));
        } else {
            uint32_t w[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t x = o + 4u * j;
                w[j] = byte_at(x) | byte_at(x + 1u) << 8 | byte_at(x + 2u) << 16 | byte_at(x + 3u) << 24;
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        reinterpret_cast<uint4*>(gout)[ov] = v;
    }
  }
}

// encode, first kernel: chunks whose blocks are all stored (incompressible data) are copied here, by small workgroups without
// the bit-packer's LDS -- eight to a CU, so that one's two HBM round trips hide behind the others' (inside encode_kernel, two
// 81 KB workgroups a CU, a stored chunk cost ~12 us whatever its length: 1.7 ms per 50 000).  Every other chunk goes onto the
// list encode_kernel works through.
constexpr uint32_t kEncStoredThreads = 256;

__global__ __launch_bounds__(kEncStoredThreads) void encode_stored_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                                          const uint32_t* __restrict__ in_len, const ChunkInfo* __restrict__ info,
                                                                          const BlockInfo* __restrict__ blocks, const BlockOut* __restrict__ plans,
                                                                          uint8_t* __restrict__ out, uint64_t out_stride, uint32_t* __restrict__ out_len,
                                                                          uint32_t* __restrict__ huff_list, uint32_t* __restrict__ small_list, uint32_t* __restrict__ tickets) {
    const uint32_t chunk = blockIdx.x;
    const uint32_t L = in_len[chunk], n_blocks = info[chunk].n_blocks;
    const BlockInfo* bi = blocks + (size_t)chunk * kMaxBlocks;
    const BlockOut* bo = plans + (size_t)chunk * kMaxBlocks;
    // lane b looks at block b (every wave: the copy wants the ranges in each wave's lanes); these loads do not wait for n_blocks
    const uint32_t lb = lane_id() < kMaxBlocks ? lane_id() : 0u;
    const uint32_t my_type = bo[lb].type, my_start = bi[lb].start, my_end = bi[lb].end;
    const bool all_stored = n_blocks > 0 && __builtin_amdgcn_ballot_w64(lane_id() < n_blocks && my_type != kStored) == 0;   // workgroup-uniform
    if (!all_stored) {
        if (threadIdx.x == 0) {
            if (n_blocks == 1u && L <= kSmallEncBytes) small_list[atomicAdd(&tickets[kTicketSmallCount], 1u)] = chunk;
            else huff_list[atomicAdd(&tickets[kTicketHuffCount], 1u)] = chunk;
        }
        return;
    }
    encode_stored_chunk<kEncStoredThreads>(in + in_off[chunk], L, n_blocks, my_start, my_end,
                                           reinterpret_cast<uint32_t*>(out + (size_t)chunk * out_stride), out_len + chunk);
}

// (ZWZ_ENC_EXP & 16, experiment builds only -- tools/encode_times.sh: wave 0 of every workgroup sums the cycles per phase in registers,
// adds them, >> 8, to g_enc_times[] at its end; launch_deflate prints them when ZWZ_ENC_TIMES is set)
#ifndef ZWZ_ENC_EXP
#define ZWZ_ENC_EXP 0
#endif
#if ZWZ_ENC_EXP & 16
__device__ unsigned long long g_enc_times[8];
#define ZWZ_ESTAMP(ph) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); eacc_[(ph)] += (uint32_t)(now_ - estamp_); estamp_ = now_; } while (0)
#else
#define ZWZ_ESTAMP(ph) do { } while (0)
#endif
// (Round 5, measured and dropped: the kernel as a template -- a one-pass launch for the chunks of one Huffman block, a two-pass launch for the others and
// for the overflows, handed over through a second list -- so that each has half the code: scratch 76 -> 20 bytes a lane in the one-pass kernel, text
// 5.66 -> 6.12 ms.  What the scratch holds is read at the head of phases, not in the trips; what the split changed was the allocation inside them.)
// kThreads / kOutW: 1 024 threads and a 64 KB staging buffer for every chunk (two workgroups a CU), or -- round 5 -- 256 threads and 16 KB for the
// chunks of one block and at most kSmallEncBytes bytes (encode_stored_kernel sorts them onto their own list): six workgroups a CU.  A 7 KB file is
// little work behind a chain of a dozen dependent round trips and barriers, and a CU with two of them in flight mostly waits.
template <uint32_t kThreads, uint32_t kOutW, uint32_t kMinWaves>
__global__ __launch_bounds__(kThreads, kMinWaves) void encode_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                                const uint32_t* __restrict__ in_len, const uint2* __restrict__ entries,
                                                                const uint64_t* __restrict__ sym, const uint64_t* __restrict__ mst,
                                                                const ChunkInfo* __restrict__ info,
                                                                const BlockInfo* __restrict__ blocks, const BlockOut* __restrict__ plans,
                                                                uint8_t* __restrict__ out, uint64_t out_stride, uint32_t* __restrict__ out_len,
                                                                const uint16_t* __restrict__ links,
                                                                const uint32_t* __restrict__ huff_list, uint32_t* __restrict__ tickets, uint32_t t_count, uint32_t t_next) {
    // Persistent (two workgroups a CU by LDS): the chunks encode_stored_kernel left on the list, handed out by a ticket counter.
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* s_out = reinterpret_cast<uint32_t*>(smem);                          // kOutW
    uint32_t* s_queue = reinterpret_cast<uint32_t*>(smem + kOutW * 4);        // kEncQueue entries per wave (8 KiB): symbol starts waiting for a full trip
    uint16_t* s_lcode = reinterpret_cast<uint16_t*>(s_queue + (kThreads / 64) * kEncQueue);   // kMaxBlocks * 288
    uint16_t* s_dcode = s_lcode + kMaxBlocks * 288;                               // kMaxBlocks * 32
    uint8_t* s_llen = reinterpret_cast<uint8_t*>(s_dcode + kMaxBlocks * 32);      // kMaxBlocks * 288
    uint8_t* s_dlen = s_llen + kMaxBlocks * 288;                                  // kMaxBlocks * 32
    __shared__ EncBlock s_blk[kMaxBlocks];
    __shared__ uint32_t s_wsum[kThreads / 64], s_msum[kThreads / 64 + 1];
    __shared__ uint32_t s_total_bytes, s_adler_a[kThreads / 64];
    __shared__ unsigned long long s_adler_b[kThreads / 64];

    __shared__ uint32_t s_item;
    const uint32_t tid = threadIdx.x;
    const uint32_t n_items = tickets[t_count];                // final: encode_stored_kernel has finished
#if ZWZ_ENC_EXP & 16
    uint64_t estamp_ = __builtin_amdgcn_s_memtime();
    uint32_t eacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (;;) {
    if (tid == 0) s_item = atomicAdd(&tickets[t_next], 1u);
    __syncthreads();
    const uint32_t item = s_item;
    if (item >= n_items) break;
    const uint32_t chunk = huff_list[item];
    const uint32_t L = in_len[chunk];
    const uint32_t nwords = (L + 63) >> 6;
    const ChunkInfo ci = info[chunk];
    const BlockInfo* bi = blocks + (size_t)chunk * kMaxBlocks;
    const BlockOut* bo = plans + (size_t)chunk * kMaxBlocks;
    const uint8_t* data = in + in_off[chunk];
    const uint32_t* chosen = chosen_of(links, chunk);
    const uint64_t* gsym = sym + (size_t)chunk * kMaskWords;
    const uint64_t* gmst = mst + (size_t)chunk * kMaskWords;
    // match ranks (cheaper to recompute than to round-trip through HBM): the chunk's mask words are dealt out to the waves in equal
    // contiguous shares -- lane l of wave w holds word w * wpw + l of both masks, the wave's segment below -- so that a short
    // chunk keeps all sixteen waves busy (with fixed 4096-position segments a 7 KB chunk was two waves' work and cost what a
    // 64 KB one did: 4.1 ms per 30 000 image-like files)
    const uint32_t wpw = (nwords + kThreads / 64u - 1u) / (kThreads / 64u);       // words per wave: 64 for a full chunk
    const uint32_t my_word = (tid >> 6) * wpw + lane_id();
    const bool has_word = lane_id() < wpw && my_word < nwords;
    const uint64_t mst_l = has_word ? gmst[my_word] : 0ull;
    const uint64_t sym_l = has_word ? gsym[my_word] : 0ull;
    uint32_t mprefix;                                         // matches before this thread's word = its first index into `chosen`
    {
        const uint32_t mcnt = (uint32_t)__popcll(mst_l), mincl = wave_scan_incl(mcnt);
        if (lane_id() == 63) s_msum[tid >> 6] = mincl;
        __syncthreads();
        uint32_t mbase = 0;
        for (uint32_t i = 0; i < (tid >> 6); i++) mbase += s_msum[i];
        mprefix = mbase + mincl - mcnt;
    }
    for (uint32_t b = 0; b < ci.n_blocks; b++) {
        if (bo[b].type == kStored) continue;   // stored blocks carry no codes (plan may not have built any)
        for (uint32_t i = tid; i < kLCodes; i += kThreads) { s_lcode[b * 288 + i] = bo[b].lcode[i]; s_llen[b * 288 + i] = bo[b].llen[i]; }
        if (tid < kDCodes) { s_dcode[b * 32 + tid] = bo[b].dcode[tid]; s_dlen[b * 32 + tid] = bo[b].dlen[tid]; }
    }
    if (tid == 0) {
        // stream layout: 2 header bytes, blocks bit-contiguous (stored blocks and the final block pad to a byte)
        uint64_t bit = 16; uint32_t before = 0;
        for (uint32_t b = 0; b < ci.n_blocks; b++) {
            EncBlock e;
            e.type = bo[b].type; e.hdr_pos = (uint32_t)bit; e.start = bi[b].start; e.end = bi[b].end;
            e.first_sym = bi[b].first_sym; e.eob_len = bo[b].eob_len; e.eob_code = bo[b].eob_code;
            e.body_bits = bo[b].body_bits; e.sym_bits_before = before; e.data_byte = 0;
            bit += bo[b].hdr_bits;
            if (e.type == kStored) {
                bit = (bit + 7) & ~7ull;
                e.body_pos = (uint32_t)bit;          // LEN/NLEN start
                e.data_byte = (uint32_t)(bit >> 3) + 4;
                bit += 32 + 8ull * (e.end - e.start);
            } else {
                e.body_pos = (uint32_t)bit;
                bit += e.body_bits;
                before += e.body_bits - e.eob_len;
            }
            if (b + 1 == ci.n_blocks) bit = (bit + 7) & ~7ull;
            s_blk[b] = e;
        }
        s_total_bytes = (uint32_t)(bit >> 3) + 4;    // + Adler-32
    }
    __syncthreads();
    ZWZ_ESTAMP(0);                                            // ticket, masks, match ranks, codes, the stream's layout (thread 0)
    {   // the staging buffer, zeroed as far as this chunk's stream reaches (a whole 64 KiB per chunk was most of what a
        // 4-byte chunk cost: the 10 000 tail chunks of BASELINE configs[1] took 0.2 ms)
        // (a chunk of one Huffman block is packed at private slots first -- below -- whose reach is enc_slot_bits')
        const bool slots = ci.n_blocks == 1u && s_blk[0].type != kStored;
        const uint32_t reach = slots ? ((s_blk[0].body_pos + 63u) >> 5) + (kThreads / 64u) * (enc_slot_bits<kThreads / 64u, kOutW>(wpw, s_blk[0].body_pos, s_blk[0].body_bits) >> 5) + 2u : 0u;
        const uint32_t nz = min(kOutW, max((s_total_bytes + 3u) / 4u + 2u, reach));
        for (uint32_t i = tid; i < nz; i += kThreads) s_out[i] = 0;
    }
    __syncthreads();

    // A chunk that is ONE Huffman block (text: 13.6 k symbols a chunk, under the 16 383 of a block) is packed in a single pass over its symbols
    // (round 5; see the symbols' block below): the stream's fixed parts then go in behind the symbols, whose waves use the buffer as scratch first.
    const bool one_pass = ci.n_blocks == 1u && s_blk[0].type != kStored && (ZWZ_ENC_EXP & 1) == 0;       // workgroup-uniform
    auto write_headers = [&](bool with_eob) {
        if (tid == 0) lds_or_bits<kOutW>(s_out, 0, 0x9c78u, 16);
        for (uint32_t b = 0; b < ci.n_blocks; b++) {
            const EncBlock e = s_blk[b];
            const uint32_t hw = (bo[b].hdr_bits + 31) >> 5;
            if (tid < hw) {
                uint32_t nb = bo[b].hdr_bits - (tid << 5); if (nb > 32) nb = 32;
                uint32_t v = bo[b].hdr[tid]; if (nb < 32) v &= (1u << nb) - 1u;
                lds_or_bits<kOutW>(s_out, (uint64_t)e.hdr_pos + (tid << 5), v, nb);
            }
            if (tid == 0) {
                if (e.type == kStored) {
                    const uint32_t len = e.end - e.start;
                    lds_or_bits<kOutW>(s_out, e.body_pos, (uint64_t)(len & 0xffffu) | ((uint64_t)(~len & 0xffffu) << 16), 32);
                } else if (with_eob) {
                    lds_or_bits<kOutW>(s_out, (uint64_t)e.body_pos + e.body_bits - e.eob_len, e.eob_code, e.eob_len);
                }
            }
        }
    };
    write_headers(!one_pass);                                 // (one pass: the headers lie in front of the first slot; the end-of-block code joins on the way out)
    bool in_slots = false;                                    // workgroup-uniform: the symbols' bits are still in the waves' slots when the stream goes out
    uint32_t pre_l = 0;                                       // .. and lane i <= 16 holds the bits in front of wave i's

    ZWZ_ESTAMP(1);                                            // staging zeroed, block headers
    uint32_t a_sum = 0; unsigned long long b_sum = 0;   // Adler partials over this thread's bytes
    uint8_t* s_out8 = reinterpret_cast<uint8_t*>(s_out);
    {
        // Adler-32 partials: a = 1 + sum d_i, b = L + sum (L - i) d_i; 16 dwords per thread, v_dot4 (the slot is readable to
        // L rounded up to 16, what lies past L is masked)
        const uint32_t* d32 = reinterpret_cast<const uint32_t*>(data);
#pragma unroll 1
        for (uint32_t u0 = 0; u0 < 16u; u0 += 4u) {                       // four loads in flight (sixteen pushed this kernel's 64 registers into scratch)
            if (u0 * kThreads * 4u >= L) break;                     // (workgroup-uniform: a short chunk has nothing there)
            uint32_t w[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) { const uint32_t i = (tid + (u0 + u) * kThreads) * 4u; w[u] = i < L ? d32[i >> 2] : 0u; }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t i = (tid + (u0 + u) * kThreads) * 4u;
                uint32_t x = w[u];
                if (i < L && i + 4u > L) x &= (1u << (8u * (L - i))) - 1u;
                const uint32_t sum = __builtin_amdgcn_udot4(x, 0x01010101u, 0u, false), ramp = __builtin_amdgcn_udot4(x, 0x03020100u, 0u, false);
                a_sum += sum;
                b_sum += (i < L ? (L - i) * sum - ramp : 0u);    // < 2^27 per dword
            }
        }
        // stored blocks among Huffman ones (rare): their bytes go straight into the stream
        for (uint32_t b = 0; b < ci.n_blocks; b++) {
            if (s_blk[b].type != kStored) continue;
            const uint32_t st0 = s_blk[b].start, len = s_blk[b].end - st0, ob0 = s_blk[b].data_byte;
            for (uint32_t i = tid; i < len; i += kThreads) if (ob0 + i < kOutW * 4) s_out8[ob0 + i] = data[st0 + i];
        }
    }
    {
        // Symbols.  Wave w owns the contiguous positions [64 wpw w, 64 wpw (w + 1)): a first pass adds up
        // its code lengths, one barrier turns the 16 wave totals into start offsets, and the second
        // pass packs bits with wave-local scans only (a barrier per 1024 positions kept the single
        // resident workgroup of a CU waiting: 30 ms for 50k text chunks).
        // Both passes work on symbol STARTS, 64 to a trip: only three positions in eight start a symbol on text, and walking the
        // positions cost ~115 vector instructions per 64 of them in each pass on a kernel the vector unit bounds.  The segment's
        // 64 words of both masks (and the match count in front of each) sit one per lane and are handed out by v_readlane; a
        // word's symbol starts are appended to the wave's queue in LDS -- position, index of the chosen record, match flag --
        // and whenever 64 are waiting they are taken as a trip, whose byte and chosen record are fetched while the trip before
        // it is being worked on.  The records come from lz_parse's compact array (read from the position-indexed one, every
        // 64-byte line of its 512 KB held a match start and both passes fetched all of it: ~50 GB a pass on text).
        const uint32_t wave = tid >> 6, lane = lane_id();
        const uint32_t seg = min(wave * wpw * 64u, L), seg_end = min(seg + wpw * 64u, L);
        // blocks are contiguous position ranges: a position's block is the number of block starts at
        // or before it (holds for covered positions too: a block ends where its last symbol ends)
        const uint32_t b1 = ci.n_blocks > 1 ? s_blk[1].start : 0xffffffffu, b2 = ci.n_blocks > 2 ? s_blk[2].start : 0xffffffffu;
        const uint32_t b3 = ci.n_blocks > 3 ? s_blk[3].start : 0xffffffffu, b4 = ci.n_blocks > 4 ? s_blk[4].start : 0xffffffffu;
        auto block_of = [&](uint32_t p) { return (uint32_t)(p >= b1) + (uint32_t)(p >= b2) + (uint32_t)(p >= b3) + (uint32_t)(p >= b4); };
        const uint32_t trips = (seg_end > seg ? seg_end - seg + 63u : 0u) >> 6;
        uint16_t* queue = reinterpret_cast<uint16_t*>(s_queue + wave * kEncQueue);
        constexpr uint32_t kCap = 2u * kEncQueue;                           // 16-bit entries: position inside the segment (12 bits) | match flag << 15
        // The queue is filled FOUR LANES A WORD: lane l takes quarter l & 3 of word next + (l >> 2) -- sixteen bits of both masks, fetched
        // from the lanes that hold the words -- and writes out its symbol starts one a turn, behind those of the lanes in front of it (a scan
        // of the quarters' counts).  On text a quarter holds three starts: a round of sixteen words is ~7 turns of a dozen instructions for ~200
        // starts; where the words are nearly all starts (few matches: image-like data) they are not queued at all.  (Rounds 3 - 5 took the words one at a time -- five readlanes, ranks among the word's starts, the entry: 35 instructions a
        // word, 175 a trip of 64 starts on text, more than the trip's own work.)  As many whole words go in as the queue has room for: at
        // least three (fewer than 64 entries wait when a round starts).
        // (Measured and dropped: ONE call site of work() an instantiation and nothing unrolled -- the kernel's code falls from 52 KB to 26 KB, on the idea
        // that the once-a-chunk pieces behind barriers run out of a cold instruction cache: text 5.68 -> 6.39 ms.)
        auto for_each_trip = [&](auto&& work) {                            // work(position, byte, record, live) for every trip of <= 64 symbol starts, in stream order
            uint32_t head = 0, tail = 0, next = 0;                         // wave-uniform: queue[head, tail) is waiting; words [0, next) are in
            uint32_t m_run = (uint32_t)__builtin_amdgcn_readlane((int)mprefix, 0);   // matches in front of the next entry = its index into `chosen`
            uint32_t p_pos = 0, p_byte = 0, p_rec = 0; uint64_t p_live = 0;  // the trip whose operands are in flight
            auto hand_over = [&](uint32_t pos, uint32_t byte, uint32_t rec, uint64_t live_mask) {
                if (p_live) work(p_pos, p_byte, p_rec, (bool)((p_live >> lane) & 1ull));
                p_pos = pos; p_byte = byte; p_rec = rec; p_live = live_mask;
            };
            auto take = [&](uint32_t n) {                                  // fetch for n waiting entries, then work on the trip taken before
                const uint32_t ent = queue[(head + lane) & (kCap - 1u)];            // (lanes >= n: stale entries, masked by `live`)
                const bool live = lane < n;
                const uint32_t pos = seg + (ent & 0xfffu);
                const uint64_t mm = __builtin_amdgcn_ballot_w64(live && (ent >> 15));
                const bool is_m = (mm >> lane) & 1ull;
                const uint32_t byte = live ? (uint32_t)data[pos] : 0u;
                const uint32_t rec = is_m ? chosen[m_run + rank_in(mm)] : 0u;       // dense: 4 bytes per match
                m_run += (uint32_t)__popcll(mm);
                head += n;
                hand_over(pos, byte, rec, n >= 64u ? ~0ull : (1ull << n) - 1ull);
            };
            auto word_of = [&](uint64_t v, uint32_t it) -> uint64_t {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, it);          // the builtin returns int:
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), it);  // widen only after the cast
                return (uint64_t)lo | ((uint64_t)hi << 32);
            };
            auto direct = [&](uint32_t w) {                                // a word that is nearly all symbol starts is a trip as it stands: lane = position
                const uint64_t sw = word_of(sym_l, w), mw = word_of(mst_l, w);
                const bool live = (sw >> lane) & 1ull, is_m = (mw >> lane) & 1ull;
                const uint32_t pos = seg + (w << 6) + lane;
                const uint32_t byte = live ? (uint32_t)data[pos] : 0u;
                const uint32_t rec = is_m ? chosen[m_run + rank_in(mw)] : 0u;
                m_run += (uint32_t)__popcll(mw);
                if (sw) hand_over(pos, byte, rec, sw);
            };
            while (next < trips) {
                const uint32_t src = next + (lane >> 2), qsh = (lane & 1u) << 4;
                const int from = (int)((src & 63u) << 2);
                const uint32_t s_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)sym_l), s_hi = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)(sym_l >> 32));
                const uint32_t m_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)mst_l), m_hi = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)(mst_l >> 32));
                uint32_t bits = src < trips ? (((lane & 2u) ? s_hi : s_lo) >> qsh) & 0xffffu : 0u;
                const uint32_t mbits = (((lane & 2u) ? m_hi : m_lo) >> qsh) & 0xffffu;
                const uint32_t c = (uint32_t)__popc(bits), incl = wave_scan_incl(c);
                if ((uint32_t)__builtin_amdgcn_readlane((int)incl, 15) >= 192u) {       // the next four words hold >= 48 starts each (image-like data: literals)
                    if (tail != head) take(tail - head);                   // (what waits goes first)
                    for (uint32_t k = 0; k < 4u && next < trips; k++, next++) direct(next);
                    continue;
                }
                const uint32_t room = kCap - (tail - head);                // > 192
                const uint64_t in = __builtin_amdgcn_ballot_w64(incl <= room);       // the counts ascend: lanes 0 .. nl - 1
                const uint32_t nl = (~in ? (uint32_t)__builtin_ctzll(~in) : 64u) & ~3u;          // whole words only (>= 4: a word holds <= 64)
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, nl - 1u);
                if (lane < nl) {
                    uint32_t at = tail + incl - c;
                    const uint32_t base = (src << 6) | ((lane & 3u) << 4);
                    while (bits) {
                        const uint32_t bpos = (uint32_t)__builtin_ctz(bits);
                        queue[at & (kCap - 1u)] = (uint16_t)(base | bpos | ((mbits >> bpos) & 1u) << 15);
                        at++; bits &= bits - 1u;
                    }
                }
                tail += total; next += nl >> 2;
                while (tail - head >= 64u) take(64u);
            }
            if (tail != head) take(tail - head);
            if (p_live) work(p_pos, p_byte, p_rec, (bool)((p_live >> lane) & 1ull));
        };

        // the waves' Adler partials go to LDS here: whoever writes the stream's last words adds them up (no barrier of its own for the checksum)
        for (uint32_t d = 32; d >= 1; d >>= 1) { a_sum += __shfl_down(a_sum, d); b_sum += __shfl_down(b_sum, d); }
        if (lane == 0) { s_adler_a[wave] = a_sum; s_adler_b[wave] = b_sum; }
        ZWZ_ESTAMP(2);                                        // Adler partials, stored blocks' bytes
        // One Huffman block: ONE pass.  A wave does not know where its segment's bits start before every wave in front of it has added
        // up its code lengths -- rounds 1-4 made a first pass over the symbols for that sum alone (28 % of the kernel: every byte and
        // every chosen record fetched twice).  Here a wave packs its segment at a PRIVATE place first -- a slot of up to nine bits a position
        // in the same staging buffer (enc_slot_bits; a segment of text needs three) -- then, the wave totals known, reads its slot into registers, clears
        // it, and ORs the words back shifted to where they belong.  A wave whose segment outgrows its slot sends the chunk through the two
        // passes below (never on the corpora).
        bool two_pass = !one_pass;
        if (one_pass) {
            const uint32_t slot_bits = enc_slot_bits<kThreads / 64u, kOutW>(wpw, s_blk[0].body_pos, s_blk[0].body_bits), slot0 = (s_blk[0].body_pos + 63u) & ~31u;
            const uint32_t q = slot0 + wave * slot_bits;                                       // a multiple of 32
            uint32_t running = 0;
            for_each_trip([&](uint32_t ent, uint32_t byte, uint32_t rec, bool live) {
                uint64_t v = 0; uint32_t nb = 0;
                if (live) symbol_bits(s_lcode, s_llen, s_dcode, s_dlen, rec, byte, v, nb);
                const uint32_t incl = wave_scan_incl(nb);
                if (nb) lds_or_bits<kOutW>(s_out, (uint64_t)q + (running + incl - nb), v, nb);
                running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            });
            if (lane == 0) s_wsum[wave] = running;
            ZWZ_ESTAMP(3);
            __syncthreads();
            ZWZ_ESTAMP(4);
            // lane i < 16: wave i's bits; the scan gives every wave all sixteen starts
            const uint32_t m_l = lane < kThreads / 64u ? s_wsum[lane] : 0u;
            pre_l = wave_scan_incl(m_l) - m_l;                                                 // lane 16: all the symbols' bits
            const bool fits = __builtin_amdgcn_ballot_w64(m_l > slot_bits) == 0 && slot0 + (kThreads / 64u) * slot_bits + 64u <= kOutW * 32u;   // (.. and the slots themselves fit the buffer)
            if (fits) {
                // The slots stay where they are: the stream is put together on its way OUT (below) -- a word of it is the header's bits (in front of
                // the first slot, written above), at most a few slots' bits funnel-shifted to where the wave totals say they belong, the end-of-block
                // code and the checksum.  No barrier from here to the end of the chunk.  (Two forms that MOVED the slots inside the buffer first -- a
                // wave its own slot, the waves in turn; four slots a turn by the whole workgroup, read + cleared, a barrier, ORed back -- cost sixteen
                // resp. nine barriers and 14 k LDS atomics a chunk: 7.32 and 7.12 ms on text against the two passes' 8.33.  And between barriers
                // a workgroup is sixteen waves each running a short serial piece at an eighth of a SIMD: the prefix over the wave totals by
                // seventeen lanes and the checksum's last step by one thread -- two barriers -- were 40 k cycles a chunk.)
                if (lane <= kThreads / 64u) s_msum[lane] = pre_l;                         // every wave writes the same seventeen words and reads back its own: no barrier
                in_slots = true;
                        } else {
                two_pass = true;                                                               // workgroup-uniform: s_wsum is everybody's
                __syncthreads();
                for (uint32_t i = tid; i < kOutW; i += kThreads) s_out[i] = 0;
                __syncthreads();
                write_headers(true);
            }
        }
      if (two_pass) {
        uint32_t mine = 0;
        for_each_trip([&](uint32_t ent, uint32_t byte, uint32_t rec, bool live) {
            if (live) {
                const uint32_t blk = block_of(ent & 0xffffu);
                if (s_blk[blk].type != kStored) mine += symbol_nbits(s_llen + blk * 288, s_dlen + blk * 32, rec, byte);
            }
        });
        for (uint32_t d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
        __syncthreads();                                      // (s_wsum may still be read by the one-pass attempt's check)
        if (lane == 0) s_wsum[wave] = mine;
        ZWZ_ESTAMP(3);                                        // first pass: code lengths
        __syncthreads();
        ZWZ_ESTAMP(4);                                        // waiting for the other waves
        uint32_t running = 0;      // symbol bits emitted before this wave's segment (all Huffman blocks)
        for (uint32_t i = 0; i < wave; i++) running += s_wsum[i];
        for_each_trip([&](uint32_t ent, uint32_t byte, uint32_t rec, bool live) {
            uint64_t v = 0; uint32_t nb = 0, blk = 0;
            if (live) {
                blk = block_of(ent & 0xffffu);
                if (s_blk[blk].type != kStored)
                    symbol_bits(s_lcode + blk * 288, s_llen + blk * 288, s_dcode + blk * 32, s_dlen + blk * 32, rec, byte, v, nb);
            }
            const uint32_t incl = wave_scan_incl(nb);
            if (nb) {
                const EncBlock& eb = s_blk[blk];
                lds_or_bits<kOutW>(s_out, (uint64_t)eb.body_pos + (running + incl - nb - eb.sym_bits_before), v, nb);
            }
            running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        });
      }
    }

    ZWZ_ESTAMP(5);                                            // second pass: bits into the staging buffer
    // Adler-32 of the raw chunk: a = 1 + sum d_i, b = L + sum (L - i) d_i   (mod 65521)
    auto adler_be = [&]() -> uint32_t {
        unsigned long long a = 1, b = L;
        for (uint32_t i = 0; i < kThreads / 64; i++) { a += s_adler_a[i]; b += s_adler_b[i]; }
        return __builtin_bswap32((uint32_t)((b % 65521ull) << 16) | (uint32_t)(a % 65521ull));
    };
    const uint32_t n_out = s_total_bytes < kChunk ? s_total_bytes : kChunk;
    uint32_t* gout = reinterpret_cast<uint32_t*>(out + (size_t)chunk * out_stride);
    if (tid == 0) out_len[chunk] = n_out;
    if (!in_slots) {
        __syncthreads();
        if (tid == 0) lds_or_bits<kOutW>(s_out, 8ull * (s_total_bytes - 4), adler_be(), 32);
        __syncthreads();
    }
    ZWZ_ESTAMP(6);                                            // waiting for the other waves, the checksum
    if (!in_slots) {
        for (uint32_t i = tid; i < ((n_out + 3) >> 2); i += kThreads) gout[i] = s_out[i];
    } else {
        // word j of the stream = bits [32 j, 32 j + 32): header words as they lie; of the symbols, slot w holds bits [body + msum[w], + wsum[w]) from its own first bit on
        const uint32_t body = s_blk[0].body_pos, slot_bits = enc_slot_bits<kThreads / 64u, kOutW>(wpw, body, s_blk[0].body_bits), slot0 = (body + 63u) & ~31u;
        const uint32_t sym_end = body + (uint32_t)__builtin_amdgcn_readlane((int)pre_l, kThreads / 64u);
        const uint32_t eob_code = s_blk[0].eob_code, ad_pos = 8u * (s_total_bytes - 4u);
        uint32_t d_of[kThreads / 64u];                  // (scalar registers: the same for every lane)
#pragma unroll
        for (uint32_t i = 0; i < kThreads / 64u; i++) d_of[i] = body + (uint32_t)__builtin_amdgcn_readlane((int)pre_l, i);
        auto piece = [](uint32_t pos, uint32_t val, uint32_t bit) -> uint32_t {       // of val laid down at bit `pos`, what falls into [bit, bit + 32)
            const int d = (int)pos - (int)bit;
            return d >= 32 || d <= -32 ? 0u : d >= 0 ? val << d : val >> (-d);
        };
        for (uint32_t j = tid; j < ((n_out + 3) >> 2); j += kThreads) {
            const uint32_t bit = j << 5;
            uint32_t v = j < (slot0 >> 5) ? s_out[j] : 0u;
            v |= piece(sym_end, eob_code, bit);
            if (bit + 32u > ad_pos) v |= piece(ad_pos, adler_be(), bit);      // (the stream's last one or two words)
            if (bit + 32u > body && bit < sym_end) {
                uint32_t w = 0;                               // the slot bit `bit` falls into (the first one for the header's last word)
#pragma unroll
                for (uint32_t i = 1; i < kThreads / 64u; i++) w += d_of[i] <= bit ? 1u : 0u;
                for (; w < kThreads / 64u; w++) {
                    const uint32_t d = body + s_msum[w];
                    if (d >= bit + 32u) break;
                    const int sft = (int)bit - (int)d, len = (int)s_wsum[w];
                    const int lo = sft < 0 ? -sft : 0, hi = len - sft < 32 ? len - sft : 32;
                    if (hi > lo) {
                        const uint32_t at = (uint32_t)((int)(slot0 + w * slot_bits) + sft), aw = at >> 5;
                        const uint32_t x = __builtin_amdgcn_alignbit(s_out[aw + 1u], s_out[aw], at & 31u);
                        const uint32_t mask = (hi >= 32 ? 0xffffffffu : (1u << hi) - 1u) & ~((1u << lo) - 1u);
                        v |= x & mask;
                    }
                }
            }
            gout[j] = v;
        }
    }
    __syncthreads();                                          // everyone has read s_item, s_total_bytes and s_out
    ZWZ_ESTAMP(7);                                            // the stream out
  }
#if ZWZ_ENC_EXP & 16
    if (tid == ((ZWZ_ENC_EXP & 32) ? 960u : 0u)) for (uint32_t ph = 0; ph < 8; ph++) atomicAdd(&g_enc_times[ph], (unsigned long long)(eacc_[ph] >> 8));   // (& 32: as the last wave sees it)
#endif
}

// ------------------------------------------------------------------------------------------------
// inflate: one wave per chunk.  Lane 0 owns block headers and table construction; symbols are decoded by the
// whole wave from a window of the bit stream (see the Huffman branch below) and the wave moves the bytes.
// The payload comes through a 2 KiB LDS ring refilled 1 KiB at a time by the whole wave (16 bytes per lane,
// coalesced): a round consumes < 1 KiB, so topping the ring up to pos + 1 KiB before every round keeps the
// decoders off global memory entirely.
// (ZWZ_INF_EXP & 16, experiment builds only -- tools/inflate_times.sh: every wave sums the cycles it spent per phase of the Huffman
// branch in registers and adds them, >> 8, to g_inf_times[] at its end; launch_inflate prints them when ZWZ_INF_TIMES is set)
#ifndef ZWZ_INF_EXP
#define ZWZ_INF_EXP 0
#endif
#if ZWZ_INF_EXP & 16
__device__ unsigned long long g_inf_times[8];
#define ZWZ_ISTAMP(ph) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); iacc_[(ph)] += (uint32_t)(now_ - istamp_); istamp_ = now_; } while (0)
#else
#define ZWZ_ISTAMP(ph) do { } while (0)
#endif
#ifndef ZWZ_INF_WAVES
#define ZWZ_INF_WAVES 5
#endif
constexpr uint32_t kInflateWavesPerSimd = ZWZ_INF_WAVES;   // (round 5: six, with the owner map cut to 512 bytes and sharing the code lengths' place -- 6.6 KB a wave -- 22.42 -> 22.27 ms on text: noise, and 8 bytes of scratch)
#ifndef ZWZ_WIN_SLOTS
#define ZWZ_WIN_SLOTS 4
#endif
constexpr uint32_t kInfRing = 2048, kInfFill = 1024, kInfMirror = ((8u * ZWZ_WIN_SLOTS + 4u + 15u) / 16u) * 16u;   // ring, refill step, bytes of the ring's start repeated behind its end (a window fetch reads 36 consecutive bytes)
constexpr uint32_t kWinSlots = ZWZ_WIN_SLOTS;   // bit offsets decoded per lane per window: 256 bits
#ifndef ZWZ_WIN_PARTS
#define ZWZ_WIN_PARTS 2
#endif
constexpr uint32_t kWinParts = ZWZ_WIN_PARTS;   // windows a round may look at (offsets from the round's first bit stay below 1024: <= 4).  Three: text 25.0 -> 24.6 ms, 7 KB image-like files 13.9 -> 15.5 ms

constexpr uint32_t kOwnCap = 1024;         // batch bytes the per-byte owner map covers (a batch is <= 64 symbols: ~300 bytes on text)

// build_decode_table (inflate_core.h) by the whole wave, same tables and return value: per-length counts by LDS adds, the canonical
// bookkeeping (Kraft sum, offsets, first codes: fifteen steps) by every lane alike, each symbol's place in sorted[] as one
// ds_add_rtn a trip -- lanes in lane order, trips in program order: symbol order, the property zwz_ctx_create checks -- which is
// also its rank among the codes of its length, so every lane knows its symbols' codes and fills their fast-table entries itself.
// (On lane 0 alone a block's three tables were 0.5 M cycles: 7 % of inflate on text, 31 % on 7 KB files.)
static __device__ __forceinline__ int wave_build_decode_table(const uint8_t* lens, uint32_t n /* <= 320 */, uint16_t* fast /* 16-byte aligned */, uint32_t fast_bits,
                                                              uint16_t* count, uint16_t* sorted, uint32_t& max_len, uint16_t* walk0, uint32_t* scr /* 48 words */) {
    const uint32_t lane = lane_id();
    uint32_t* s_cnt = scr; uint32_t* s_pos = scr + 16; uint32_t* s_fc = scr + 32;
    uint32_t l[5];
#pragma unroll
    for (uint32_t r = 0; r < 5; r++) { const uint32_t i = lane + 64u * r; l[r] = i < n ? (uint32_t)lens[i] : 0u; }
    if (lane < 16u) s_cnt[lane] = 0;
    for (uint32_t x = lane; x < (1u << fast_bits) / 8u; x += 64u) reinterpret_cast<uint4*>(fast)[x] = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (uint32_t r = 0; r < 5; r++) if (l[r]) atomicAdd(&s_cnt[l[r]], 1u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int left = 1;
    uint32_t first = 0, index = 0, off = 0, code = 0, my_off = 0, my_fc = 0, my_cnt = 0;
    max_len = 0;
    for (uint32_t len = 1; len < 16u; len++) {
        const uint32_t c = s_cnt[len];
        if (c) max_len = len;
        left = (left << 1) - (int)c;
        if (left < 0) return -1;                                   // (wave-uniform)
        if (len <= fast_bits) { index += c; first += c; first <<= 1; }
        if (lane == len) { my_off = off; my_fc = code; my_cnt = c; }
        off += c; code = (code + c) << 1;
    }
    if (lane < 16u) { s_pos[lane] = my_off; s_fc[lane] = my_fc | my_off << 16; count[lane] = (uint16_t)my_cnt; }   // (lane 0: zeros, as build_decode_table's count[0])
    if (walk0 && lane == 0) { walk0[0] = (uint16_t)first; walk0[1] = (uint16_t)index; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (uint32_t r = 0; r < 5; r++) {
        if (l[r]) {
            const uint32_t i = lane + 64u * r, pos = atomicAdd(&s_pos[l[r]], 1u);
            sorted[pos] = (uint16_t)i;
            if (l[r] <= fast_bits) {
                const uint32_t w = s_fc[l[r]], cd = (w & 0xffffu) + (pos - (w >> 16));
                const uint16_t e = (uint16_t)((i << 4) | l[r]);
                for (uint32_t j = bit_reverse(cd, l[r]); j < (1u << fast_bits); j += 1u << l[r]) fast[j] = e;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return left > 0 ? 1 : 0;
}

// inflate_dyn_lengths (inflate_core.h) by the whole wave: the code-length symbols are a Huffman stream of their own (codes <= 7 bits,
// up to 7 extra bits), so they are decoded the way the block's symbols are -- every lane decodes the symbol that would start at its
// bit of a 64-bit window, a scalar hop marks the real starts -- and the run lengths (16: repeat the last length, 17 / 18: zeros)
// are expanded from a prefix sum.  Returns the bit position behind the last length, or 0xffffffff when anything is out of the
// ordinary (input that ends, a repeat with nothing before it, a run past the end): the caller then lets lane 0 run the
// sequential function from where the reader stands, which also gives zlib's exact verdict.  lens[] is only complete on success.
static __device__ __forceinline__ uint32_t wave_dyn_lengths(const uint8_t* ring, uint32_t bp, uint32_t total_bits, const uint16_t* cl_fast /* 7-bit */,
                                                            uint8_t* lens, uint32_t want /* nlen + ndist */) {
    const uint32_t lane = lane_id();
    uint32_t have = 0, carry = 0xffffffffu;                 // lengths written; the last length written (none yet)
    while (have < want) {
        const uint32_t a = bp + lane;
        const uint32_t byte = (a >> 3) & (kInfRing - 1u);
        const uint32_t* w = reinterpret_cast<const uint32_t*>(ring) + (byte >> 2);
        const uint32_t bits = __builtin_amdgcn_alignbyte(w[1], w[0], byte & 3u) >> (a & 7u);     // >= 25 valid bits
        const uint32_t e = cl_fast[bits & 127u], l = e & 15u, sym = e >> 4;
        const uint32_t xb = sym < 16u ? 0u : sym == 16u ? 2u : sym == 17u ? 3u : 7u, nb = l + xb;
        const uint32_t xv = (bits >> l) & ((1u << xb) - 1u);
        const bool bad = e == 0u || a + nb > total_bits;
        // the real symbols: the orbit of offset 0 under "offset + bits of the symbol decoded there"
        const uint32_t jr = lane + (nb ? nb : 1u);               // (strictly forward whatever the table holds: the hop below must end)
        uint64_t M = 0; uint32_t at = 0;
        while (at < 64u) { M |= 1ull << at; at = (uint32_t)__builtin_amdgcn_readlane((int)jr, (int)at); }
        const bool mine = (M >> lane) & 1ull;
        if (__ballot(mine && bad)) return 0xffffffffu;
        const uint32_t rep = !mine ? 0u : sym < 16u ? 1u : sym == 18u ? 11u + xv : 3u + xv;
        const uint32_t incl = wave_scan_incl(rep), before = have + incl - rep;
        // the header ends exactly at `want`: symbols from there on are the block's own
        const uint64_t over = __ballot(mine && before + rep > want), in_hdr = __ballot(mine && before < want);
        if (over & in_hdr) return 0xffffffffu;                   // a run that crosses the end
        const bool live = mine && before < want;
        // what a 16 repeats: the length written last before it -- by the nearest earlier symbol that is not a 16 (17 / 18: zero), else the carry
        const uint32_t def = live && sym != 16u ? lane + 1u : 0u;
        const uint32_t near = wave_scan_max_incl(def);           // inclusive: a 16 contributes nothing, so it sees the nearest definer before it
        const uint32_t dval = sym < 16u ? sym : 0u;
        const uint32_t pv = near ? (uint32_t)__shfl((int)dval, (int)(near - 1u)) : carry;
        const uint32_t val = sym == 16u ? pv : dval;
        if (__ballot(live && sym == 16u && pv == 0xffffffffu)) return 0xffffffffu;   // a repeat with nothing before it
        if (live) for (uint32_t k = 0; k < rep; k++) lens[before + k] = (uint8_t)val;
        const uint64_t lv = __ballot(live);
        const uint32_t last = 63u - (uint32_t)__builtin_clzll(lv);                // lv != 0: offset 0 is always live here
        carry = (uint32_t)__builtin_amdgcn_readlane((int)val, (int)last);
        have = (uint32_t)__builtin_amdgcn_readlane((int)(before + rep), (int)last);
        bp += (uint32_t)__builtin_amdgcn_readlane((int)jr, (int)last);
    }
    return bp;
}

struct InflateWaveMem {
    InflateTables t;
    uint8_t lens[320];
    uint32_t batch[kBatch], pos[kBatch];
    uint32_t sym[kBatch + 1];                 // a round's symbols in orbit order: bits << 13 | kind << 10 | offset from the round's first bit
    __attribute__((aligned(16))) uint8_t ownb[kOwnCap];   // per output byte of a batch: 1 + the batch symbol that writes it (batches of <= kOwnCap bytes)
    __attribute__((aligned(16))) uint8_t ring[kInfRing + kInfMirror];
};

// kSerialHeader: block headers and tables by lane 0 alone (inflate_block_rest) -- the form that does not depend on the lane order of the
// returning LDS add; chosen per context when its known-answer test of the wave-built tables fails (zwz_api.cpp), never otherwise.  A
// template parameter, not a kernel argument: the production kernel's code is untouched by the other form's.
template <bool kSerialHeader>
__global__ __launch_bounds__(kInflateThreads, kInflateWavesPerSimd) void inflate_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                                  const uint32_t* __restrict__ in_len, uint32_t n,
                                                                  uint8_t* __restrict__ out, uint64_t out_stride,
                                                                  uint32_t* __restrict__ out_len, uint32_t* __restrict__ status,
                                                                  const uint4* __restrict__ order /* (offset, length, chunk) longest payloads first, or null */) {
    __shared__ InflateWaveMem s_mem[kInflateThreads / 64];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t slot = blockIdx.x * (kInflateThreads / 64) + wave;
    if (slot >= n) return;
    // (the order's entry carries the chunk's offset and length along: read through a chunk number they were a second, dependent
    // round trip at the head of every wave -- a fifth of a stored chunk's whole time)
    const uint4 oe = order ? order[slot] : make_uint4(0, 0, 0, slot);
    const uint32_t chunk = oe.w;
    InflateWaveMem& m = s_mem[wave];
    const uint8_t* src = in + (order ? ((uint64_t)oe.y << 32 | oe.x) : in_off[chunk]);   // 16-byte aligned (API contract)
    uint8_t* dst = out + (size_t)chunk * out_stride;
    const uint32_t nin = order ? oe.z : in_len[chunk];
    const uint32_t nin16 = (nin + 15u) & ~15u;         // readable extent (API contract)

    uint32_t fill_end = 0;                             // ring holds payload bytes [fill_end - 2048, fill_end)
    auto top_up = [&](uint32_t pos) {                  // wave-uniform: make [pos, pos + 1 KiB) resident
        pos = __builtin_amdgcn_readfirstlane(pos);
        if (fill_end + kInfRing < pos + kInfFill) fill_end = pos & ~(kInfFill - 1u);   // jumped (stored block): restart
        while (fill_end < pos + kInfFill && fill_end < nin16) {
            const uint32_t o = fill_end + lane * 16u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (o < nin16) v = *reinterpret_cast<const uint4*>(src + o);
            *reinterpret_cast<uint4*>(m.ring + (o & (kInfRing - 1u))) = v;
            if ((o & (kInfRing - 1u)) < kInfMirror) *reinterpret_cast<uint4*>(m.ring + kInfRing + (o & (kInfRing - 1u))) = v;   // wrap-around mirror
            fill_end += kInfFill;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };

    InflateState st;
#if ZWZ_INF_EXP & 16
    uint64_t istamp_ = __builtin_amdgcn_s_memtime();
    uint32_t iacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    top_up(0);
    uint32_t go = 0;
    if (lane == 0) go = inflate_begin(st, m.ring, nin, kInfRing - 1u) ? 1u : 0u;
    go = __builtin_amdgcn_readfirstlane(go);
    uint32_t fenced = 0;      // every output byte below this offset is visible to the whole wave
    while (go) {
        uint32_t kind = kBlkStop, soff = 0, slen = 0, opos = 0;
        top_up(st.br.pos);
        {   // the block header: lane 0 reads the bits, the wave builds the tables (inflate_core.h: inflate_block_rest, in its pieces)
            uint32_t v = 0, ok = 0;
            if (lane == 0) { opos = st.out_pos; ok = inflate_block_type(st, v) ? 1u : 0u; }
            v = __builtin_amdgcn_readfirstlane(v); ok = __builtin_amdgcn_readfirstlane(ok);
            if (ok && (kSerialHeader || v == 0u || v == 3u)) {
                if (lane == 0) kind = inflate_block_rest(st, kSerialHeader ? &m.t : nullptr, m.lens, v, soff, slen);
                if (kSerialHeader) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            } else if (ok) {
                uint32_t nlen = 288, ndist = 30, good = 1, max_len = 0;
                if (v == 1u) {
                    for (uint32_t i = lane; i < 288u; i += 64u) m.lens[i] = (uint8_t)static_lit_len(i);
                    if (lane < 30u) m.lens[288u + lane] = 5;
                } else {
                    uint8_t* cl = reinterpret_cast<uint8_t*>(m.pos);          // (batch / pos are idle between blocks)
                    if (lane == 0) good = inflate_dyn_begin(st, cl, nlen, ndist) ? 1u : 0u;
                    good = __builtin_amdgcn_readfirstlane(good); nlen = __builtin_amdgcn_readfirstlane(nlen); ndist = __builtin_amdgcn_readfirstlane(ndist);
                    if (good) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        // the code-length code reuses the distance-table slots (7-bit fast index fits in 8)
                        if (wave_build_decode_table(cl, 19u, m.t.dist_fast, 7u, m.t.dist_count, m.t.dist_sym, max_len, nullptr, m.batch) != 0) {
                            if (lane == 0) st.status = kInfDataError;
                            good = 0;
                        }
                    }
                    if (good) {
                        uint32_t bp0 = 0;
                        if (lane == 0) bp0 = st.br.bit_pos();
                        bp0 = __builtin_amdgcn_readfirstlane(bp0);
                        const uint32_t bp1 = wave_dyn_lengths(m.ring, bp0, nin * 8u, m.t.dist_fast, m.lens, nlen + ndist);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        if (lane == 0) {
                            if (bp1 != 0xffffffffu && m.lens[256] != 0) st.br.seek_bit(bp1);
                            else good = inflate_dyn_lengths(st, m.t, m.lens, nlen, ndist) ? 1u : 0u;    // the sequential function, for its exact verdict
                        }
                        good = __builtin_amdgcn_readfirstlane(good);
                    }
                }
                if (good) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const int lr = wave_build_decode_table(m.lens, nlen, m.t.lit_fast, kLitFastBits, m.t.lit_count, m.t.lit_sym, max_len, m.t.lit_walk, m.batch);
                    if (v == 2u) { if (lane == 0) good = inflate_table_ok(st, lr, max_len, true) ? 1u : 0u; good = __builtin_amdgcn_readfirstlane(good); }
                }
                if (good) {
                    const int dr = wave_build_decode_table(m.lens + nlen, ndist, m.t.dist_fast, kDistFastBits, m.t.dist_count, m.t.dist_sym, max_len, m.t.dist_walk, m.batch);
                    if (v == 2u) { if (lane == 0) good = inflate_table_ok(st, dr, max_len, false) ? 1u : 0u; good = __builtin_amdgcn_readfirstlane(good); }
                }
                kind = good ? kBlkHuffman : kBlkStop;
            }
        }
        kind = __builtin_amdgcn_readfirstlane(kind);
        if (kind == kBlkStop) break;
        if (kind == kBlkHuffman) {
            // the two fast tables into the window decode's form (inflate_core.h: pack_lit_entry): kind and bits to skip per entry
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (uint32_t x = lane; x < (1u << kLitFastBits); x += 64u) m.t.lit_fast[x] = pack_lit_entry(m.t.lit_fast[x]);
            for (uint32_t x = lane; x < (1u << kDistFastBits); x += 64u) m.t.dist_fast[x] = pack_dist_entry(m.t.dist_fast[x]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        if (kind == kBlkStored) {
            soff = __builtin_amdgcn_readfirstlane(soff); slen = __builtin_amdgcn_readfirstlane(slen);
            opos = __builtin_amdgcn_readfirstlane(opos);
            uint32_t room = kChunk - opos, cp = slen < room ? slen : room;
            // stored bytes: whole 16-byte vectors of output, each from five aligned input words shifted by the two
            // ranges' relative misalignment (a 5-byte block header sits between them); bytes at the ragged ends singly.
            // (One byte per lane per trip moved 1.7 TB/s with 5 k waves in flight; this is ~10x fewer instructions.)
            {
                const uint32_t head = min(cp, (16u - (opos & 15u)) & 15u);               // dst + opos + head is 16-byte aligned (dst is)
                if (lane < head) dst[opos + lane] = src[soff + lane];
                const uint32_t s0 = soff + head, d0 = opos + head, sh = s0 & 3u;
                const uint32_t* sw = reinterpret_cast<const uint32_t*>(src + (s0 & ~3u));
                uint32_t nvec = (cp - head) >> 4;
                while (nvec && (s0 & ~3u) + 16u * nvec + 4u > nin16) nvec--;            // the fifth word must lie inside the readable extent
                for (uint32_t v = lane; v < nvec; v += 64) {
                    const uint32_t* q = sw + 4u * v;
                    const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
                    *reinterpret_cast<uint4*>(dst + d0 + 16u * v) = make_uint4(__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
                                                                                __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh));
                }
                for (uint32_t i = head + 16u * nvec + lane; i < cp; i += 64) dst[opos + i] = src[soff + i];
            }
            uint32_t stop = 0;
            if (lane == 0) {
                st.out_pos += cp;
                if (cp < slen) st.status = kInfOverflow;
                stop = st.status != kInfRunning;
            }
            if (__builtin_amdgcn_readfirstlane(stop)) break;
        } else {
            // Huffman block.  Decoding is sequential only in where symbols START; what a symbol is,
            // given its start bit, is a pure table lookup.  So every lane decodes the symbol that would
            // start at each of its kWinSlots bit offsets of a 256-bit window (window bit o <-> lane o & 63,
            // slot o >> 6), and a short scalar loop then hops through the true chain with v_readlane,
            // dropping the symbols into the batch registers with v_writelane.  (Lane 0 decoding alone
            // cost ~1900 cycles per symbol.)  Codes longer than the fast tables, and anything odd, fall
            // back to the sequential decoder for one symbol, which also keeps zlib's exact error and
            // truncation behaviour.
            enum : uint32_t { kLit = 0, kMatch = 1, kEob = 2, kSlow = 3, kNeed = 4, kErr = 5 };
            static_assert(kLit == kPkLit && kMatch == kPkMatch && kEob == kPkEob && kSlow == kPkNone && kErr == kPkErr, "the packed entries carry this enum");
            uint32_t bp = 0, opos_u = 0;
            if (lane == 0) { bp = st.br.bit_pos(); opos_u = st.out_pos; }
            bp = __builtin_amdgcn_readfirstlane(bp); opos_u = __builtin_amdgcn_readfirstlane(opos_u);
            const uint32_t total_bits = nin * 8u;
            uint32_t block_done = 0, stop_status = kInfRunning;
            ZWZ_ISTAMP(0);                                                    // block header, tables
            while (!block_done) {
                top_up(bp >> 3);
                // A round looks at up to kWinParts windows of 256 bits one after the other -- decode, follow the orbit, hand the
                // symbols to the batch -- and then does everything that costs per round (output positions, end-of-batch tests,
                // owner map, copy) once for what they yielded together: ~36 symbols on text, against a batch of 64.
                uint32_t nsym = 0;                                            // symbols handed to the batch (wave-uniform)
                uint32_t at = 0;                                              // wave-uniform: the orbit's current offset from bp; kOrbitEnd once it has stopped
                constexpr uint32_t kOrbitEnd = 0xffffu;
#pragma unroll 1
              for (uint32_t wb = 0; wb < 64u * kWinSlots * kWinParts; wb += 64u * kWinSlots) {      // wb: this window's first offset
                // A slot says only what KIND of symbol would start at its bit and how many bits it would take: two table lookups on
                // packed entries (inflate_core.h), no arithmetic on lengths or distances -- 93 % of the slots are not symbols.  The
                // values are decoded once per round, by lane i for symbol i, in the pass behind the orbit.  (Round 3 decoded every
                // slot in full: ~70 vector instructions a slot, half of the kernel's vector work.)
                // (a lane's four slots are 64 bits apart: the same byte alignment and bit shift, eight bytes on -- ONE address and nine
                // consecutive dwords serve all four, where four separate fetches were twelve reads and four address computations)
                uint32_t inf[kWinSlots];
                const uint32_t near_end_s = __builtin_amdgcn_readfirstlane((uint32_t)(bp + wb + 64u * kWinSlots + 64u > total_bits));   // can a symbol of this window reach past the payload?  (a symbol is < 64 bits)
                const uint32_t a0 = bp + wb + lane;
                const uint32_t byte0 = (a0 >> 3) & (kInfRing - 1u);
                uint32_t wd[2u * kWinSlots + 1u];
                {
                    const uint32_t* w = reinterpret_cast<const uint32_t*>(m.ring) + (byte0 >> 2);   // (reads up to 35 bytes past byte0 & ~3: the ring's mirror)
#pragma unroll
                    for (uint32_t i = 0; i < 2u * kWinSlots + 1u; i++) wd[i] = w[i];
                }
#pragma unroll
                for (uint32_t r = 0; r < kWinSlots; r++) {
                    const uint32_t lo = __builtin_amdgcn_alignbyte(wd[2u * r + 1u], wd[2u * r], byte0), hi = __builtin_amdgcn_alignbyte(wd[2u * r + 2u], wd[2u * r + 1u], byte0);   // (the shift's low two bits count)
                    const uint64_t bits = (((uint64_t)hi << 32) | lo) >> (a0 & 7u);     // >= 57 valid bits
                    const uint32_t e = m.t.lit_fast[(uint32_t)bits & ((1u << kLitFastBits) - 1u)];
                    const uint32_t skip = e & 15u;
                    const uint32_t de = m.t.dist_fast[(uint32_t)(bits >> skip) & ((1u << kDistFastBits) - 1u)];
                    const bool is_len = (e >> 13) == kPkLen;
                    uint32_t kind = is_len ? de >> 13 : e >> 13;                          // (the packed kinds are this enum's values)
                    const uint32_t nb = skip + (is_len ? de & 31u : 0u);
                    if (near_end_s) {                                             // (scalar: the payload's last bits -- a symbol that would reach past them, or starts past them)
                        const int32_t avail = (int32_t)total_bits - (int32_t)(a0 + r * 64u);
                        if (avail <= 0 || (kind <= kEob && (int32_t)nb > avail)) kind = kNeed;
                    }
                    inf[r] = kind | (nb << 3);
                }
                // The real symbols are the orbit of offset 0 under "offset -> offset + bits of the symbol decoded there".
                // It is followed by a scalar loop that does nothing but mark: one v_readlane (the next offset, out of the
                // lane that decoded this one), one bit set in a scalar mask, a compare and a branch per symbol -- ~18 symbols a
                // round on text.  (History: a scalar loop that also classified and stored each symbol cost ~50 scalar
                // instructions a symbol, and a CU has one scalar issue slot a cycle for its 20 waves; pointer doubling over the
                // 256 offsets in vector code -- 5-6 rounds of LDS gathers and ballots, ~120 vector instructions a round -- took its
                // place, and the vector unit became what the kernel runs out of.)  Ranks and output positions come from ballots
                // and DPP scans, and one pass of lane-parallel tests finds where the batch ends (64 symbols, end of block, an
                // error, a code for the sequential decoder).
                ZWZ_ISTAMP(1);                                                // refill + the window's 256 decodes
                uint32_t jr[kWinSlots];
#pragma unroll
                for (uint32_t r = 0; r < kWinSlots; r++) {
                    const uint32_t nbits = inf[r] >> 3;
                    jr[r] = (((inf[r] & 7u) <= kMatch) & (nbits != 0u)) ? wb + r * 64u + lane + nbits : kOrbitEnd;   // only literals and matches lead on (and strictly forward)
                }
                uint64_t M[kWinSlots];                                        // reached offsets, slot by slot (wave-uniform)
#pragma unroll
                for (uint32_t r = 0; r < kWinSlots; r++) {
                    uint64_t mk = 0;
                    const uint32_t lim = wb + 64u * (r + 1u);
                    // (the mark is ONE scalar instruction -- s_bitset1_b64 takes the bit number from the low six bits of `at`, as v_readlane does its lane:
                    // offsets are counted from a multiple of 64 -- where `mk |= 1ull << (at - wb - 64 r)` compiled to subtract, shift, or: a hop is a chain of
                    // dependent scalar issues, each a turn among the SIMD's five waves, and six of them became four)
                    auto hop = [&] { asm volatile("s_bitset1_b64 %0, %1" : "+s"(mk) : "s"(at)); at = (uint32_t)__builtin_amdgcn_readlane((int)jr[r], (int)at); };
                    while (at < lim) {                                        // (at >= wb + 64 r: offsets only grow)
                        hop();                                                // four hops a trip by hand (the optimizer does not unroll around a
                        if (at >= lim) break;                                 // v_readlane): three of four loop branches fall through instead of
                        hop();                                                // being taken
                        if (at >= lim) break;
                        hop();
                        if (at >= lim) break;
                        hop();
                    }
                    M[r] = mk;
                }
                // The reached offsets hand their symbols to the batch in orbit order: a marked (slot, lane) writes (kind, value)
                // and (offset, bits) to entry `rank` of an LDS array -- and then ONE pass with lane i on symbol i does what
                // used to be done slot by slot over the sparse window (four scans, four sets of tests, ~190 vector instructions
                // a round): output positions from one scan, the first symbol at which the batch must end (an end of block, a
                // code for the sequential decoder, an error, output that would not fit) from one ballot.  Entry kBatch is the
                // symbol a full batch leaves for the next round.
#pragma unroll
                for (uint32_t r = 0; r < kWinSlots; r++) {
                    const uint32_t rank = nsym + rank_in(M[r]);
                    if (((M[r] >> lane) & 1ull) && rank <= kBatch) m.sym[rank] = (wb + r * 64u + lane) | inf[r] << 10;
                    nsym += (uint32_t)__popcll(M[r]);
                }
                ZWZ_ISTAMP(2);                                                // the orbit, symbols to the batch
                if (at >= kOrbitEnd || nsym > kBatch) break;                  // the orbit has stopped, or the batch is full
              }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                uint32_t cur = 0, k = 0, stop = 0xffu, opos_new = opos_u;
                {
                    // (bitwise on purpose: with && and || the structurizer turns each test into exec-mask control flow)
                    const uint32_t have = (uint32_t)(lane < nsym);             // nsym >= 1: offset 0 is always reached
                    const uint32_t sy = m.sym[lane];                           // (stale beyond nsym: masked by `have`)
                    const uint32_t kd = (sy >> 10) & 7u, s_off = sy & 0x3ffu, s_nb = sy >> 13;
                    // the symbol's value, decoded here and only here: lane i reads symbol i's bits again (same ring, same tables)
                    uint32_t v;
                    {
                        const uint32_t a = bp + s_off;
                        const uint32_t byte = (a >> 3) & (kInfRing - 1u);
                        const uint32_t* w = reinterpret_cast<const uint32_t*>(m.ring) + (byte >> 2);
                        const uint32_t lo = __builtin_amdgcn_alignbyte(w[1], w[0], byte & 3u), hi = __builtin_amdgcn_alignbyte(w[2], w[1], byte & 3u);
                        const uint64_t bits = (((uint64_t)hi << 32) | lo) >> (a & 7u);
                        const uint32_t e = m.t.lit_fast[(uint32_t)bits & ((1u << kLitFastBits) - 1u)];
                        const uint32_t s = (e >> 4) & 511u, skip = e & 15u;
                        const uint32_t c = (s - 257u) & 31u;                                 // (a length code where kd == kMatch; anything elsewhere)
                        const uint32_t xb = length_extra_bits(c);
                        const uint32_t len = length_base(c) + 3u + ((uint32_t)(bits >> ((skip - xb) & 15u)) & ((1u << xb) - 1u));
                        const uint64_t rest = bits >> skip;
                        const uint32_t de = m.t.dist_fast[(uint32_t)rest & ((1u << kDistFastBits) - 1u)];
                        const uint32_t d = (de >> 5) & 31u, dxb = dist_extra_bits(d) & 15u;
                        const uint32_t dist = dist_base(d) + 1u + ((uint32_t)(rest >> (((de & 31u) - dxb) & 31u)) & ((1u << dxb) - 1u));
                        v = kd == kMatch ? (len << 16) | dist : kd == kLit ? s : 0u;
                    }
                    const uint32_t is_lit = have & (uint32_t)(kd == kLit), is_match = have & (uint32_t)(kd == kMatch), mlen = v >> 16;
                    const uint32_t ol = is_lit + is_match * mlen;
                    const uint32_t sc = wave_scan_incl(ol);
                    const uint32_t pos = opos_u + sc - ol;
                    const uint32_t ends = have & ((uint32_t)(kd >= kEob) | (is_lit & (uint32_t)(pos >= kChunk)) |
                                                  (is_match & ((uint32_t)((v & 0xffffu) > pos) | (uint32_t)(pos + mlen > kChunk))));
                    const uint64_t C = __ballot(ends != 0);
                    if (C) {                                               // wave-uniform: the round ends at this symbol
                        const uint32_t lc = (uint32_t)__builtin_ctzll(C);
                        const uint32_t c_kind = (uint32_t)__builtin_amdgcn_readlane((int)kd, (int)lc), c_val = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lc);
                        const uint32_t c_off = (uint32_t)__builtin_amdgcn_readlane((int)s_off, (int)lc), c_nb = (uint32_t)__builtin_amdgcn_readlane((int)s_nb, (int)lc);
                        opos_new = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)lc);
                        cur = c_off;
                        k = lc;
                        if (c_kind == kLit) { stop = kErr; stop_status = kInfOverflow; }
                        else if (c_kind == kMatch) { stop = kErr; stop_status = (c_val & 0xffffu) > opos_new ? kInfDataError : kInfOverflow; }
                        else if (c_kind == kEob) { cur += c_nb; stop = kEob; }
                        else { stop = c_kind; if (c_kind == kNeed) stop_status = kInfNeedInput; else if (c_kind == kErr) stop_status = kInfDataError; }
                    } else if (nsym > kBatch) {                            // batch full: the next round starts at symbol kBatch
                        k = kBatch;
                        cur = m.sym[kBatch] & 0x3ffu;
                        opos_new = opos_u + (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
                    } else {                                               // plain symbols all the way: the chain leaves the window behind the last of them
                        k = nsym;
                        cur = (uint32_t)__builtin_amdgcn_readlane((int)(s_off + s_nb), (int)(nsym - 1u));
                        opos_new = opos_u + (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
                    }
                    m.batch[lane] = v;
                    m.pos[lane] = lane < k ? pos : 0xffffffffu;            // symbols from the end of the batch on do not belong to it
                }
                bp += cur;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // The wave moves the bytes of this batch, one output byte per lane per trip: find the
                // symbol that produces the byte (binary search over the batch's start offsets), follow
                // back-references that point into this same batch until they land on a literal of the
                // batch or on output of an earlier batch, then load/store.  All loads of a trip are in
                // flight together (copying match by match cost one L2 round trip per match: ~9 ms a chunk).
                // The batch goes through the wave's LDS arrays (a ds_bpermute-based lookup returned wrong
                // owners here; indexed LDS reads are also cheaper than eight bpermutes).
                ZWZ_ISTAMP(3);                                                // one pass over the batch: positions, where it ends
                const uint32_t bstart = opos_u;
                const uint32_t bbytes = opos_new - bstart;
                opos_u = opos_new;
                // Which symbol writes a byte: for batches of up to kOwnCap bytes (nearly all) a per-byte map -- every symbol
                // marks its first byte with 1 + its index, a max-scan carries the marks forward -- so a lookup is one LDS byte;
                // larger batches search the symbols' start offsets (6 dependent LDS reads).  It matters where matches a few
                // bytes back refer to each other (smooth "image-like" data): every hop of the chase below is such a lookup.
                const bool mapped = bbytes <= kOwnCap;                             // wave-uniform
                if (mapped) {
                    for (uint32_t i = lane; i < (bbytes + 15u) >> 4; i += 64u) reinterpret_cast<uint4*>(m.ownb)[i] = make_uint4(0, 0, 0, 0);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (lane < k) m.ownb[m.pos[lane] - bstart] = (uint8_t)(lane + 1u);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    uint32_t carry = 0;
                    for (uint32_t j0 = 0; j0 < bbytes; j0 += 64) {
                        const uint32_t o = max(wave_scan_max_incl((uint32_t)m.ownb[j0 + lane]), carry);
                        m.ownb[j0 + lane] = (uint8_t)o;
                        carry = (uint32_t)__builtin_amdgcn_readlane((int)o, 63);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
                ZWZ_ISTAMP(4);                                                // the owner map
                auto owner = [&](uint32_t pos, uint32_t& ov, uint32_t& op) {      // symbol of the batch that writes byte `pos`
                    uint32_t lo = 0;
                    if (mapped) lo = (uint32_t)m.ownb[pos - bstart] - 1u;
                    else {
#pragma unroll
                        for (uint32_t stp = 32; stp >= 1; stp >>= 1) { const uint32_t q = m.pos[(lo + stp) & 63u]; if (q <= pos) lo += stp; }
                    }
                    ov = m.batch[lo]; op = m.pos[lo];
                };
                // byte `at` of a match that starts at op with distance d comes from here (the modulo only where a match
                // overlaps itself: a wave-uniform branch, rare on text)
                auto source_of = [&](uint32_t at, uint32_t op, uint32_t d, bool live) -> uint32_t {
                    const uint32_t off = at - op;
                    uint32_t from = at - d;
                    if (__ballot(live && off >= d)) from = off < d ? from : op - d + off % (d | (uint32_t)!live);
                    return from;
                };
                // (Round 5, measured and dropped: the sources of four trips worked out first, their loads sent out together, then the stores -- on the
                // idea that a trip's load is an L2 round trip the wave stands through, three a round on text: inflate 22.93 -> 22.87 ms, the copy's share
                // of a wave's time 30 -> 32 %.  With twenty waves a CU the round trips are covered by the other waves; the kernel is bound by vector issue.)
                bool need_fence = false;
                for (uint32_t j0 = 0; j0 < bbytes; j0 += 64) {
                    const uint32_t pos = bstart + j0 + lane;
                    const bool in = j0 + lane < bbytes;
                    uint32_t ov = 0, op = 0;
                    owner(in ? pos : bstart, ov, op);
                    bool lit = ov < 256u;
                    uint32_t src = source_of(pos, op, ov & 0xffffu, in && !lit);
                    // chase references into this batch (wave-uniform loop, lanes drop out as they resolve)
                    while (__ballot(in && !lit && src >= bstart)) {
                        const bool go2 = in && !lit && src >= bstart;
                        uint32_t ov2 = 0, op2 = 0;
                        owner(go2 ? src : bstart, ov2, op2);
                        const uint32_t s2 = source_of(src, op2, ov2 & 0xffffu, go2 && ov2 >= 256u);
                        if (go2) {
                            if (ov2 < 256u) { lit = true; ov = ov2; }
                            else src = s2;
                        }
                    }
                    const uint64_t far = __ballot(in && !lit && src + 1u > fenced);
                    if (far && !(ZWZ_INF_EXP & 2)) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_s_waitcnt(0); fenced = bstart; }   // (& 2: timing only -- what the wait for the previous rounds' stores costs; the output is then wrong now and then)
                    // back-references read bytes this CU stored a moment ago: agent-scope (sc1) loads are served
                    // by L2 and cannot hit a stale L1 line that was cached before the store
                    if (in) dst[pos] = lit ? (uint8_t)ov : __hip_atomic_load(&dst[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    (void)need_fence;
                }
                ZWZ_ISTAMP(5);                                                // the bytes
                if (stop == kEob) block_done = 1;
                else if (stop == kSlow) {
                    // one symbol through the sequential decoder (long code, or its exact failure mode)
                    uint32_t k1 = 0, d1 = 0, nbp = bp, nop = opos_u, stt = kInfRunning;
                    if (lane == 0) {
                        st.br.seek_bit(bp); st.out_pos = opos_u;
                        bool d;
                        k1 = inflate_decode_batch<true>(st, m.t, kChunk, m.batch, m.pos, d, 1u);
                        d1 = d; nbp = st.br.bit_pos(); nop = st.out_pos; stt = st.status;
                    }
                    k1 = __builtin_amdgcn_readfirstlane(k1); d1 = __builtin_amdgcn_readfirstlane(d1);
                    bp = __builtin_amdgcn_readfirstlane(nbp); opos_u = __builtin_amdgcn_readfirstlane(nop);
                    stt = __builtin_amdgcn_readfirstlane(stt);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (k1) {
                        const uint32_t mv = m.batch[0], mp = m.pos[0];
                        if (mv < 256u) { if (lane == 0) dst[mp] = (uint8_t)mv; }
                        else {
                            const uint32_t len = mv >> 16, dist = mv & 0xffffu;
                            const uint32_t from = mp - dist, span = len < dist ? len : dist;
                            if (from + span > fenced) {
                                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                                __builtin_amdgcn_s_waitcnt(0);
                                fenced = mp;
                            }
                            for (uint32_t i = lane; i < len; i += 64) dst[mp + i] = __hip_atomic_load(&dst[from + (i % dist)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    if (d1) { block_done = 1; stop_status = stt; }
                } else if (stop != 0xffu) { block_done = 1; }          // need / error / overflow: status already set
            }
            // hand the position back to lane 0's reader for the next block header
            uint32_t halt = 0;
            if (lane == 0) {
                st.out_pos = opos_u;
                if (stop_status != kInfRunning) st.status = stop_status;
                else st.br.seek_bit(bp);
                halt = st.status != kInfRunning;
            }
            if (__builtin_amdgcn_readfirstlane(halt)) break;
        }
        uint32_t fin = 0;
        if (lane == 0 && st.last) { st.status = kInfEnd; fin = 1; }
        if (__builtin_amdgcn_readfirstlane(fin)) break;
    }
    if (lane == 0) { out_len[chunk] = st.out_pos; status[chunk] = st.status; }
#if ZWZ_INF_EXP & 16
    ZWZ_ISTAMP(6);                                                            // stored blocks, the sequential decoder, the rest
    if (lane == 0) for (uint32_t ph = 0; ph < 8; ph++) atomicAdd(&g_inf_times[ph], (unsigned long long)(iacc_[ph] >> 8));
#endif
}

// ------------------------------------------------------------------------------------------------
// launchers
#define ZWZ_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

// lz_links stands on a property of ds_wrxchg_rtn_b32 the ISA manual does not spell out: lanes of one instruction that name the
// same address are served in ascending lane order (each gets what the nearest lower such lane wrote, the lowest gets the old
// value, the highest lane's value stays).  zwz_ctx_create checks it on the device before anything else runs, under the
// conditions lz_links creates: one workgroup of lz_links' shape on EVERY CU of every XCD (two per CU are launched, the LDS
// lets one be resident), the real 128 KiB table, bucket patterns from all-equal to spread over the whole table, and eight
// other waves of the workgroup streaming 16-byte reads and writes through the rest of the LDS the whole time, as the
// feeders do.  A context whose device fails the check is not created (ZWZ_E_NO_DEVICE): there is no other flavour.
// Per round: plain read of each lane's bucket, the exchange, plain read again; a lane must have received the value of the
// nearest lower lane naming its bucket (else what the bucket held), and the bucket must hold the highest such lane's value.
constexpr uint32_t kProbeRounds = 384;
__global__ __launch_bounds__(kLinksThreads) void exchange_order_probe_kernel(uint32_t* __restrict__ violations) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tab[];            // 32768 buckets, then the traffic area
    constexpr uint32_t kBuckets = 32768u;
    uint32_t* area = tab + kBuckets + 4u;                                     // 16 KiB, as lz_links' two hand-over buffers
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    for (uint32_t i = tid; i < kBuckets; i += kLinksThreads) tab[i] = 0xdead0000u + i;
    for (uint32_t i = tid; i < 4096u; i += kLinksThreads) area[i] = i;
    __syncthreads();
    if (wave != 0) {
        // feeder-like traffic: every lane streams 16-byte vectors through its own part of the area; the loop is bounded, so
        // every wave reaches the end whatever wave 0 does
        uint4* a4 = reinterpret_cast<uint4*>(area) + (wave - 1u) * 128u + lane;
        uint4 v = a4[0];
        for (uint32_t it = 0; it < 3u * kProbeRounds; it++) {
            a4[64 * (it & 1u)] = make_uint4(v.x + it, v.y ^ it, v.z + lane, v.w);
            v = a4[64 * ((it + 1u) & 1u)];
        }
        if (v.x == 0xffffffffu && v.y == 0x12345678u) violations[1] = v.z;   // (keeps the loop alive; never true in practice)
        return;
    }
    typedef __attribute__((address_space(3))) uint32_t* lds_word_ptr;
    const uint32_t tab_a = (uint32_t)(uintptr_t)(lds_word_ptr)tab;
    uint32_t bad = 0, rng = 0x9e3779b9u * (lane + 1u) + 0x85ebca6bu * (blockIdx.x + 1u);
    for (uint32_t r = 0; r < kProbeRounds; r++) {
        rng = rng * 1664525u + 1013904223u;
        const uint32_t spread = 1u << (3u * (r % 6u));                           // 1, 8, 64, 512, 4096, 32768 buckets ...
        const uint32_t base = (0x9e3779b1u * (r + 17u * blockIdx.x)) & (kBuckets - 1u) & ~(spread - 1u);   // ... anywhere in the table
        const uint32_t a = base + ((rng >> 11) & (spread - 1u));
        const uint32_t la = tab_a + 4u * a, mine = 0x10000u * (r + 1u) + lane;
        uint32_t pre, old, post;
        asm volatile("ds_read_b32 %0, %3\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "ds_wrxchg_rtn_b32 %1, %3, %4\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "ds_read_b32 %2, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(pre), "=&v"(old), "=&v"(post) : "v"(la), "v"(mine) : "memory");
        uint32_t want = pre, last = lane;
        for (uint32_t j = 0; j < 64; j++) {
            const uint32_t aj = (uint32_t)__builtin_amdgcn_readlane((int)a, (int)j);
            if (aj != a) continue;
            if (j < lane) want = 0x10000u * (r + 1u) + j;
            if (j > lane) last = j;
        }
        bad += (uint32_t)(old != want) + (uint32_t)(post != 0x10000u * (r + 1u) + last);
    }
    if (bad) atomicAdd(violations, bad);
}

hipError_t probe_exchange_order(hipStream_t s, bool* holds) {
    uint32_t* d = nullptr;
    uint32_t h[2] = {1, 0};
    *holds = false;
    int dev = 0, cus = 0;
    ZWZ_TRY(hipGetDevice(&dev));
    ZWZ_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus <= 0) cus = 256;
    ZWZ_TRY(hipMalloc(&d, 2 * sizeof(uint32_t)));
    hipError_t e = hipMemsetAsync(d, 0, 2 * sizeof(uint32_t), s);
    if (e == hipSuccess) { hipLaunchKernelGGL(exchange_order_probe_kernel, dim3(2u * (uint32_t)cus), dim3(kLinksThreads), kLinksLdsBytes, s, d); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    *holds = e == hipSuccess && h[0] == 0;
    return e;
}

// lz_links: a workgroup per CU (a.cu_count; the self-test passes fewer to make every workgroup cross chunk boundaries), chunks
// strided over them; an epoch is 16 bits, so a launch covers at most 65 535 chunks per workgroup
static hipError_t launch_links(const DeflateArgs& a, hipStream_t s, bool listed) {
    const uint32_t cus = a.cu_count ? a.cu_count : 256u;
    if (listed) {   // lz_dense_list's sparse chunks (link_stat is its array of marks: a sparse chunk's starts at 0)
        const uint32_t G = a.n < cus ? a.n : cus;
        if ((uint64_t)a.n > 65535ull * G) return hipErrorInvalidValue;      // (an epoch is 16 bits; max_batch keeps slices far below)
        hipLaunchKernelGGL(lz_links_kernel, dim3(G), dim3(kLinksThreads), kLinksLdsBytes, s, a.in, a.in_off, a.in_len, a.links, a.link_stat, a.n,
                           a.sparse_list, a.tickets + kTicketSparseCount);
        return hipGetLastError();
    }
    ZWZ_TRY(hipMemsetAsync(a.link_stat, 0, (size_t)a.n * sizeof(uint32_t), s));
    for (uint32_t done = 0; done < a.n;) {
        const uint32_t left = a.n - done, G = left < cus ? left : cus;
        const uint64_t cap = 65535ull * G;
        const uint32_t m = left < cap ? left : (uint32_t)cap;
        hipLaunchKernelGGL(lz_links_kernel, dim3(G), dim3(kLinksThreads), kLinksLdsBytes, s, a.in, a.in_off + done, a.in_len + done,
                           a.links + (size_t)done * kLinkStride, a.link_stat + done, m, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
        done += m;
    }
    return hipGetLastError();
}

// lz_links alone over a batch (zwz_ctx_create's known-answer test of the hand-scheduled kernel)
hipError_t launch_links_only(const DeflateArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    return launch_links(a, s, false);
}

uint32_t exp_flags_kernels() { return (uint32_t)(ZWZ_MATCH_EXP) | (uint32_t)(ZWZ_PARSE_EXP) << 8 | (uint32_t)(ZWZ_ENC_EXP) << 16 | (uint32_t)(ZWZ_INF_EXP) << 24; }

hipError_t configure_kernels() {
    ZWZ_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lz_links_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLinksLdsBytes));
    ZWZ_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(exchange_order_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLinksLdsBytes));
    ZWZ_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(lz_match_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMatchLdsBytes));
    ZWZ_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(encode_kernel<kEncodeThreads, kOutWords, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEncodeLdsBytes));
    ZWZ_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(encode_kernel<kSmallEncThreads, kSmallEncOutWords, kSmallEncWgs>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmallEncLdsBytes));
    ZWZ_TRY(configure_lazy_kernels());
    return configure_band_kernels();
}

hipError_t launch_deflate(const DeflateArgs& a, hipStream_t s, hipEvent_t* ev /* kNumDeflateStages + 1 or null */) {
    if (a.n == 0) return hipSuccess;
    ZWZ_TRY(hipMemsetAsync(a.tickets, 0, kTicketBytes, s));
    if (ev) ZWZ_TRY(hipEventRecord(ev[0], s));
    // Chain-heavy chunks (lz_dense_list: a sample of each chunk's trigrams): positions sorted by (bucket, position), then the banded
    // search (zwz_band.hip).  The rest: chain links, then lz_match's screening pass over them.  ZWZ_MATCH=walk sends every chunk
    // through links + lz_match (its sorted walk included), =band every chunk through the band.
    const uint32_t which = a.match_mode;         // the context's option (zwz_ctx_set_option; ZWZ_MATCH is read once, at zwz_ctx_create)
    const bool lazy = which == kMatchLazy || which == kMatchAutoLazy || (which == kMatchAuto && kAutoIsLazy);      // chain-heavy chunks: sort + lz_lazy (search and parse in one), else sort + band + lz_parse
    const uint32_t force = (which == kMatchBand || which == kMatchLazy) ? 2u : 0u;    // 2: every chunk counts as chain-heavy
    // (stage 0 of the profile = everything that prepares the search: marks and lists, chain links of the sparse chunks, the sorted
    // arrays of the chain-heavy ones; stage 1 = the searches themselves: lz_match, lz_match_band)
    if (which == 1u) ZWZ_TRY(launch_links(a, s, false));
    else {
        ZWZ_TRY(launch_dense_list(a, s, force));
        ZWZ_TRY(launch_links(a, s, true));
        ZWZ_TRY(launch_sort(a, s));
        ZWZ_TRY(launch_place(a, s));
    }
    if (ev) ZWZ_TRY(hipEventRecord(ev[1], s));
    hipLaunchKernelGGL(lz_match_kernel, dim3(a.n), dim3(kMatchThreads), kMatchLdsBytes, s, a.in, a.in_off, a.in_len,
                       a.links, a.entries, a.has128, a.perm, a.link_stat, which == 1u ? 0u : force == 2u ? 2u : 1u, a.tickets + 40);
#if ZWZ_MATCH_EXP & 16
    if (getenv("ZWZ_MATCH_TIMES")) {
        uint32_t h[64];
        ZWZ_TRY(hipStreamSynchronize(s));
        ZWZ_TRY(hipMemcpy(h, a.tickets, sizeof h, hipMemcpyDeviceToHost));
        fprintf(stderr, "ZWZ_MATCH_TIMES n=%u stage0=%u stage=%u screen=%u refine=%u search=%u wait=%u flush_slide=%u\n", a.n, h[40], h[46], h[41], h[42], h[43], h[44], h[45]);
    }
#endif
    if (which != 1u && lazy) ZWZ_TRY(launch_lazy(a, s));
    if (which != 1u && !lazy) {
        ZWZ_TRY(launch_match_band(a, s));
        if ((exp_flags_band() & 16u) && getenv("ZWZ_BAND_TIMES")) {       // experiment builds (zwz_band.hip, ZWZ_BAND_EXP & 16): cycles >> 8 per phase, summed over the workgroups' first threads
            uint32_t h[64];
            ZWZ_TRY(hipStreamSynchronize(s));
            ZWZ_TRY(hipMemcpy(h, a.tickets, sizeof h, hipMemcpyDeviceToHost));
            fprintf(stderr, "ZWZ_BAND_TIMES n=%u copy=%u scan=%u build=%u count=%u order=%u pass1=%u compact=%u pass2=%u flush=%u ticket=%u\n", a.n, h[16], h[17], h[18], h[19], h[24], h[20], h[25], h[21], h[22], h[23]);
        }
    }
    if (ev) ZWZ_TRY(hipEventRecord(ev[2], s));
    hipLaunchKernelGGL(lz_parse_kernel, dim3((a.n + kParseThreads / 64 - 1) / (kParseThreads / 64)), dim3(kParseThreads), 0, s, a.in_len, a.n, a.entries, a.has128, a.sym, a.mst, a.info, a.links,
                       (which != 1u && lazy) ? a.link_stat : (const uint32_t*)nullptr);
#if ZWZ_PARSE_EXP & 16
    if (getenv("ZWZ_PARSE_TIMES")) {
        unsigned long long h[8], z[8] = {0};
        ZWZ_TRY(hipStreamSynchronize(s));
        ZWZ_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_parse_times), sizeof h));
        ZWZ_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_parse_times), z, sizeof z));
        fprintf(stderr, "ZWZ_PARSE_TIMES n=%u fetch=%llu steps=%llu orbit=%llu events=%llu cover=%llu out=%llu\n", a.n, h[0], h[1], h[2], h[3], h[4], h[5]);
    }
#endif
    if (ev) ZWZ_TRY(hipEventRecord(ev[3], s));
    hipLaunchKernelGGL(blockify_kernel, dim3(a.n), dim3(kBlockifyThreads), 0, s, a.in, a.in_off, a.in_len, a.entries, a.sym, a.mst,
                       a.info, a.blocks, a.links);
    if (ev) ZWZ_TRY(hipEventRecord(ev[4], s));
    ZWZ_TRY(launch_plan(a, s));
    if (ev) ZWZ_TRY(hipEventRecord(ev[5], s));
    {
        uint32_t* huff_list = reinterpret_cast<uint32_t*>(a.perm);          // lz_match's work-order array is dead by now
        uint32_t* small_list = huff_list + a.n;                             // (plan's list of open blocks is dead too)
        hipLaunchKernelGGL(encode_stored_kernel, dim3(a.n), dim3(kEncStoredThreads), 0, s, a.in, a.in_off, a.in_len, a.info, a.blocks, a.plans,
                           a.out, a.out_stride, a.out_len, huff_list, small_list, a.tickets);
        const uint32_t cus = a.cu_count ? a.cu_count : 256u;
        hipLaunchKernelGGL((encode_kernel<kEncodeThreads, kOutWords, 8>), dim3(a.n < 2u * cus ? a.n : 2u * cus), dim3(kEncodeThreads), kEncodeLdsBytes, s, a.in, a.in_off, a.in_len, a.entries, a.sym,
                           a.mst, a.info, a.blocks, a.plans, a.out, a.out_stride, a.out_len, a.links, huff_list, a.tickets, (uint32_t)kTicketHuffCount, (uint32_t)kTicketHuffNext);
        hipLaunchKernelGGL((encode_kernel<kSmallEncThreads, kSmallEncOutWords, kSmallEncWgs>), dim3(a.n < kSmallEncWgs * cus ? a.n : kSmallEncWgs * cus), dim3(kSmallEncThreads), kSmallEncLdsBytes, s, a.in, a.in_off, a.in_len, a.entries, a.sym,
                           a.mst, a.info, a.blocks, a.plans, a.out, a.out_stride, a.out_len, a.links, small_list, a.tickets, (uint32_t)kTicketSmallCount, (uint32_t)kTicketSmallNext);
    }
#if ZWZ_ENC_EXP & 16
    if (getenv("ZWZ_ENC_TIMES")) {
        unsigned long long h[8], z[8] = {0};
        ZWZ_TRY(hipStreamSynchronize(s));
        ZWZ_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_enc_times), sizeof h));
        ZWZ_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_enc_times), z, sizeof z));
        fprintf(stderr, "ZWZ_ENC_TIMES n=%u setup=%llu zero_hdr=%llu adler=%llu pass1=%llu wait1=%llu pass2=%llu wait2=%llu out=%llu\n", a.n, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
#endif
    if (ev) ZWZ_TRY(hipEventRecord(ev[6], s));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// md5_files: RFC 1321 digest of whole files whose bytes sit in consecutive chunk slots (SURVEY.md §8 f1;
// md5_of_file(), verification.cpp:6-30).  MD5 is a chain over a file's 64-byte blocks, so the parallel axis
// is the file: one lane per file, the sixteen message words and the state in registers, the 64 steps fully
// unrolled (shared with the host class through ZWZ_MD5_STEPS).  A file's bytes are not contiguous -- slots
// hold 65 535 bytes on a 65 536 stride -- so words are fetched through a small cursor that steps from slot
// to slot and assembles the one word per chunk that straddles a boundary byte by byte.
struct Md5Cursor {
    const uint8_t* in; const uint64_t* off; const uint32_t* len;
    uint32_t slot; const uint8_t* ptr; uint32_t rem;
    __device__ __forceinline__ void open(uint32_t s) { slot = s; ptr = in + off[s]; rem = len[s]; }
    __device__ __forceinline__ uint32_t byte() {           // caller guarantees the file has bytes left
        while (rem == 0) open(slot + 1);
        rem--;
        return *ptr++;
    }
    __device__ __forceinline__ uint32_t word() {
        if (rem >= 4u) { uint32_t v; __builtin_memcpy(&v, ptr, 4); ptr += 4; rem -= 4; return v; }
        uint32_t v = byte();
        v |= byte() << 8; v |= byte() << 16; v |= byte() << 24;
        return v;
    }
};

__global__ __launch_bounds__(64) void md5_files_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                       const uint32_t* __restrict__ in_len, const uint32_t* __restrict__ files,
                                                       uint32_t n_files, uint32_t* __restrict__ digests) {
    const uint32_t f = blockIdx.x * 64u + threadIdx.x;
    if (f >= n_files) return;
    const uint32_t slot0 = files[2 * f], nslots = files[2 * f + 1];
    uint64_t total = 0;
    for (uint32_t i = 0; i < nslots; i++) total += in_len[slot0 + i];
    auto rol = [](uint32_t x, int sh) { return __builtin_rotateleft32(x, (uint32_t)sh); };
    uint32_t h0 = 0x67452301u, h1 = 0xefcdab89u, h2 = 0x98badcfeu, h3 = 0x10325476u;
    uint32_t w[16];
#define ZWZ_MD5_BLOCK { uint32_t a = h0, b = h1, c = h2, d = h3; ZWZ_MD5_STEPS h0 += a; h1 += b; h2 += c; h3 += d; }
    Md5Cursor cur{in, in_off, in_len, 0, nullptr, 0};
    if (nslots) cur.open(slot0);
    for (uint64_t blk = 0; blk < total / 64u; blk++) {
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) w[j] = cur.word();
        ZWZ_MD5_BLOCK
    }
    // padding: the rest of the message, 0x80, zeros, the bit length in the last two words of a block
    const uint32_t r = (uint32_t)(total & 63u);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) {
        uint32_t v = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t pos = 4u * j + k;
            const uint32_t bt = pos < r ? cur.byte() : (pos == r ? 0x80u : 0u);
            v |= bt << (8u * k);
        }
        w[j] = v;
    }
    const uint64_t bits = total * 8u;
    if (r >= 56u) {
        ZWZ_MD5_BLOCK
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) w[j] = 0;
    }
    w[14] = (uint32_t)bits; w[15] = (uint32_t)(bits >> 32);
    ZWZ_MD5_BLOCK
#undef ZWZ_MD5_BLOCK
    digests[4 * f] = h0; digests[4 * f + 1] = h1; digests[4 * f + 2] = h2; digests[4 * f + 3] = h3;   // little-endian: the 16 digest bytes
}

hipError_t launch_md5_files(const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, const uint32_t* files, uint32_t n_files,
                            uint32_t* digests, hipStream_t s) {
    if (n_files == 0) return hipSuccess;
    hipLaunchKernelGGL(md5_files_kernel, dim3((n_files + 63u) / 64u), dim3(64), 0, s, in, in_off, in_len, files, n_files, digests);
    return hipGetLastError();
}

// A chunk is one wave's work from start to end, and a full-size chunk takes a lone wave milliseconds: in a batch of mixed sizes the
// long ones must start first or they are the batch's tail (370 000 image-like files: the few 64 KB chunks among the 7 KB ones).
// Counting sort of the chunk numbers by payload length (256-byte classes), longest first; one workgroup, ~20 us per 50 000.
__global__ __launch_bounds__(1024) void inflate_order_kernel(const uint64_t* __restrict__ in_off, const uint32_t* __restrict__ in_len, uint32_t n, uint4* __restrict__ order) {
    __shared__ uint32_t s_bin[256];
    const uint32_t tid = threadIdx.x;
    if (tid < 256u) s_bin[tid] = 0;
    __syncthreads();
    auto cls = [](uint32_t len) { const uint32_t c = len >> 8; return 255u - (c < 255u ? c : 255u); };      // longest first
    // (lanes of a wave that share a class add to its counter as one: a batch of equal-sized chunks would otherwise be
    // 50 000 additions to the same LDS word, one after the other)
    // (.. and a wave whose lanes are of MANY classes -- 370 000 log-normal files: thirty classes a wave, thirty turns of the loop below, 3.4 ms a
    // launch on the one workgroup -- lets the LDS sort its adds out: a returning add a lane.  Round 5: 3.4 -> see DESIGN.md)
    auto grouped_add = [&](uint32_t c, bool live) -> uint32_t {           // -> the lane's slot; wave-uniform control flow
        uint32_t slot = 0;
        uint64_t todo = __builtin_amdgcn_ballot_w64(live);
        if (todo) {
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)__builtin_ctzll(todo));
            const uint64_t m = __builtin_amdgcn_ballot_w64(live && c == c0);
            if (__popcll(todo & ~m) > 8) {                                // more than a few lanes of other classes
                if (live) slot = atomicAdd(&s_bin[c], 1u);
                return slot;
            }
        }
        while (todo) {
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)__builtin_ctzll(todo));
            const uint64_t m = __builtin_amdgcn_ballot_w64(live && c == c0);
            uint32_t base = 0;
            if (lane_id() == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&s_bin[c0], (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
            if (live && c == c0) slot = base + rank_in(m);
            todo &= ~m;
        }
        return slot;
    };
    for (uint32_t i = tid; i - tid < n; i += 1024u) (void)grouped_add(i < n ? cls(in_len[i]) : 0u, i < n);
    __syncthreads();
    // A batch of one or two sizes (incompressible files: full chunks and 4-byte tails) keeps the order it came in: there is no
    // tail to shorten, and its copies, which run at HBM speed, lost a fifth of it when neighbouring slots were no longer
    // neighbouring waves.
    const bool keep = __syncthreads_count(tid < 256u && s_bin[tid] != 0u) <= 2;
    if (keep) {
        for (uint32_t i = tid; i < n; i += 1024u) { const uint64_t off = in_off[i]; order[i] = make_uint4((uint32_t)off, (uint32_t)(off >> 32), in_len[i], i); }
        return;
    }
    if (tid < 64u) {                                                       // exclusive scan of the 256 counts by one wave, four a lane
        const uint32_t a = s_bin[4 * tid], b = s_bin[4 * tid + 1], c = s_bin[4 * tid + 2], d = s_bin[4 * tid + 3];
        const uint32_t own = a + b + c + d, ex = wave_scan_incl(own) - own;
        s_bin[4 * tid] = ex; s_bin[4 * tid + 1] = ex + a; s_bin[4 * tid + 2] = ex + a + b; s_bin[4 * tid + 3] = ex + a + b + c;
    }
    __syncthreads();
    for (uint32_t i = tid; i - tid < n; i += 1024u) {
        const uint32_t len = i < n ? in_len[i] : 0u;
        const uint32_t slot = grouped_add(cls(len), i < n);
        if (i < n) { const uint64_t off = in_off[i]; order[slot] = make_uint4((uint32_t)off, (uint32_t)(off >> 32), len, i); }
    }
}

hipError_t launch_inflate(const InflateArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    const uint32_t per = kInflateThreads / 64;
    if (a.order) hipLaunchKernelGGL(inflate_order_kernel, dim3(1), dim3(1024), 0, s, a.in_off, a.in_len, a.n, a.order);
    if (a.serial_header) hipLaunchKernelGGL(inflate_kernel<true>, dim3((a.n + per - 1) / per), dim3(kInflateThreads), 0, s, a.in, a.in_off, a.in_len, a.n, a.out,
                                            a.out_stride, a.out_len, a.status, (const uint4*)a.order);
    else hipLaunchKernelGGL(inflate_kernel<false>, dim3((a.n + per - 1) / per), dim3(kInflateThreads), 0, s, a.in, a.in_off, a.in_len, a.n, a.out,
                            a.out_stride, a.out_len, a.status, (const uint4*)a.order);
#if ZWZ_INF_EXP & 16
    if (getenv("ZWZ_INF_TIMES")) {
        unsigned long long h[8], z[8] = {0};
        ZWZ_TRY(hipStreamSynchronize(s));
        ZWZ_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_inf_times), sizeof h));
        ZWZ_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_inf_times), z, sizeof z));
        fprintf(stderr, "ZWZ_INF_TIMES n=%u header=%llu decode=%llu orbit=%llu symbols=%llu owners=%llu copy=%llu other=%llu\n", a.n, h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
    }
#endif
    return hipGetLastError();
}

}  // namespace zwz
