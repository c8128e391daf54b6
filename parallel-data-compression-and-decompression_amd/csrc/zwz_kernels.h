// zwz_kernels.h -- HBM layout of the batch pipeline's intermediates and the launcher interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "huff_core.h"
#include "zwz_common.h"
#include "lz_band.h"

namespace zwz {

// Per-chunk strides of the intermediates (elements).  Everything is indexed [chunk][position].
constexpr uint32_t kLinkStride = 65536;    // uint16 chain links
constexpr uint32_t kEntryStride = 65536;   // uint2 (e128, e32) match records
constexpr uint32_t kLinksThreads = 576;                            // one inserter wave + eight feeder waves per chunk
constexpr uint32_t kLinksLdsBytes = 131072 + 16 + 2 * 2048 * 4 + 512;  // 32-bit head table + spare slot + two bucket-address/link buffers + slack for read-ahead
constexpr uint32_t kTile = 16384, kTilesPerChunk = 4;
constexpr uint32_t kMatchThreads = 1024;
// lz_match's window is a RING (round 4; rounds 1-3 slid a linear window down after every tile: two LDS passes over 147 KB and two barriers a
// slide, a sixth of the kernel on incompressible data).  Position x lives at ring index x - (x >= kMatchRing ? kMatchRing : 0): positions are below
// 65 536 + lookahead < 2 * kMatchRing, so the modulo is a subtract and a minimum.  The ring must hold, while tile [ts, te) is searched, the history
// zlib may look at and the lookahead of the tile's last positions, in whole 16-byte vectors: [ts - 32 512, te + 272) = 49 168 bytes -- and a multiple
// of 16, so that no vector and no group of four positions straddles its end.  Reads that run on from a ring index (the scan's lookahead, a
// candidate's bytes: up to 258 + 8 of them) find the ring's first kMatchMirror bytes repeated behind its end.
constexpr uint32_t kMatchRing = 49168, kMatchMirror = 288;
constexpr uint32_t kMatchDataBytes = kMatchRing + kMatchMirror;  // 49 456
constexpr uint32_t kMatchLinkBytes = 2 * kMatchRing;            // 98 336: links of the same positions (no mirror: single 16-bit reads, vectors inside a tile)
constexpr uint32_t kMatchListBytes = 4096 + 9728;               // bucket counts of a sorted tile (8 KB) / per-wave work lists of a sparse tile (432 entries a wave)
constexpr uint32_t kMatchLdsBytes = kMatchDataBytes + kMatchLinkBytes + 2048 + kMatchListBytes;   // + has128 bits: 163 664 of 163 840 (with 80 static)
static_assert(kMatchRing % 16 == 0 && kMatchRing >= 32512 + 16384 + 272 && kMatchMirror % 16 == 0 && kMatchMirror >= 258 + 8 + 16, "lz_match ring");
constexpr uint32_t kSortedStride = 65536;  // uint32 (bucket << 16 | position) words of lz_sort, sorted by (bucket, position)
constexpr uint32_t kSortThreads = 256;                           // lz_sort: 64 KiB of packed counters, two workgroups a CU
constexpr uint32_t kPlaceThreads = 1024, kPlaceLdsBytes = 131072;  // lz_place: the chunk's sorted positions, 16 bits each
// lz_lazy (zwz_lazy.hip): a lane per 128-position segment of the chunk.  Its scratch is the chunk's entries space (which the band would
// fill): the step memo of every position, G (fresh-search marks of lanes beyond their segment), and -- written by lz_sort -- the ends of
// the 32 768 buckets in the sorted array (16 bits each).
constexpr uint32_t kLazyThreads = 512, kLazySeg = 128;
constexpr uint32_t kLazyScratchWords = 2 * kEntryStride, kLazyStepOff = 0, kLazyMarkOff = 65536, kLazyBendOff = 65536 + 2048;   // 32-bit words
static_assert(kLazyBendOff + 16384 <= kLazyScratchWords, "lz_lazy scratch fits the entries space");
constexpr uint32_t kBandThreads = 1024;
#ifndef ZWZ_BAND_TILE
#define ZWZ_BAND_TILE 6016
#endif
// sorted entries per tile of lz_match_band: 94 groups of 64 -- a full chunk's 65 533 entries are ELEVEN tiles (5 632, rounds 3-4a: twelve; the per-tile
// phases and the 128-entry halo are paid per tile: text match stage 70.2 -> 68.3 ms; 5 120: 70.9 ms; ten tiles would need 167 KB of LDS)
constexpr uint32_t kBandTile = ZWZ_BAND_TILE;
constexpr uint32_t kBandLdsBytes = (65536 + 64) + (kBandTile + 128) * 12 + kBandTile * 2 + 8192;   // bytes, words + 8-byte comparison words, counts, has128 bits: 159 552 at 6 016
// lz_parse and inflate give a chunk to a WAVE; a workgroup of one wave frees its LDS and its slot the moment that chunk is done, where four waves a
// workgroup (rounds 1-4a) waited for their slowest: lz_parse 15.3 -> 13.7 ms on text, 6.8 -> 6.0 ms per 100 000 small files; inflate 23.4 -> 22.8 / 13.0 -> 12.7
// (128 threads: no change either way)
#ifndef ZWZ_PARSE_THREADS
#define ZWZ_PARSE_THREADS 64
#endif
constexpr uint32_t kParseThreads = ZWZ_PARSE_THREADS;            // chunks per workgroup = threads / 64
constexpr uint32_t kBlockifyThreads = 256;
constexpr uint32_t kEncodeThreads = 1024;                       // == kMaskWords
constexpr uint32_t kOutWords = 16384;                           // 65536-byte staging, first 65535 kept
constexpr uint32_t kEncQueue = 128;                               // entries of a wave's queue of symbol starts (a power of two >= 127)
#ifndef ZWZ_SMALL_ENC_BYTES
#define ZWZ_SMALL_ENC_BYTES 12288
#define ZWZ_SMALL_ENC_OUT_WORDS 4096
#define ZWZ_SMALL_ENC_WGS 6
#endif
constexpr uint32_t kSmallEncBytes = ZWZ_SMALL_ENC_BYTES, kSmallEncThreads = 256, kSmallEncOutWords = ZWZ_SMALL_ENC_OUT_WORDS, kSmallEncWgs = ZWZ_SMALL_ENC_WGS;   // encode's small form: chunks of one block and <= 12 KB, 16 KB of staging, four waves
constexpr uint32_t kSmallEncLdsBytes = kSmallEncOutWords * 4 + (kSmallEncThreads / 64) * 128 * 4 + 5 * (288 * 2 + 32 * 2 + 288 + 32);
static_assert(kEncQueue == 128 && kMaxBlocks == 5, "kSmallEncLdsBytes is written out with these");
constexpr uint32_t kEncodeLdsBytes = kOutWords * 4 + (kEncodeThreads / 64) * kEncQueue * 4 + kMaxBlocks * (288 * 2 + 32 * 2 + 288 + 32);   // 78 528 bytes: two workgroups per CU (with __launch_bounds__(1024, 8): 64 registers)
#ifndef ZWZ_INFLATE_THREADS
#define ZWZ_INFLATE_THREADS 64
#endif
constexpr uint32_t kInflateThreads = ZWZ_INFLATE_THREADS;       // chunks per workgroup = threads / 64
constexpr int kNumDeflateStages = 6;

struct ChunkInfo { uint32_t n_sym, n_blocks; };

struct BlockInfo {
    uint16_t lfreq[kLCodes];
    uint16_t dfreq[kDCodes];
    uint32_t start, end;       // raw byte range covered
    uint32_t flush_pos;        // zlib's strstart at the loop top before this block's flush
    uint32_t first_sym;
};

struct BlockOut {
    uint32_t type, hdr_bits, body_bits, eob_len, eob_code;
    uint32_t hdr[kHdrWords];
    uint16_t lcode[kLCodes + 2];
    uint16_t dcode[kDCodes + 2];
    uint8_t llen[kLCodes + 2];
    uint8_t dlen[kDCodes + 2];
};

// Stored-block shortcut, stage 1 -> stage 2: a block's histograms sorted ascending, plus the exact
// sums the lower bound needs (huff_core.h: shortcut_type).
enum : uint32_t { kProbeNone = 0, kProbeOpen = 1, kProbeStored = 2, kProbeStatic = 3 };
struct BlockProbe {
    uint32_t static_len, extra_bits, used, m_l, m_d, stored_len;
    uint32_t state;               // kProbe*: not a block / probed, undecided / "stored" certain / "static" certain
    uint32_t stored_ok;
    uint16_t lit[288];            // ascending non-zero literal/length counts, m_l of them
    uint16_t dist[32];            // ascending non-zero distance counts, m_d of them
};
static_assert(sizeof(BlockProbe) == 32 + 640, "BlockProbe layout");

constexpr uint32_t kChosenOffset = 4096;   // bytes into a chunk's link array
constexpr uint32_t kChosenCap = 21846;     // matches per chunk: fewer than 65535 / 3 + 1
// The plan stage's merge lists (zwz_plan.hip: plan_heap -> plan) sit at the end of the same space: per block the literal/length
// tree's merges (<= 285 words), then the distance tree's (<= 29).
constexpr uint32_t kPairsOffset = 98304, kPairLitWords = 288, kPairWords = 320;
static_assert(kMaxBlocks * sizeof(BlockProbe) <= kChosenOffset && kChosenOffset + kChosenCap * 4 <= kPairsOffset &&
              kPairsOffset + kMaxBlocks * kPairWords * 4 <= kLinkStride * 2, "links space: probes, then chosen records, then merge lists");

struct DeflateArgs {
    const uint8_t* in; const uint64_t* in_off; const uint32_t* in_len; uint32_t n;   // chunk bases 16-byte aligned
    uint8_t* out; uint64_t out_stride; uint32_t* out_len;                            // out_stride % 4 == 0, >= 65536
    // workspace (sized for n chunks)
    uint16_t* links; uint2* entries; uint64_t* has128; uint64_t* sym; uint64_t* mst;
    uint16_t* perm;            // lz_match work order of the current tile, kTile entries per chunk
    uint32_t* link_stat;       // per chunk: positions with a chain predecessor (lz_links -> lz_match's choice of work order)
    uint32_t* tickets;         // kTicketBytes of counters (kTicket*; zeroed by launch_deflate)
    uint32_t* sorted;          // kSortedStride words per chunk: lz_sort's dest[p] (16 bits each), then lz_place's position of every sorted index
    uint32_t* dense_list;      // chunks that take the sort + band path (lz_dense_list), tickets[kTicketDenseCount] of them
    uint32_t* sparse_list;     // the others that have any bytes: lz_links' work, tickets[kTicketSparseCount] of them
    uint32_t cu_count;         // sizes the persistent grids (0: 256)
    uint32_t match_mode;       // kMatchAuto (production: lz_dense_list decides per chunk) | kMatchWalk | kMatchBand -- the context's option, same records either way
    uint32_t plan_serial;      // 1: the lane-serial block flush (plan_serial_kernel) instead of heap + wave
    ChunkInfo* info; BlockInfo* blocks; BlockOut* plans;
    BlockProbe* probes;        // = links (dead once lz_match has run): chunk c's kMaxBlocks probes open ITS link space
    // The chosen record of every match symbol, compact and in stream order (lz_parse -> blockify, encode): a chunk's
    // array also lives in its dead `links` space, kChosenOffset bytes in (behind the probes), < 21 846 entries.
};

struct InflateArgs {
    const uint8_t* in; const uint64_t* in_off; const uint32_t* in_len; uint32_t n;
    uint8_t* out; uint64_t out_stride; uint32_t* out_len; uint32_t* status;
    uint4* order;              // n entries of scratch (the launch fills it: (offset, length, chunk) by payload length, longest first), or null: as they come
    uint32_t serial_header;    // 1: block headers and tables by lane 0 alone (inflate_block_rest) -- no ordered LDS adds
};
// match_mode: auto = lz_dense_list decides per chunk between links + lz_match (+ lz_parse) and sort + lz_lazy; walk / band / lazy = every chunk through
// that search; autoband / autolazy = the per-chunk choice with the band + lz_parse / lz_lazy for the chain-heavy ones.  Same bytes whichever runs.
enum : uint32_t { kMatchAuto = 0, kMatchWalk = 1, kMatchBand = 2, kMatchLazy = 3, kMatchAutoBand = 4, kMatchAutoLazy = 5 };
#ifndef ZWZ_AUTO_LAZY
#define ZWZ_AUTO_LAZY 0
#endif
constexpr bool kAutoIsLazy = ZWZ_AUTO_LAZY != 0;      // what "auto" sends chain-heavy chunks through: lz_lazy, or the band + lz_parse
// Experiment defines this library was built with (all zero in the product; tools/gpu.sh times builds libzwz_hip_exp.so with one set)
uint32_t exp_flags_kernels();   // ZWZ_MATCH_EXP | ZWZ_PARSE_EXP << 8 | ZWZ_ENC_EXP << 16 | ZWZ_INF_EXP << 24
uint32_t exp_flags_band();      // ZWZ_BAND_EXP

constexpr size_t kTicketBytes = 256;
enum : uint32_t { kTicketHuffCount = 0, kTicketHuffNext = 1, kTicketDenseCount = 2, kTicketSortNext = 3, kTicketBandNext = 4, kTicketPlaceNext = 5, kTicketSparseCount = 6, kTicketOpenCount = 7, kTicketLazyNext = 8, kTicketSmallCount = 9, kTicketSmallNext = 10 };   // indices into DeflateArgs::tickets
// lz_match on its own (ZWZ_MATCH=walk): a chunk four of whose five positions have a chain predecessor (lz_links' count) takes the sorted walk.
__host__ __device__ inline bool chunk_is_dense(uint32_t linked, uint32_t L) { return linked * 5u >= L * 4u; }
// With the band kernels: lz_dense_list looks at a chunk's first kDenseSample positions and calls it chain-heavy -- sort + band --
// by sample_is_dense below (incompressible bytes: one position in thirty falls into a bucket taken already; the text corpus: two in
// five); its mark in link_stat[] tells lz_match which chunks are not its own.
constexpr uint32_t kDenseMark = 0xffffffffu;   // (kDenseSample and sample_is_dense: lz_band.h, portable -- tests/emu pins the rule)
constexpr size_t kWorkspaceBytesPerChunk =
    (size_t)kLinkStride * 2 + (size_t)kEntryStride * 8 + 3 * (size_t)kMaskWords * 8 + (size_t)kTile * 2 + 4 + sizeof(ChunkInfo) + (size_t)kSortedStride * 4 + 8 +
    kMaxBlocks * (sizeof(BlockInfo) + sizeof(BlockOut));

hipError_t configure_kernels();
hipError_t probe_exchange_order(hipStream_t s, bool* holds);   // see exchange_order_probe_kernel
hipError_t launch_deflate(const DeflateArgs& a, hipStream_t s, hipEvent_t* stage_events);
hipError_t launch_links_only(const DeflateArgs& a, hipStream_t s);
hipError_t configure_band_kernels();
hipError_t launch_dense_list(const DeflateArgs& a, hipStream_t s, uint32_t which);   // 0: by a sample of each chunk, 2: every chunk is chain-heavy
hipError_t launch_sort(const DeflateArgs& a, hipStream_t s);
hipError_t launch_place(const DeflateArgs& a, hipStream_t s);
hipError_t launch_match_band(const DeflateArgs& a, hipStream_t s);
hipError_t configure_lazy_kernels();                                                     // zwz_lazy.hip
hipError_t launch_lazy(const DeflateArgs& a, hipStream_t s);
uint32_t exp_flags_lazy();
hipError_t launch_plan(const DeflateArgs& a, hipStream_t s);                             // zwz_plan.hip
hipError_t launch_inflate(const InflateArgs& a, hipStream_t s);
hipError_t launch_md5_files(const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, const uint32_t* files, uint32_t n_files,
                            uint32_t* digests, hipStream_t s);

}  // namespace zwz
