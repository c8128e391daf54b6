/*
 * zwz.h -- C ABI of the MI355X-native chunk codec: the drop-in boundary for the reference's hot path.
 *
 * The reference (JoernZheng/parallel-data-compression-and-decompression) has no plugin/FFI
 * surface; its seams are C++ prototypes in process.hpp:37-42 and two inline zlib call sites.
 * Each entry point below names the reference interface it replaces:
 *
 *   zwz_deflate_batch*     the zlib call pair in consumer(), compression.cpp:119-134
 *                          (deflateInit level 6 / deflate(Z_FINISH) into a 65535-byte buffer /
 *                          deflateEnd), once per Chunk (process.hpp:21-28) -- here for a batch
 *   zwz_inflate_batch*     decompress_chunk(), decompression.cpp:11-37 (inflateInit / inflate
 *                          loop / inflateEnd, return codes ignored), once per CompressedChunk
 *                          (process.hpp:30-35) -- here for a batch
 *   zwz_compress_dir       do_compression(input_dir, output_dir, file_record, world_rank),
 *                          process.hpp:39 / compression.cpp:161-194 (+ the rank < file_count
 *                          guard of compress(), main.cpp:44-51)
 *   zwz_decompress_dir     do_decompression(input_dir, output_dir), process.hpp:40 /
 *                          decompression.cpp:165-178
 *   zwz_sort_files_by_size sort_files_by_size(path), process.hpp:37 / file_sort.cpp:24-43
 *   zwz_count_non_empty_lines  count_non_empty_lines(file), process.hpp:38 / file_tools.cpp:6-23
 *   zwz_md5_of_file        md5_of_file(path), process.hpp:41 / verification.cpp:6-30
 *   zwz_md5_files_dev      the same digest for many files at once, from their chunks in device memory
 *                          (the call sites compression.cpp:98 and decompression.cpp:136 batched)
 *
 * Conventions: plain pointers and sizes, no exceptions across the boundary, 0 = success and
 * negative zwz_status codes otherwise, caller owns every buffer.  A context binds one GPU, one
 * HIP stream and a device workspace; calls on one context are serialised by the caller, distinct
 * contexts are independent.  There is no CPU fallback: without a usable GPU zwz_ctx_create fails.
 *
 * Bit-exactness contract: for every chunk, out[0..out_len) equals the first min(len, 65535)
 * bytes of zlib 1.2.11's level-6 stream of that chunk -- what the reference stores in a .zwz
 * record.
 */
#ifndef ZWZ_H
#define ZWZ_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZWZ_CHUNK_SIZE 65535u      /* process.hpp:12 CHUNK_SIZE */
#define ZWZ_DEV_STRIDE 65536u      /* chunk slot stride in device buffers (16-byte aligned slots) */
#define ZWZ_MD5_HEX_LEN 32u        /* process.hpp:14 MD5_DATA_SIZE */
#define ZWZ_LOSSLESS_CHUNK_SIZE 65509u /* opt-in: largest chunk whose stream always fits 65535 bytes: 4 stored blocks, n + 26 */

typedef enum zwz_status {
    ZWZ_OK = 0,
    ZWZ_E_INVALID = -1,   /* bad argument (null pointer, misaligned device slot, size > 65535) */
    ZWZ_E_HIP = -2,       /* a HIP runtime call failed; zwz_last_error() has the text */
    ZWZ_E_NO_DEVICE = -3, /* no usable gfx950 device (none visible, or it fails every one of zwz_ctx_create's self-tests: see zwz_ctx_set_option) */
    ZWZ_E_IO = -4,        /* file system error */
    ZWZ_E_NOMEM = -5,
    ZWZ_E_FORMAT = -6     /* malformed .zwz shard */
} zwz_status;

/* Per-chunk status written by the inflate entry points (the reference ignores zlib's return
 * codes, decompression.cpp:31; these only report, they never suppress output). */
typedef enum zwz_inflate_status {
    ZWZ_INF_END = 0,         /* final block reached */
    ZWZ_INF_NEED_INPUT = 1,  /* payload ended early (reference-truncated chunk): partial output kept */
    ZWZ_INF_DATA_ERROR = 2,  /* invalid stream: output up to the error kept */
    ZWZ_INF_OVERFLOW = 3     /* stream decodes past 65535 bytes (not producible by the reference) */
} zwz_inflate_status;

typedef struct zwz_ctx zwz_ctx;

const char *zwz_strerror(int status);
const char *zwz_last_error(void);               /* thread-local detail for ZWZ_E_HIP / ZWZ_E_IO */
int zwz_device_count(int *count);

/* max_batch_chunks bounds the device workspace (~680 KiB per chunk); larger batches are
 * processed in slices.  0 selects the default (8192). */
int zwz_ctx_create(int device, uint32_t max_batch_chunks, zwz_ctx **ctx);
void zwz_ctx_destroy(zwz_ctx *ctx);
void *zwz_ctx_stream(zwz_ctx *ctx);             /* the context's hipStream_t */
int zwz_ctx_sync(zwz_ctx *ctx);

/* ---- device-resident batches (asynchronous on the context's stream) ------------------------
 * d_in + d_in_off[i] is chunk i: d_in_len[i] <= 65535 bytes, d_in and every d_in_off[i] multiples
 * of 16, and the slot readable up to its length rounded up to 16 (the kernels stream whole 16-byte
 * vectors; the extra bytes never influence a result).  Chunk i's result goes to
 * d_out + i * out_stride (out_stride % 16 == 0, >= 65536) with its length in d_out_len[i].
 * All pointers are device memory on the context's GPU. */
int zwz_deflate_batch_dev(zwz_ctx *ctx, const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                          uint32_t n, uint8_t *d_out, uint64_t out_stride, uint32_t *d_out_len);
int zwz_inflate_batch_dev(zwz_ctx *ctx, const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                          uint32_t n, uint8_t *d_out, uint64_t out_stride, uint32_t *d_out_len, uint32_t *d_status);

/* ---- host-buffer batches (synchronous; staged through pinned memory) ------------------------
 * in + in_off[i] is chunk i; results at out + i * 65535, lengths in out_len[i]. */
int zwz_deflate_batch(zwz_ctx *ctx, const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint32_t n,
                      uint8_t *out, uint32_t *out_len);
int zwz_inflate_batch(zwz_ctx *ctx, const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint32_t n,
                      uint8_t *out, uint32_t *out_len, uint32_t *status);

/* ---- stage timing (HIP events on the context's stream) -------------------------------------
 * With profiling on, every deflate slice records events around its 6 stages; the accumulated
 * milliseconds since the last reset are returned in ms[0..6) in pipeline order -- links (marks of the
 * chain-heavy chunks, chain links of the others, the sorted arrays: lz_dense_list, lz_lists, lz_links,
 * lz_sort, lz_place), match (lz_match, lz_match_band), parse, blockify, plan (three kernels), encode
 * (two kernels) --, ms[6] = inflate (its launch order + the kernel). */
#define ZWZ_NUM_STAGES 7
int zwz_ctx_set_profiling(zwz_ctx *ctx, int on);
int zwz_ctx_stage_ms(zwz_ctx *ctx, float *ms, int reset);

/* MD5 (RFC 1321) of n_files files whose bytes already sit in device chunk slots: file i is the concatenation
 * of slots d_files[2i] .. d_files[2i] + d_files[2i+1] - 1 (slot k = d_in_len[k] bytes at d_in + d_in_off[k]).
 * d_digests receives 16 bytes per file (4-byte aligned).  Asynchronous on the context's stream.  One lane per
 * file: meant for many small and medium files; hash very large files on the host (zwz_md5_of_file). */
int zwz_md5_files_dev(zwz_ctx *ctx, const uint8_t *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                      const uint32_t *d_files, uint32_t n_files, uint8_t *d_digests);

/* ---- directory level (the reference's per-rank pipeline) ------------------------------------ */
int zwz_sort_files_by_size(const char *src_dir, char *record_path_out, size_t cap);
int zwz_count_non_empty_lines(const char *file_path);
int zwz_md5_of_file(const char *path, char hex_out[33]);
/* Shard `rank` of `nranks`: lines i of file_record with i % nranks == rank, written to
 * <dst>/compressed_<rank>.zwz.  Ranks >= the number of listed files write nothing. */
int zwz_compress_dir(zwz_ctx *ctx, const char *src_dir, const char *dst_dir, const char *file_record, int rank,
                     int nranks);
/* Every <src>/ *.zwz -> files under <dst>; md5_mismatches (optional) counts files whose MD5
 * differs from the stored one (the reference only prints them, decompression.cpp:140-146).
 * A shard that ends inside a record is decoded up to the damage (as the reference's reader would) and the
 * call returns ZWZ_E_FORMAT. */
int zwz_decompress_dir(zwz_ctx *ctx, const char *src_dir, const char *dst_dir, int *md5_mismatches);

/* The same job shared by nranks processes, one GPU each (the reference parallelises over shards only,
 * decompression.cpp:165-178, and decodes inside a shard serially, :65-154).  Shards are taken in name order: with at
 * least nranks of them, shard j belongs to rank j % nranks and no rank talks to another.  With fewer (BASELINE config 5:
 * one shard holding one huge file) every shard is split: rank r inflates the r-th contiguous range of its records, and
 * the ranks all-gather how many bytes each decoded (a chunk's place in its file is the sum of the decoded lengths in
 * front of it).  `exchange` is that all-gather: every rank calls it with `count` values in `mine` and receives
 * nranks * count values, rank-major, in `all`; it returns 0 on success.  It doubles as the barrier before MD5
 * verification, is called the same number of times on every rank, and may be NULL when nranks == 1 or there is a shard
 * per rank.  The launcher supplies it: torch.distributed / RCCL (cli.py), marker files (csrc/main.cpp).  <dst> must be
 * one file system for all ranks.  md5_mismatches counts the files THIS rank verified. */
typedef int (*zwz_allgather_u64_fn)(void *user, const uint64_t *mine, uint64_t *all, uint32_t count);
int zwz_decompress_dir_ranked(zwz_ctx *ctx, const char *src_dir, const char *dst_dir, int rank, int nranks,
                              zwz_allgather_u64_fn exchange, void *user, int *md5_mismatches);

/* "A final gather of per-shard .zwz blobs" (north star; the reference has none: every MPI rank writes compressed_<rank>.zwz itself,
 * compression.cpp:151-170).  The protocol, stated once for every launcher (csrc/main.cpp: RCCL; cli.py: torch.distributed): sizes and
 * readiness by all-gather, then every rank's shard travels to rank 0 in pieces of at most piece_bytes (0: 64 MiB), one send for one
 * receive; an I/O failure on either side marks a shard bad without leaving any send unmatched.  Rank 0 writes
 * <out_dir>/compressed_<r>.zwz (through a .part name).  my_shard_path: this rank's shard, "" or NULL = nothing to contribute (rank 0, an
 * idle or failed rank).  Returns 1 only if every rank saw every transfer and every write succeed -- only then may a sender delete its
 * copy.  The hooks move HOST memory; all of them return 0 on success; a failing send / recv / all-gather means the transport is gone.
 * prepare / release (optional): the transport's own staging for pieces of that size.  No GPU call is made by the library here. */
typedef struct zwz_gather_hooks {
    void *user;
    int (*allgather_u64)(void *user, const uint64_t *mine, uint64_t *all, uint32_t count);
    int (*send)(void *user, const void *buf, uint64_t nbytes, int to_rank);
    int (*recv)(void *user, void *buf, uint64_t nbytes, int from_rank);
    int (*prepare)(void *user, uint64_t piece_bytes);
    void (*release)(void *user);
} zwz_gather_hooks;
int zwz_gather_shards(int rank, int nranks, const char *my_shard_path, const char *out_dir, uint64_t piece_bytes,
                      const zwz_gather_hooks *hooks);

/* Opt-in, never the default, NOT bit-exact with the reference's shards: raw bytes per Chunk for
 * zwz_compress_dir, 1..65535; 0 restores the reference's 65535 (process.hpp:12).  ZWZ_LOSSLESS_CHUNK_SIZE (65509) is the
 * largest size whose level-6 stream always fits the reference's 65535-byte payload buffer (compression.cpp:127-132), so
 * nothing is truncated and every file round-trips with a matching MD5; the container is unchanged and the reference's
 * decoder reads such shards.  The environment variables ZWZ_LOSSLESS=1 / ZWZ_CHUNK_SIZE=<n> do the same for the CLI. */
int zwz_ctx_set_chunk_size(zwz_ctx *ctx, uint32_t bytes);

/* Test and diagnosis switches of one context; no reference counterpart (the reference's zlib has one code path,
 * compression.cpp:119-134 / decompression.cpp:16-36) and no effect on any byte produced -- they choose between kernels that compute
 * the same thing, so that tests can drive each of them and a device that fails a self-test at zwz_ctx_create still gets a codec:
 *   "match"           "auto" (default: per chunk, by a sample of its trigrams) | "walk" (chain links + walk) | "band" (sort + banded search)
 *                     | "lazy" (sort + lz_lazy: search and lazy parse in one kernel, the searches on demand) | "autoband" / "autolazy"
 *                     (the per-chunk choice with the band / lz_lazy for chain-heavy chunks; "auto" is "autoband")
 *   "plan"            "wave" (default) | "serial"   block flush: a lane per heap + a wave per block, or all of it on one lane
 *   "inflate_header"  "wave" (default) | "serial"   a block's decoding tables by the whole wave, or by lane 0
 * Defaults come from ZWZ_MATCH / ZWZ_PLAN / ZWZ_INFLATE_HEADER, read once in zwz_ctx_create (never per launch; a value that is not
 * understood is reported on stderr and ignored).  ZWZ_E_INVALID for an unknown name or value; ZWZ_E_NO_DEVICE for a kernel form
 * that failed its self-test on this device at zwz_ctx_create -- it stays off ("auto" / "" then mean what the device can run). */
int zwz_ctx_set_option(zwz_ctx *ctx, const char *name, const char *value);

#ifdef __cplusplus
}
#endif
#endif
