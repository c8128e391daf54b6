/*
 * zwz_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic behind the reference's hot path:
 *   - compression.cpp:118-134   one-shot zlib deflate (level 6) of a <=65535 B chunk into a
 *                               65535 B buffer (silently truncated)
 *   - decompression.cpp:11-37   zlib inflate of one chunk payload, return codes ignored
 *   - verification.cpp:6-30     MD5 of a whole file as 32 lowercase hex chars
 *   - compression.cpp:24-104    chunking + .zwz record framing
 *   - decompression.cpp:45-163  .zwz record parsing + per-file reassembly
 *
 * The arithmetic itself lives in zlib (un-vendored system library, unpinned by the reference;
 * the container this oracle was pinned in has zlib1g 1:1.2.11.dfsg-2ubuntu9.2) and OpenSSL MD5
 * (RFC 1321).  Their sources are not in /root/reference, so this file restates the published
 * algorithms (RFC 1950/1951/1321 + zlib 1.2.11's level-6 heuristics).
 *
 * PARITY PINNING: the reference ships no tests or golden vectors.  This oracle is pinned by
 *   (1) tests/golden/ fixtures minted from the reference binary built from /root/reference
 *       (oracle/Makefile -> oracle/_ref/main) and from libz 1.2.11 (tests/golden/make_golden.py),
 *   (2) a live differential fuzz against the box's libz when it reports 1.2.11.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 */
#ifndef ZWZ_ORACLE_H
#define ZWZ_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZO_CHUNK_SIZE 65535u /* process.hpp:12 */

/* Full zlib stream (78 9c | deflate blocks | adler32 BE) of in[0..n), zlib 1.2.11 level 6,
 * windowBits 15, memLevel 8, default strategy, one deflate(Z_FINISH) call.  Any n.
 * Returns stream length, or 0 if cap is too small (a stream is never 0 bytes). */
size_t zo_deflate6(const uint8_t *in, size_t n, uint8_t *out, size_t cap);

/* Reference payload of one chunk (compression.cpp:127-132): first min(len,65535) stream bytes.
 * out must hold 65535 bytes.  n <= 65535. */
uint32_t zo_chunk_payload(const uint8_t *in, uint32_t n, uint8_t *out);

/* LZ77 symbol stream of the level-6 parse (debug/test aid for the GPU parse kernels).
 * dist[i]==0 -> literal lc[i]; else match of length lc[i]+3 at distance dist[i].
 * Arrays must hold n entries.  Returns the symbol count. */
size_t zo_lz77_symbols(const uint8_t *in, size_t n, uint16_t *dist, uint8_t *lc);

/* zlib inflate of one payload with the reference's semantics (decompression.cpp:24-36):
 * return codes ignored, output = every byte zlib would have produced before it stops
 * (stream end, input exhausted, or data error).  Writes at most cap bytes; returns the number
 * of bytes that WOULD be produced (so > cap signals overflow).  *status (optional):
 * 0 stream ended cleanly incl. adler32, 1 input exhausted, 2 data error, 3 adler mismatch. */
size_t zo_inflate(const uint8_t *in, size_t n, uint8_t *out, size_t cap, int *status);

uint32_t zo_adler32(const uint8_t *in, size_t n);

/* RFC 1321 MD5 -> 32 lowercase hex chars + NUL (verification.cpp:24-27). */
void zo_md5_hex(const uint8_t *in, size_t n, char hex[33]);

/* ---- container (.zwz) ------------------------------------------------------------------ */

/* Append the records of one file (all its chunks + trailing md5hex) to a growing buffer, exactly
 * as producer()+consumer()+data_writer() would (compression.cpp:52-64,73-104), including the
 * extra empty chunk when size % 65535 == 0.  (buf, len, cap) form a realloc-grown byte vector. */
int zo_append_file_records(const char *relpath, const uint8_t *data, size_t size,
                           uint8_t **buf, size_t *len, size_t *cap);

/* Whole-shard compress: files listed one relpath per line in record_file (sorted list,
 * file_sort.cpp:33-40); lines i with i % nranks == rank belong to this shard
 * (compression.cpp:38-41).  Writes <out_dir>/compressed_<rank>.zwz.  Returns 0 on success. */
int zo_compress_shard(const char *in_dir, const char *out_dir, const char *record_file, int rank,
                      int nranks);

/* Decode one shard file into out_dir (decompression.cpp:45-163).  Returns the number of files
 * whose MD5 did not match (>=0) or -1 on I/O error. */
int zo_decompress_shard(const char *shard_path, const char *out_dir);

#ifdef __cplusplus
}
#endif
#endif
