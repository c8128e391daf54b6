/*
 * zo_container.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restatement of the reference's per-rank pipeline around the codec:
 *   producer()      compression.cpp:24-71    rank-local file selection + 65535-byte chunking
 *   data_writer()   compression.cpp:73-104   .zwz record framing + md5hex after a file's last chunk
 *   do_compression  compression.cpp:151-170  output name compressed_<rank>.zwz
 *   decompress_zwz  decompression.cpp:45-163 record parsing, per-path reorder, MD5 verdict
 * Format: SURVEY.md Appendix A.
 */
#define _GNU_SOURCE
#include "zwz_oracle.h"
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

static int vec_reserve(uint8_t **buf, size_t *cap, size_t need) {
    if (need <= *cap) return 0;
    size_t nc = *cap ? *cap : 1 << 16;
    while (nc < need) nc *= 2;
    uint8_t *nb = (uint8_t *)realloc(*buf, nc);
    if (!nb) return -1;
    *buf = nb; *cap = nc;
    return 0;
}
static int vec_put(uint8_t **buf, size_t *len, size_t *cap, const void *p, size_t n) {
    if (vec_reserve(buf, cap, *len + n)) return -1;
    memcpy(*buf + *len, p, n);
    *len += n;
    return 0;
}

int zo_append_file_records(const char *relpath, const uint8_t *data, size_t size,
                           uint8_t **buf, size_t *len, size_t *cap) {
    int32_t path_len = (int32_t)strlen(relpath);
    uint8_t payload[ZO_CHUNK_SIZE];
    size_t off = 0;
    int32_t seq = 0;
    for (;;) {
        /* source.read(65535): a short read (incl. 0 bytes) is what sets eof (compression.cpp:52-58) */
        size_t take = size - off < ZO_CHUNK_SIZE ? size - off : ZO_CHUNK_SIZE;
        uint8_t last = take < ZO_CHUNK_SIZE;
        uint32_t plen = zo_chunk_payload(data + off, (uint32_t)take, payload);
        int32_t total = 4 + path_len + 4 + 1 + (int32_t)plen;
        if (vec_put(buf, len, cap, &total, 4) || vec_put(buf, len, cap, &path_len, 4) ||
            vec_put(buf, len, cap, relpath, (size_t)path_len) || vec_put(buf, len, cap, &seq, 4) ||
            vec_put(buf, len, cap, &last, 1) || vec_put(buf, len, cap, payload, plen))
            return -1;
        off += take; seq++;
        if (last) break;
    }
    char hex[33];
    zo_md5_hex(data, size, hex);
    return vec_put(buf, len, cap, hex, 32);
}

static uint8_t *slurp(const char *path, size_t *size) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *p = (uint8_t *)malloc(sz > 0 ? (size_t)sz : 1);
    if (p && sz > 0 && fread(p, 1, (size_t)sz, f) != (size_t)sz) { free(p); p = NULL; }
    fclose(f);
    *size = (size_t)sz;
    return p;
}

static int line_is_blank(const char *s, size_t n) {
    for (size_t i = 0; i < n; i++)
        if (!(s[i] == ' ' || (s[i] >= '\t' && s[i] <= '\r'))) return 0;
    return 1;
}

int zo_compress_shard(const char *in_dir, const char *out_dir, const char *record_file, int rank, int nranks) {
    size_t rsz;
    char *rec = (char *)slurp(record_file, &rsz);
    if (!rec) return -1;
    /* main.cpp:44-47: ranks >= number of non-blank lines create no shard */
    int nonblank = 0;
    for (size_t p = 0; p < rsz;) {
        size_t e = p;
        while (e < rsz && rec[e] != '\n') e++;
        if (!line_is_blank(rec + p, e - p)) nonblank++;
        p = e + 1;
    }
    if (rank >= nonblank) { free(rec); return 0; }

    char path[4096];
    snprintf(path, sizeof path, "%s/compressed_%d.zwz", out_dir, rank);
    FILE *dest = fopen(path, "wb");
    if (!dest) { free(rec); return -1; }
    uint8_t *buf = NULL; size_t len = 0, cap = 0;
    int line_no = 0, next = rank, rc = 0;
    for (size_t p = 0; p < rsz; line_no++) {
        size_t e = p;
        while (e < rsz && rec[e] != '\n') e++;
        if (line_no == next) {
            next += nranks;
            char rel[2048];
            size_t l = e - p < sizeof rel - 1 ? e - p : sizeof rel - 1;
            memcpy(rel, rec + p, l); rel[l] = 0;
            snprintf(path, sizeof path, "%s/%s", in_dir, rel);
            size_t fsz;
            uint8_t *data = slurp(path, &fsz);
            if (data) { /* open failure: reference logs and skips (compression.cpp:45-48) */
                len = 0;
                if (zo_append_file_records(rel, data, fsz, &buf, &len, &cap)) rc = -1;
                else if (fwrite(buf, 1, len, dest) != len) rc = -1;
                free(data);
            }
        }
        p = e + 1;
    }
    free(buf); free(rec);
    if (fclose(dest)) rc = -1;
    return rc;
}

/* ---------------------------------------------------------------- decompress */
typedef struct pending { int32_t seq; uint8_t last; uint8_t *payload; uint32_t plen; struct pending *next; } pending_t;
typedef struct ofile { char *rel; FILE *f; int32_t expected; pending_t *pend; char md5[33]; struct ofile *next; } ofile_t;

static void mkdirs_for(const char *file_path) {
    char tmp[4096];
    snprintf(tmp, sizeof tmp, "%s", file_path);
    for (char *p = tmp + 1; *p; p++)
        if (*p == '/') { *p = 0; mkdir(tmp, 0777); *p = '/'; }
}

static void inflate_to(FILE *f, const uint8_t *payload, uint32_t plen) {
    static uint8_t out[1 << 20];
    size_t n = zo_inflate(payload, plen, out, sizeof out, NULL);
    if (n > sizeof out) n = sizeof out;
    fwrite(out, 1, n, f);
}

int zo_decompress_shard(const char *shard_path, const char *out_dir) {
    size_t sz;
    uint8_t *sh = slurp(shard_path, &sz);
    if (!sh) return -1;
    ofile_t *open_files = NULL;
    int mismatches = 0;
    size_t p = 0;
    while (p + 4 <= sz) {
        int32_t total, path_len, seq;
        memcpy(&total, sh + p, 4); p += 4;
        if (p + 4 > sz) break;
        memcpy(&path_len, sh + p, 4); p += 4;
        if (path_len < 0 || p + (size_t)path_len + 5 > sz) break;
        char *rel = (char *)malloc((size_t)path_len + 1);
        memcpy(rel, sh + p, (size_t)path_len); rel[path_len] = 0; p += (size_t)path_len;
        memcpy(&seq, sh + p, 4); p += 4;
        uint8_t last = sh[p++];
        int32_t plen = total - (4 + path_len + 4 + 1);
        if (plen < 0 || plen > 65535) { free(rel); break; }
        /* A shard that ends inside a record (decompression.cpp:82-92): the reference reads the payload into a zero-filled
         * vector of the declared size and the MD5 into a zero-filled string, processes the record, then stops at eof. */
        int damaged = 0;
        uint8_t *padded = NULL;
        const uint8_t *payload = sh + p;
        if (p + (size_t)plen > sz) {
            damaged = 1;
            padded = (uint8_t *)calloc((size_t)plen + 1, 1);
            memcpy(padded, sh + p, sz - p);
            payload = padded; p = sz;
        } else p += (size_t)plen;
        char md5[33] = {0};
        if (last) {
            size_t have = sz - p < 32 ? sz - p : 32;
            memcpy(md5, sh + p, have); p += have;
            if (have < 32) damaged = 1;
        }

        ofile_t *of = open_files;
        while (of && strcmp(of->rel, rel)) of = of->next;
        char file_path[4096];
        snprintf(file_path, sizeof file_path, "%s/%s", out_dir, rel);
        if (!of) {
            mkdirs_for(file_path);
            FILE *f = fopen(file_path, "wb");
            if (!f) { free(rel); free(padded); if (damaged) break; continue; }
            of = (ofile_t *)calloc(1, sizeof *of);
            of->rel = strdup(rel); of->f = f; of->next = open_files; open_files = of;
        }
        if (last) memcpy(of->md5, md5, 33);
        if (of->expected == seq) {
            inflate_to(of->f, payload, (uint32_t)plen);
            of->expected++;
            for (;;) { /* drain in-order pending chunks (decompression.cpp:124-130) */
                pending_t **pp = &of->pend, *hit = NULL;
                for (; *pp; pp = &(*pp)->next) if ((*pp)->seq == of->expected) { hit = *pp; *pp = hit->next; break; }
                if (!hit) break;
                inflate_to(of->f, hit->payload, hit->plen);
                of->expected++;
                free(hit->payload); free(hit);
            }
            if (last && of->expected == seq + 1 && !of->pend) {
                fclose(of->f);
                size_t osz; uint8_t *od = slurp(file_path, &osz);
                char hex[33];
                zo_md5_hex(od ? od : (const uint8_t *)"", od ? osz : 0, hex);
                free(od);
                if (memcmp(hex, of->md5, 32)) mismatches++;
                ofile_t **q = &open_files;
                while (*q != of) q = &(*q)->next;
                *q = of->next;
                free(of->rel); free(of);
            }
        } else {
            pending_t *pd = (pending_t *)calloc(1, sizeof *pd);
            pd->seq = seq; pd->last = last; pd->plen = (uint32_t)plen;
            pd->payload = (uint8_t *)malloc((size_t)plen + 1);
            memcpy(pd->payload, payload, (size_t)plen);
            pd->next = of->pend; of->pend = pd;
        }
        free(rel);
        free(padded);
        if (damaged) break;
    }
    while (open_files) {
        ofile_t *of = open_files; open_files = of->next;
        fclose(of->f);
        while (of->pend) { pending_t *n = of->pend->next; free(of->pend->payload); free(of->pend); of->pend = n; }
        free(of->rel); free(of);
    }
    free(sh);
    return mismatches;
}
