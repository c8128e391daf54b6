/*
 * zo_inflate.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restatement of what decompress_chunk() (decompression.cpp:11-37) obtains from zlib's
 * inflateInit/inflate loop with every return code ignored: RFC 1950 header check, RFC 1951
 * block decoding, and -- for payloads the reference truncated at 65535 bytes -- "emit every
 * symbol whose bits are completely present, then stop silently" (Inflate).
 * The Adler-32 trailer is computed but never rejects (decompression.cpp:31).
 */
#include "zwz_oracle.h"
#include <string.h>

typedef struct {
    const uint8_t *in; size_t n, pos;
    uint64_t hold; int bits;
    uint8_t *out; size_t cap, produced;
} is_t;

/* Canonical Huffman decoder: counts per length + symbols sorted by (length, symbol). */
typedef struct { uint16_t count[16]; uint16_t sym[288]; int max_len; int nsyms; } huff_t;

/* Returns 0 ok, -1 over-subscribed, 1 incomplete. */
static int huff_build(huff_t *h, const uint8_t *lens, int n) {
    uint16_t offs[16];
    memset(h->count, 0, sizeof h->count);
    for (int i = 0; i < n; i++) h->count[lens[i]]++;
    h->nsyms = n - h->count[0];
    h->max_len = 0;
    for (int l = 1; l < 16; l++) if (h->count[l]) h->max_len = l;
    int left = 1;
    for (int l = 1; l < 16; l++) { left <<= 1; left -= h->count[l]; if (left < 0) return -1; }
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h->count[l]);
    for (int i = 0; i < n; i++) if (lens[i]) h->sym[offs[lens[i]]++] = (uint16_t)i;
    return left > 0 ? 1 : 0;
}

static int refill(is_t *s, int need) {
    while (s->bits < need) {
        if (s->pos >= s->n) return 0;
        s->hold |= (uint64_t)s->in[s->pos++] << s->bits;
        s->bits += 8;
    }
    return 1;
}
static unsigned take(is_t *s, int k) {
    unsigned v = (unsigned)(s->hold & ((1ull << k) - 1));
    s->hold >>= k; s->bits -= k;
    return v;
}

/* Decode one symbol bit by bit.  Returns symbol, -1 if input ran out before the code completed
 * (nothing consumed), -2 if the bits match no code (incomplete table). */
static int huff_decode(is_t *s, const huff_t *h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        if (!refill(s, len)) return -1;
        code |= (int)((s->hold >> (len - 1)) & 1);
        int count = h->count[len];
        if (code - count < first) { take(s, len); return h->sym[index + (code - first)]; }
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -2;
}

static void emit(is_t *s, unsigned b) {
    if (s->produced < s->cap) s->out[s->produced] = (uint8_t)b;
    s->produced++;
}

static const uint16_t len_base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t len_extra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t dist_base[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t dist_extra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};

enum { ST_END = 0, ST_NEED_INPUT = 1, ST_DATA_ERROR = 2, ST_ADLER = 3 };

/* Decode the symbols of one Huffman block.  Returns ST_END at end-of-block (continue), else stop code. */
static int codes(is_t *s, const huff_t *lh, const huff_t *dh, int lit_incomplete_ok) {
    (void)lit_incomplete_ok;
    for (;;) {
        /* A match is all-or-nothing: remember the bit position to know nothing else matters. */
        int sym = huff_decode(s, lh);
        if (sym == -1) return ST_NEED_INPUT;
        if (sym == -2) return ST_DATA_ERROR;
        if (sym < 256) { emit(s, (unsigned)sym); continue; }
        if (sym == 256) return ST_END;
        sym -= 257;
        if (sym >= 29) return ST_DATA_ERROR; /* invalid literal/length code */
        if (!refill(s, len_extra[sym])) return ST_NEED_INPUT;
        unsigned len = len_base[sym] + take(s, len_extra[sym]);
        int ds = huff_decode(s, dh);
        if (ds == -1) return ST_NEED_INPUT;
        if (ds == -2 || ds >= 30) return ST_DATA_ERROR; /* invalid distance code */
        if (!refill(s, dist_extra[ds])) return ST_NEED_INPUT;
        unsigned dist = dist_base[ds] + take(s, dist_extra[ds]);
        if (dist > s->produced) return ST_DATA_ERROR; /* invalid distance too far back */
        for (unsigned i = 0; i < len; i++) {
            size_t from = s->produced - dist;
            emit(s, from < s->cap ? s->out[from] : 0);
        }
    }
}

size_t zo_inflate(const uint8_t *in, size_t n, uint8_t *out, size_t cap, int *status) {
    is_t st = {in, n, 0, 0, 0, out, cap, 0};
    is_t *s = &st;
    int rc = ST_NEED_INPUT;
    huff_t lh, dh;
    uint8_t lens[320];

    if (!refill(s, 16)) goto done;
    {
        unsigned cmf = take(s, 8), flg = take(s, 8);
        rc = ST_DATA_ERROR;
        if (((cmf << 8) + flg) % 31) goto done;      /* incorrect header check */
        if ((cmf & 15) != 8) goto done;              /* unknown compression method */
        if ((cmf >> 4) + 8 > 15) goto done;          /* invalid window size */
        if (flg & 0x20) goto done;                   /* preset dictionary: zlib stops with Z_NEED_DICT */
    }
    for (;;) {
        rc = ST_NEED_INPUT;
        if (!refill(s, 3)) goto done;
        unsigned last = take(s, 1), type = take(s, 2);
        if (type == 0) {
            take(s, s->bits & 7);
            if (!refill(s, 32)) goto done;
            unsigned len = take(s, 16), nlen = take(s, 16);
            if ((len ^ 0xffff) != nlen) { rc = ST_DATA_ERROR; goto done; }
            /* bits is now 0: the rest of the block is whole bytes */
            while (len) {
                if (s->pos >= s->n) goto done;
                emit(s, s->in[s->pos++]); len--;
            }
        } else if (type == 1) {
            for (int i = 0; i < 288; i++) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            huff_build(&lh, lens, 288);
            for (int i = 0; i < 30; i++) lens[i] = 5;
            huff_build(&dh, lens, 30);
            rc = codes(s, &lh, &dh, 0);
            if (rc != ST_END) goto done;
        } else if (type == 2) {
            static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
            if (!refill(s, 14)) goto done;
            unsigned nlen = take(s, 5) + 257, ndist = take(s, 5) + 1, ncode = take(s, 4) + 4;
            if (nlen > 286 || ndist > 30) { rc = ST_DATA_ERROR; goto done; }
            uint8_t cl[19] = {0};
            for (unsigned i = 0; i < ncode; i++) { if (!refill(s, 3)) goto done; cl[order[i]] = (uint8_t)take(s, 3); }
            huff_t ch;
            if (huff_build(&ch, cl, 19) != 0) { rc = ST_DATA_ERROR; goto done; } /* must be complete */
            unsigned have = 0;
            while (have < nlen + ndist) {
                /* zlib peeks the code and its extra bits together before consuming either */
                is_t save = *s;
                int sym = huff_decode(s, &ch);
                if (sym == -1) goto done;
                if (sym == -2) { rc = ST_DATA_ERROR; goto done; }
                if (sym < 16) { lens[have++] = (uint8_t)sym; continue; }
                unsigned prev = 0, rep, xb = sym == 16 ? 2 : sym == 17 ? 3 : 7;
                if (!refill(s, (int)xb)) { *s = save; goto done; }
                if (sym == 16) {
                    if (have == 0) { rc = ST_DATA_ERROR; goto done; }
                    prev = lens[have - 1]; rep = 3 + take(s, 2);
                } else if (sym == 17) rep = 3 + take(s, 3);
                else rep = 11 + take(s, 7);
                if (have + rep > nlen + ndist) { rc = ST_DATA_ERROR; goto done; }
                while (rep--) lens[have++] = (uint8_t)prev;
            }
            rc = ST_DATA_ERROR;
            if (lens[256] == 0) goto done; /* missing end-of-block */
            int lr = huff_build(&lh, lens, (int)nlen);
            if (lr < 0 || (lr > 0 && lh.max_len != 1)) goto done; /* incomplete only if single 1-bit code */
            int dr = huff_build(&dh, lens + nlen, (int)ndist);
            if (dr < 0 || (dr > 0 && dh.max_len > 1)) goto done;
            rc = codes(s, &lh, &dh, 0);
            if (rc != ST_END) goto done;
        } else { rc = ST_DATA_ERROR; goto done; }
        if (last) break;
    }
    /* trailer */
    take(s, s->bits & 7);
    rc = ST_NEED_INPUT;
    if (!refill(s, 32)) goto done;
    {
        uint32_t b0 = take(s, 8), b1 = take(s, 8), b2 = take(s, 8), b3 = take(s, 8);
        uint32_t want = (b0 << 24) | (b1 << 16) | (b2 << 8) | b3;
        size_t k = st.produced < cap ? st.produced : cap;
        rc = (st.produced <= cap && zo_adler32(out, k) != want) ? ST_ADLER : ST_END;
    }
done:
    if (status) *status = rc;
    return st.produced;
}
