/*
 * zo_deflate.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restatement of what `deflateInit(&strm, Z_DEFAULT_COMPRESSION); deflate(&strm, Z_FINISH)`
 * (compression.cpp:119-131) computes with zlib 1.2.11: level 6 = lazy matching with
 * good_length 8, max_lazy 16, nice_length 128, max_chain 128; 32 KiB window in a 64 KiB buffer
 * that slides once strstart >= 65274; blocks flushed every 16383 symbols; heap-built Huffman
 * trees with zlib's tie-break; stored / static / dynamic choice; zlib wrapper.
 * Contract: SURVEY.md Appendix B.  Written from the published algorithm, structure is this
 * repo's own (whole input visible up front, growable bit sink).
 */
#include "zwz_oracle.h"
#include <stdlib.h>
#include <string.h>

enum {
    W_BITS = 15, W_SIZE = 1 << W_BITS, W_MASK = W_SIZE - 1, WINDOW_BYTES = 2 * W_SIZE,
    HASH_BITS = 15, HASH_SIZE = 1 << HASH_BITS, HASH_MASK = HASH_SIZE - 1, HASH_SHIFT = 5,
    MIN_MATCH = 3, MAX_MATCH = 258, MIN_LOOKAHEAD = MAX_MATCH + MIN_MATCH + 1,
    MAX_DIST = W_SIZE - MIN_LOOKAHEAD, TOO_FAR = 4096,
    GOOD_LENGTH = 8, MAX_LAZY = 16, NICE_LENGTH = 128, MAX_CHAIN = 128,
    LIT_BUFSIZE = 16384,
    L_CODES = 286, D_CODES = 30, BL_CODES = 19, HEAP_SIZE = 2 * L_CODES + 1, END_BLOCK = 256,
    MAX_BITS = 15, MAX_BL_BITS = 7, REP_3_6 = 16, REPZ_3_10 = 17, REPZ_11_138 = 18
};

static const uint8_t extra_lbits[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint8_t extra_dbits[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const uint8_t extra_blbits[19] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,3,7};
static const uint8_t bl_order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

/* RFC 1951 code tables, derived once. */
static uint8_t length_code[256];  /* len-3 -> 0..28 */
static uint16_t base_length[29];
static uint8_t dist_code_lo[256]; /* dist-1 < 256 */
static uint8_t dist_code_hi[256]; /* (dist-1) >> 7 */
static uint16_t base_dist[30];
static uint8_t static_llen[288];
static uint16_t static_lcode[288];
static uint16_t static_dcode[30];
static int tables_ready;

static unsigned bit_reverse(unsigned v, int len) {
    unsigned r = 0;
    while (len-- > 0) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

static void init_tables(void) {
    if (tables_ready) return;
    int len = 0;
    for (int c = 0; c < 28; c++) {
        base_length[c] = (uint16_t)len;
        for (int k = 0; k < (1 << extra_lbits[c]); k++) length_code[len++] = (uint8_t)c;
    }
    length_code[255] = 28; /* length 258 uses code 285, no extra bits */
    base_length[28] = 0;
    int d = 0;
    for (int c = 0; c < 16; c++) {
        base_dist[c] = (uint16_t)d;
        for (int k = 0; k < (1 << extra_dbits[c]); k++) dist_code_lo[d++] = (uint8_t)c;
    }
    d >>= 7;
    for (int c = 16; c < 30; c++) {
        base_dist[c] = (uint16_t)(d << 7);
        for (int k = 0; k < (1 << (extra_dbits[c] - 7)); k++) dist_code_hi[d++] = (uint8_t)c;
    }
    for (int n = 0; n < 288; n++) static_llen[n] = n < 144 ? 8 : n < 256 ? 9 : n < 280 ? 7 : 8;
    unsigned next[10] = {0};
    unsigned cnt[10] = {0};
    for (int n = 0; n < 288; n++) cnt[static_llen[n]]++;
    unsigned code = 0;
    for (int b = 1; b <= 9; b++) { code = (code + cnt[b - 1]) << 1; next[b] = code; }
    for (int n = 0; n < 288; n++) static_lcode[n] = (uint16_t)bit_reverse(next[static_llen[n]]++, static_llen[n]);
    for (int n = 0; n < 30; n++) static_dcode[n] = (uint16_t)bit_reverse((unsigned)n, 5);
    tables_ready = 1;
}

static inline int d_code(unsigned dm1) { return dm1 < 256 ? dist_code_lo[dm1] : dist_code_hi[dm1 >> 7]; }

typedef struct { uint16_t freq[HEAP_SIZE]; uint16_t len[HEAP_SIZE]; uint16_t code[HEAP_SIZE]; uint16_t dad[HEAP_SIZE]; } tree_t;

typedef struct {
    /* input */
    const uint8_t *src; size_t src_len, src_pos;
    /* window + hash chains */
    uint8_t window[WINDOW_BYTES + 8];
    uint16_t head[HASH_SIZE];
    uint16_t prev[W_SIZE];
    unsigned ins_h, strstart, lookahead, match_start, match_length, prev_match, prev_length;
    int match_available;
    long block_start;
    /* symbol buffer of the current block */
    uint16_t d_buf[LIT_BUFSIZE];
    uint8_t l_buf[LIT_BUFSIZE];
    unsigned last_lit;
    /* trees */
    tree_t lt, dt, bt;
    int l_max_code, d_max_code;
    uint16_t bl_count[MAX_BITS + 1];
    int heap[HEAP_SIZE]; int heap_len, heap_max; uint8_t depth[HEAP_SIZE];
    unsigned long opt_len, static_len;
    /* bit sink */
    uint8_t *out; size_t out_len, out_cap; int overflow;
    uint32_t bi_buf; int bi_valid;
    /* optional symbol trace */
    uint16_t *trace_dist; uint8_t *trace_lc; size_t trace_n;
} ds_t;

/* ------------------------------------------------------------------ bit sink */
static void put_byte(ds_t *s, unsigned b) {
    if (s->out_len < s->out_cap) s->out[s->out_len] = (uint8_t)b; else s->overflow = 1;
    s->out_len++;
}
static void send_bits(ds_t *s, unsigned value, int length) {
    s->bi_buf |= (uint32_t)value << s->bi_valid;
    s->bi_valid += length;
    while (s->bi_valid >= 8) { put_byte(s, s->bi_buf & 0xff); s->bi_buf >>= 8; s->bi_valid -= 8; }
}
static void bi_windup(ds_t *s) {
    if (s->bi_valid > 0) put_byte(s, s->bi_buf & 0xff);
    s->bi_buf = 0; s->bi_valid = 0;
}

/* ------------------------------------------------------------------ trees */
static void init_block(ds_t *s) {
    memset(s->lt.freq, 0, sizeof s->lt.freq);
    memset(s->dt.freq, 0, sizeof s->dt.freq);
    memset(s->bt.freq, 0, sizeof s->bt.freq);
    s->lt.freq[END_BLOCK] = 1;
    s->opt_len = s->static_len = 0;
    s->last_lit = 0;
}

#define SMALLER(t, n, m) ((t)->freq[n] < (t)->freq[m] || ((t)->freq[n] == (t)->freq[m] && s->depth[n] <= s->depth[m]))

static void sift_down(ds_t *s, tree_t *t, int k) {
    int v = s->heap[k], j = k << 1;
    while (j <= s->heap_len) {
        if (j < s->heap_len && SMALLER(t, s->heap[j + 1], s->heap[j])) j++;
        if (SMALLER(t, v, s->heap[j])) break;
        s->heap[k] = s->heap[j]; k = j; j <<= 1;
    }
    s->heap[k] = v;
}

static void gen_bitlen(ds_t *s, tree_t *t, int max_code, const uint8_t *extra, int base, int max_length,
                       const uint8_t *stat_len) {
    int h, overflow = 0;
    for (int b = 0; b <= MAX_BITS; b++) s->bl_count[b] = 0;
    t->len[s->heap[s->heap_max]] = 0; /* root */
    for (h = s->heap_max + 1; h < HEAP_SIZE; h++) {
        int n = s->heap[h];
        int bits = t->len[t->dad[n]] + 1;
        if (bits > max_length) { bits = max_length; overflow++; }
        t->len[n] = (uint16_t)bits;
        if (n > max_code) continue; /* internal node */
        s->bl_count[bits]++;
        int xbits = n >= base ? extra[n - base] : 0;
        unsigned long f = t->freq[n];
        s->opt_len += f * (unsigned)(bits + xbits);
        if (stat_len) s->static_len += f * (unsigned)(stat_len[n] + xbits);
    }
    if (overflow == 0) return;
    do {
        int bits = max_length - 1;
        while (s->bl_count[bits] == 0) bits--;
        s->bl_count[bits]--;
        s->bl_count[bits + 1] += 2;
        s->bl_count[max_length]--;
        overflow -= 2;
    } while (overflow > 0);
    for (int bits = max_length; bits != 0; bits--) {
        int n = s->bl_count[bits];
        while (n != 0) {
            int m = s->heap[--h];
            if (m > max_code) continue;
            if (t->len[m] != (unsigned)bits) {
                s->opt_len += ((unsigned long)bits - t->len[m]) * t->freq[m];
                t->len[m] = (uint16_t)bits;
            }
            n--;
        }
    }
}

static void gen_codes(ds_t *s, tree_t *t, int max_code) {
    unsigned next_code[MAX_BITS + 1], code = 0;
    for (int b = 1; b <= MAX_BITS; b++) { code = (code + s->bl_count[b - 1]) << 1; next_code[b] = code; }
    for (int n = 0; n <= max_code; n++) {
        int len = t->len[n];
        if (len) t->code[n] = (uint16_t)bit_reverse(next_code[len]++, len);
    }
}

/* static_len_of(sym) for lit (288 entries), 5 for dist, none for bl */
static int build_tree(ds_t *s, tree_t *t, int elems, const uint8_t *extra, int base, int max_length,
                      const uint8_t *stat_len) {
    int n, m, max_code = -1, node;
    s->heap_len = 0; s->heap_max = HEAP_SIZE;
    for (n = 0; n < elems; n++) {
        if (t->freq[n] != 0) { s->heap[++s->heap_len] = max_code = n; s->depth[n] = 0; }
        else t->len[n] = 0;
    }
    while (s->heap_len < 2) {
        node = s->heap[++s->heap_len] = (max_code < 2 ? ++max_code : 0);
        t->freq[node] = 1; s->depth[node] = 0; s->opt_len--;
        if (stat_len) s->static_len -= stat_len[node];
    }
    for (n = s->heap_len / 2; n >= 1; n--) sift_down(s, t, n);
    node = elems;
    do {
        n = s->heap[1]; s->heap[1] = s->heap[s->heap_len--]; sift_down(s, t, 1);
        m = s->heap[1];
        s->heap[--s->heap_max] = n; s->heap[--s->heap_max] = m;
        t->freq[node] = (uint16_t)(t->freq[n] + t->freq[m]);
        s->depth[node] = (uint8_t)((s->depth[n] >= s->depth[m] ? s->depth[n] : s->depth[m]) + 1);
        t->dad[n] = t->dad[m] = (uint16_t)node;
        s->heap[1] = node++;
        sift_down(s, t, 1);
    } while (s->heap_len >= 2);
    s->heap[--s->heap_max] = s->heap[1];
    gen_bitlen(s, t, max_code, extra, base, max_length, stat_len);
    gen_codes(s, t, max_code);
    return max_code;
}

/* One pass over a code-length array; emit!=0 sends the RLE, emit==0 tallies bl frequencies. */
static void rle_tree(ds_t *s, const tree_t *t, int max_code, int emit) {
    int prevlen = -1, curlen, nextlen = t->len[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    for (int n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = n == max_code ? 0xffff : t->len[n + 1];
        if (++count < max_count && curlen == nextlen) continue;
        if (count < min_count) {
            if (emit) { do { send_bits(s, s->bt.code[curlen], s->bt.len[curlen]); } while (--count != 0); }
            else s->bt.freq[curlen] = (uint16_t)(s->bt.freq[curlen] + count);
        } else if (curlen != 0) {
            if (curlen != prevlen) {
                if (emit) { send_bits(s, s->bt.code[curlen], s->bt.len[curlen]); count--; }
                else s->bt.freq[curlen]++;
            }
            if (emit) { send_bits(s, s->bt.code[REP_3_6], s->bt.len[REP_3_6]); send_bits(s, (unsigned)count - 3, 2); }
            else s->bt.freq[REP_3_6]++;
        } else if (count <= 10) {
            if (emit) { send_bits(s, s->bt.code[REPZ_3_10], s->bt.len[REPZ_3_10]); send_bits(s, (unsigned)count - 3, 3); }
            else s->bt.freq[REPZ_3_10]++;
        } else {
            if (emit) { send_bits(s, s->bt.code[REPZ_11_138], s->bt.len[REPZ_11_138]); send_bits(s, (unsigned)count - 11, 7); }
            else s->bt.freq[REPZ_11_138]++;
        }
        count = 0; prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
}

static void compress_block(ds_t *s, const uint16_t *lcode, const uint8_t *llen8, const uint16_t *llen16,
                           const uint16_t *dcode, const uint16_t *dlen16, int dlen_const) {
#define LLEN(c) (llen8 ? llen8[c] : llen16[c])
#define DLEN(c) (dlen16 ? dlen16[c] : dlen_const)
    for (unsigned lx = 0; lx < s->last_lit; lx++) {
        unsigned dist = s->d_buf[lx], lc = s->l_buf[lx];
        if (dist == 0) { send_bits(s, lcode[lc], LLEN(lc)); continue; }
        unsigned code = length_code[lc];
        send_bits(s, lcode[code + 257], LLEN(code + 257));
        if (extra_lbits[code]) send_bits(s, lc - base_length[code], extra_lbits[code]);
        dist--;
        code = (unsigned)d_code(dist);
        send_bits(s, dcode[code], DLEN(code));
        if (extra_dbits[code]) send_bits(s, dist - base_dist[code], extra_dbits[code]);
    }
    send_bits(s, lcode[END_BLOCK], LLEN(END_BLOCK));
#undef LLEN
#undef DLEN
}

static void flush_block(ds_t *s, int last) {
    long bs = s->block_start;
    const uint8_t *buf = bs >= 0 ? &s->window[bs] : NULL;
    unsigned long stored_len = (unsigned long)((long)s->strstart - bs);

    s->l_max_code = build_tree(s, &s->lt, L_CODES, extra_lbits, 257, MAX_BITS, static_llen);
    static const uint8_t five[30] = {5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5};
    s->d_max_code = build_tree(s, &s->dt, D_CODES, extra_dbits, 0, MAX_BITS, five);
    rle_tree(s, &s->lt, s->l_max_code, 0);
    rle_tree(s, &s->dt, s->d_max_code, 0);
    build_tree(s, &s->bt, BL_CODES, extra_blbits, 0, MAX_BL_BITS, NULL);
    int max_blindex;
    for (max_blindex = BL_CODES - 1; max_blindex >= 3; max_blindex--)
        if (s->bt.len[bl_order[max_blindex]] != 0) break;
    s->opt_len += 3 * ((unsigned long)max_blindex + 1) + 5 + 5 + 4;

    unsigned long opt_lenb = (s->opt_len + 3 + 7) >> 3, static_lenb = (s->static_len + 3 + 7) >> 3;
    if (static_lenb <= opt_lenb) opt_lenb = static_lenb;

    if (stored_len + 4 <= opt_lenb && buf != NULL) {
        send_bits(s, (0u << 1) + (unsigned)last, 3);
        bi_windup(s);
        put_byte(s, stored_len & 0xff); put_byte(s, (stored_len >> 8) & 0xff);
        put_byte(s, ~stored_len & 0xff); put_byte(s, (~stored_len >> 8) & 0xff);
        for (unsigned long i = 0; i < stored_len; i++) put_byte(s, buf[i]);
    } else if (static_lenb == opt_lenb) {
        send_bits(s, (1u << 1) + (unsigned)last, 3);
        compress_block(s, static_lcode, static_llen, NULL, static_dcode, NULL, 5);
    } else {
        send_bits(s, (2u << 1) + (unsigned)last, 3);
        send_bits(s, (unsigned)(s->l_max_code + 1 - 257), 5);
        send_bits(s, (unsigned)(s->d_max_code + 1 - 1), 5);
        send_bits(s, (unsigned)(max_blindex + 1 - 4), 4);
        for (int r = 0; r <= max_blindex; r++) send_bits(s, s->bt.len[bl_order[r]], 3);
        rle_tree(s, &s->lt, s->l_max_code, 1);
        rle_tree(s, &s->dt, s->d_max_code, 1);
        compress_block(s, s->lt.code, NULL, s->lt.len, s->dt.code, s->dt.len, 0);
    }
    init_block(s);
    if (last) bi_windup(s);
    s->block_start = (long)s->strstart;
}

/* ------------------------------------------------------------------ LZ77 */
#define UPDATE_HASH(h, c) ((h) = (((h) << HASH_SHIFT) ^ (c)) & HASH_MASK)

static unsigned insert_string(ds_t *s, unsigned str) {
    UPDATE_HASH(s->ins_h, s->window[str + MIN_MATCH - 1]);
    unsigned hh = s->prev[str & W_MASK] = s->head[s->ins_h];
    s->head[s->ins_h] = (uint16_t)str;
    return hh;
}

static void fill_window(ds_t *s) {
    do {
        unsigned more = (unsigned)(WINDOW_BYTES - s->lookahead - s->strstart);
        if (s->strstart >= W_SIZE + MAX_DIST) {
            memmove(s->window, s->window + W_SIZE, W_SIZE - more);
            s->match_start -= W_SIZE; s->strstart -= W_SIZE; s->block_start -= W_SIZE;
            for (int i = 0; i < HASH_SIZE; i++) s->head[i] = (uint16_t)(s->head[i] >= W_SIZE ? s->head[i] - W_SIZE : 0);
            for (int i = 0; i < W_SIZE; i++) s->prev[i] = (uint16_t)(s->prev[i] >= W_SIZE ? s->prev[i] - W_SIZE : 0);
            more += W_SIZE;
        }
        if (s->src_pos == s->src_len) break;
        size_t n = s->src_len - s->src_pos;
        if (n > more) n = more;
        memcpy(s->window + s->strstart + s->lookahead, s->src + s->src_pos, n);
        s->src_pos += n;
        /* bytes beyond the data must compare like zlib's zero-initialised window tail */
        s->lookahead += (unsigned)n;
        if (s->lookahead >= MIN_MATCH && s->strstart == 0 && s->src_pos == n) {
            s->ins_h = s->window[0];
            UPDATE_HASH(s->ins_h, s->window[1]);
        }
    } while (s->lookahead < MIN_LOOKAHEAD && s->src_pos != s->src_len);
}

static unsigned longest_match(ds_t *s, unsigned cur_match) {
    unsigned chain_length = MAX_CHAIN;
    const uint8_t *scan = s->window + s->strstart;
    int best_len = (int)s->prev_length;
    int nice_match = NICE_LENGTH;
    unsigned limit = s->strstart > (unsigned)MAX_DIST ? s->strstart - MAX_DIST : 0;
    if (s->prev_length >= GOOD_LENGTH) chain_length >>= 2;
    if ((unsigned)nice_match > s->lookahead) nice_match = (int)s->lookahead;
    /* comparisons never look past the real data: a run that reaches the end of the lookahead
     * is already >= nice_match, which ends the search. */
    int max_len = s->lookahead < MAX_MATCH ? (int)s->lookahead : MAX_MATCH;
    do {
        const uint8_t *match = s->window + cur_match;
        int len = 0;
        while (len < max_len && match[len] == scan[len]) len++;
        if (len > best_len) {
            s->match_start = cur_match;
            best_len = len;
            if (len >= nice_match) break;
        }
    } while ((cur_match = s->prev[cur_match & W_MASK]) > limit && --chain_length != 0);
    return (unsigned)best_len <= s->lookahead ? (unsigned)best_len : s->lookahead;
}

static void tally(ds_t *s, unsigned dist, unsigned lc, int *bflush) {
    if (s->trace_dist) { s->trace_dist[s->trace_n] = (uint16_t)dist; s->trace_lc[s->trace_n] = (uint8_t)lc; }
    s->trace_n++;
    s->d_buf[s->last_lit] = (uint16_t)dist;
    s->l_buf[s->last_lit++] = (uint8_t)lc;
    if (dist == 0) s->lt.freq[lc]++;
    else { s->lt.freq[length_code[lc] + 257]++; s->dt.freq[d_code(dist - 1)]++; }
    *bflush = s->last_lit == LIT_BUFSIZE - 1;
}

static void deflate_slow(ds_t *s) {
    int bflush;
    for (;;) {
        if (s->lookahead < MIN_LOOKAHEAD) {
            fill_window(s);
            if (s->lookahead == 0) break;
        }
        unsigned hash_head = 0;
        if (s->lookahead >= MIN_MATCH) hash_head = insert_string(s, s->strstart);
        s->prev_length = s->match_length; s->prev_match = s->match_start;
        s->match_length = MIN_MATCH - 1;
        if (hash_head != 0 && s->prev_length < MAX_LAZY && s->strstart - hash_head <= (unsigned)MAX_DIST) {
            s->match_length = longest_match(s, hash_head);
            if (s->match_length == MIN_MATCH && s->strstart - s->match_start > TOO_FAR)
                s->match_length = MIN_MATCH - 1;
        }
        if (s->prev_length >= MIN_MATCH && s->match_length <= s->prev_length) {
            unsigned max_insert = s->strstart + s->lookahead - MIN_MATCH;
            tally(s, s->strstart - 1 - s->prev_match, s->prev_length - MIN_MATCH, &bflush);
            s->lookahead -= s->prev_length - 1;
            s->prev_length -= 2;
            do {
                if (++s->strstart <= max_insert) insert_string(s, s->strstart);
            } while (--s->prev_length != 0);
            s->match_available = 0;
            s->match_length = MIN_MATCH - 1;
            s->strstart++;
            if (bflush) flush_block(s, 0);
        } else if (s->match_available) {
            tally(s, 0, s->window[s->strstart - 1], &bflush);
            if (bflush) flush_block(s, 0);
            s->strstart++; s->lookahead--;
        } else {
            s->match_available = 1; s->strstart++; s->lookahead--;
        }
    }
    if (s->match_available) { tally(s, 0, s->window[s->strstart - 1], &bflush); s->match_available = 0; }
    flush_block(s, 1);
}

static ds_t *ds_new(const uint8_t *in, size_t n) {
    init_tables();
    ds_t *s = (ds_t *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->src = in; s->src_len = n;
    s->match_length = s->prev_length = MIN_MATCH - 1;
    init_block(s);
    return s;
}

size_t zo_deflate6(const uint8_t *in, size_t n, uint8_t *out, size_t cap) {
    ds_t *s = ds_new(in, n);
    if (!s) return 0;
    s->out = out; s->out_cap = cap;
    put_byte(s, 0x78); put_byte(s, 0x9c);
    deflate_slow(s);
    uint32_t a = zo_adler32(in, n);
    put_byte(s, a >> 24); put_byte(s, (a >> 16) & 0xff); put_byte(s, (a >> 8) & 0xff); put_byte(s, a & 0xff);
    size_t len = s->overflow ? 0 : s->out_len;
    free(s);
    return len;
}

uint32_t zo_chunk_payload(const uint8_t *in, uint32_t n, uint8_t *out) {
    /* compression.cpp:127-132: avail_out = 65535, return code ignored */
    static const size_t cap = ZO_CHUNK_SIZE + 4096;
    uint8_t *tmp = (uint8_t *)malloc(cap);
    size_t len = zo_deflate6(in, n, tmp, cap);
    if (len > ZO_CHUNK_SIZE) len = ZO_CHUNK_SIZE;
    memcpy(out, tmp, len);
    free(tmp);
    return (uint32_t)len;
}

size_t zo_lz77_symbols(const uint8_t *in, size_t n, uint16_t *dist, uint8_t *lc) {
    ds_t *s = ds_new(in, n);
    if (!s) return 0;
    size_t cap = n + n / 1000 * 6 + 4096;
    s->out = (uint8_t *)malloc(cap); s->out_cap = cap;
    s->trace_dist = dist; s->trace_lc = lc;
    deflate_slow(s);
    size_t k = s->trace_n;
    free(s->out); free(s);
    return k;
}

uint32_t zo_adler32(const uint8_t *in, size_t n) {
    uint32_t a = 1, b = 0;
    while (n) {
        size_t k = n < 5552 ? n : 5552;
        n -= k;
        while (k--) { a += *in++; b += a; }
        a %= 65521; b %= 65521;
    }
    return (b << 16) | a;
}
