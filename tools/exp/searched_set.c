/* Offline study for DESIGN.md section 4 (round 5): which positions does zlib's deflate_slow hand to longest_match, and how well does a
 * cheap predictor cover that set?  Built and driven by tools/exp/searched_set.py (gcc -O2 -shared).  Study code, not product, not oracle.
 *
 * A restatement of the level-6 search and lazy parse parameterised by the longest chain a search may walk:
 *   mode TRUE   chain = 32 if prev_length >= 8 else 128                          (zlib)
 *   mode CHEAP  chain = min(that, K)                                              (the nearest K candidates: one trip of the band)
 *   mode MIXED  TRUE at positions whose exact flag is set, CHEAP elsewhere        (the parse that checks what it consumes)
 * A cheap record is exact by construction when the true walk would have ended within K candidates (chain exhausted, limit reached,
 * or nice_length hit).  */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_DIST 32506
#define TOO_FAR 4096

typedef struct {
    const uint8_t *d; int n;
    uint16_t *prev;        /* prev[p] = previous position with the same hash at the time p was inserted (0 = NIL) */
    uint8_t *exact;        /* per position: a full-chain record has been computed */
    int K;
    /* counters of the last parse */
    long searched, searched_long, cheap_used, cheap_inexact, steps_walked;
} St;

static void build_links(St *s) {
    static uint16_t head[32768];
    memset(head, 0, sizeof head);
    for (int p = 0; p + 2 < s->n; p++) {
        unsigned h = ((s->d[p] << 10) ^ (s->d[p + 1] << 5) ^ s->d[p + 2]) & 0x7fff;
        s->prev[p] = head[h]; head[h] = (uint16_t)p;
    }
}

/* longest_match at p with best = prev_length; walks at most `chain` candidates.  *ended = 1 when the walk ended for a reason other
 * than the chain budget K' < zlib's budget (i.e. the result equals zlib's). */
static int search(St *s, int p, int prev_length, int chain_true, int chain_use, int *mstart, int *ended) {
    const uint8_t *d = s->d; int n = s->n;
    int lookahead = n - p, best = prev_length, nice = lookahead < 128 ? lookahead : 128;
    int limit = p > MAX_DIST ? p - MAX_DIST : 0;
    int cur = s->prev[p], chain = chain_use, maxlen = lookahead < 258 ? lookahead : 258;
    *ended = 0;
    if (cur == 0 || p - cur > MAX_DIST) { *ended = 1; return best; }
    for (;;) {
        int len = 0; s->steps_walked++;
        while (len < maxlen && d[cur + len] == d[p + len]) len++;
        if (len > best) { best = len; *mstart = cur; if (len >= nice) { *ended = 1; break; } }
        cur = s->prev[cur];
        if (cur <= limit) { *ended = 1; break; }
        if (--chain == 0) { if (chain_use == chain_true) *ended = 1; break; }
    }
    return best < lookahead ? best : lookahead;
}

enum { TRUE_ = 0, CHEAP = 1, MIXED = 2 };

/* The lazy parse.  visited[p] bit0 = handed to longest_match, bit1 = with the short (32) chain.  miss[] collects searched positions whose
 * record was taken from a cheap search that is not known to be exact.  Returns the number of symbols; sym[] = (start << 9) | len (len 0 =
 * literal) for comparison between parses. */
static int parse(St *s, int mode, uint8_t *visited, int *miss, int *nmiss, uint32_t *sym, int *first_miss) {
    const uint8_t *d = s->d; int n = s->n; (void)d;
    int p = 0, match_length = 2, match_start = 0, match_available = 0, ns = 0;
    *nmiss = 0; *first_miss = -1;
    s->searched = s->searched_long = s->cheap_used = s->cheap_inexact = 0;
    if (visited) memset(visited, 0, n + 1);
    while (p < n) {
        int lookahead = n - p;
        int prev_length = match_length, prev_match = match_start;
        match_length = 2;
        int has = lookahead >= 3 && s->prev[p] != 0 && p - s->prev[p] <= MAX_DIST;
        if (has && prev_length < 16) {
            int ct = prev_length >= 8 ? 32 : 128, cu = ct, ended;
            if (mode == CHEAP || (mode == MIXED && !s->exact[p])) cu = s->K < ct ? s->K : ct;
            int ms = match_start;
            match_length = search(s, p, prev_length, ct, cu, &ms, &ended);
            if (match_length > prev_length) match_start = ms;   /* zlib: match_start only moves when a longer match is found */
            s->searched++; if (ct == 128) s->searched_long++;
            if (visited) visited[p] |= 1 | (ct == 32 ? 2 : 0);
            if (cu != ct) { s->cheap_used++; if (!ended) { s->cheap_inexact++; miss[(*nmiss)++] = p; if (*first_miss < 0) *first_miss = p; } }
            if (match_length <= 5 && match_length == 3 && p - match_start > TOO_FAR) match_length = 2;
        }
        if (prev_length >= 3 && match_length <= prev_length) {
            if (sym) sym[ns] = ((uint32_t)(prev_match) << 9) | (uint32_t)prev_length;
            ns++;
            p += prev_length - 1; match_available = 0; match_length = 2;
        } else if (match_available) {
            if (sym) sym[ns] = ((uint32_t)(p - 1) << 9);
            ns++; p++;
        } else { match_available = 1; p++; }
    }
    if (match_available) { if (sym) sym[ns] = ((uint32_t)(n - 1) << 9); ns++; }
    return ns;
}

/* out[]: 0 n, 1 symbols, 2 searched(true), 3 searched with the long chain, 4 candidates walked by zlib (true parse),
 * 5 candidates an all-positions search walks (both records: the 128 walk, as the band does), 6 positions with any candidate,
 * 7 |V1| predicted searched set, 8 |V1 & true|, 9 rounds until the checked parse consumed exact records only,
 * 10 full searches issued in total (V1 inexact + repairs), 11 full-search candidates walked in total, 12 cheap candidates walked (all positions),
 * 13 sum over repair rounds of the positions re-parsed (from the round's first miss to the end), 14 misses in round 2, 15 misses in round 3,
 * 16 identical to the true symbol stream (0/1), 17 positions where cheap == exact by construction (walk ended within K),
 * 18 mean resync distance x 100 of cheap-vs-true divergences, 19 number of divergences */
int study_chunk(const uint8_t *d, int n, int K, int widen, long *out) {
    St s; memset(&s, 0, sizeof s);
    s.d = d; s.n = n; s.K = K;
    s.prev = calloc(n + 8, 2); s.exact = calloc(n + 8, 1);
    uint8_t *vt = malloc(n + 8), *v1 = malloc(n + 8), *vm = malloc(n + 8);
    int *miss = malloc(sizeof(int) * (n + 8)), nmiss, fm;
    uint32_t *st = malloc(4 * (n + 8)), *sm = malloc(4 * (n + 8));
    build_links(&s);
    memset(out, 0, 24 * sizeof(long));
    out[0] = n;
    s.steps_walked = 0;
    int nst = parse(&s, TRUE_, vt, miss, &nmiss, st, &fm);
    out[1] = nst; out[2] = s.searched; out[3] = s.searched_long; out[4] = s.steps_walked;
    /* what the band does today: every position with a candidate walks its (up to) 128 chain */
    long all = 0, withc = 0, byk = 0, cheap_all = 0;
    for (int p = 1; p + 2 < n; p++) {
        int cur = s.prev[p], limit = p > MAX_DIST ? p - MAX_DIST : 0, c = 0;
        if (cur == 0 || p - cur > MAX_DIST) continue;
        withc++;
        while (1) { c++; cur = s.prev[cur]; if (cur <= limit || c == 128) break; }
        all += c; cheap_all += c < K ? c : K; if (c <= K) byk++;
    }
    out[5] = all; out[6] = withc; out[12] = cheap_all; out[17] = byk;
    /* round 1: the cheap parse predicts the searched set */
    parse(&s, CHEAP, v1, miss, &nmiss, sm, &fm);
    long nv1 = 0, inter = 0, full = 0;
    for (int p = 0; p < n; p++) if (v1[p] & 1) { nv1++; if (vt[p] & 1) inter++; }
    /* optional widening: also flag the w positions behind every predicted one */
    if (widen) {
        for (int p = n - 1; p >= 0; p--) if (v1[p] & 1) for (int w = 1; w <= widen && p + w < n; w++) v1[p + w] |= 4;
        nv1 = inter = 0;
        for (int p = 0; p < n; p++) if (v1[p]) { nv1++; if (vt[p] & 1) inter++; }
    }
    out[7] = nv1; out[8] = inter;
    /* resync distance: walk both visited sets; a divergence starts where the cheap parse searches a position the true one does not (or v.v.)
     * and ends at the next position both search after which they agree for 8 searched positions in a row */
    {
        long ndiv = 0, dist = 0; int in_div = 0, start = 0, agree = 0;
        for (int p = 0; p < n; p++) {
            int a = vt[p] & 1, b = v1[p] & 1;
            if (a != b) { if (!in_div) { in_div = 1; start = p; ndiv++; } agree = 0; }
            else if (a && in_div) { if (++agree == 8) { in_div = 0; dist += p - start; } }
        }
        out[18] = ndiv ? dist * 100 / ndiv : 0; out[19] = ndiv;
    }
    s.steps_walked = 0;
    {   /* full searches for the predicted set (only where the cheap record is not exact by construction) */
        for (int p = 0; p < n; p++) if (v1[p]) {
            int cur = s.prev[p], limit = p > MAX_DIST ? p - MAX_DIST : 0, c = 0;
            s.exact[p] = 1;
            if (cur == 0 || p - cur > MAX_DIST || p + 2 >= n) continue;
            while (1) { c++; cur = s.prev[cur]; if (cur <= limit || c == 128) break; }
            if (c > K) { full++; out[11] += c; }
        }
    }
    int rounds = 1; long reparsed = 0;
    for (;;) {
        int nsm = parse(&s, MIXED, vm, miss, &nmiss, sm, &fm);
        if (nmiss == 0) { out[16] = (nsm == nst && memcmp(sm, st, 4 * (size_t)nst) == 0); break; }
        rounds++;
        if (rounds == 2) out[14] = nmiss; if (rounds == 3) out[15] = nmiss;
        reparsed += n - fm;
        for (int i = 0; i < nmiss; i++) {
            int p = miss[i], cur = s.prev[p], limit = p > MAX_DIST ? p - MAX_DIST : 0, c = 0;
            s.exact[p] = 1; full++;
            while (1) { c++; cur = s.prev[cur]; if (cur <= limit || c == 128) break; }
            out[11] += c;
            for (int w = 1; w <= widen && p + w < n; w++) if (!s.exact[p + w]) {   /* the repair widens too */
                int q = p + w; s.exact[q] = 1; cur = s.prev[q]; limit = q > MAX_DIST ? q - MAX_DIST : 0; c = 0;
                if (cur == 0 || q - cur > MAX_DIST || q + 2 >= n) continue;
                while (1) { c++; cur = s.prev[cur]; if (cur <= limit || c == 128) break; }
                if (c > K) { full++; out[11] += c; }
            }
        }
        if (rounds > 200) break;
    }
    out[9] = rounds; out[10] = full; out[13] = reparsed;
    free(s.prev); free(s.exact); free(vt); free(v1); free(vm); free(miss); free(st); free(sm);
    return 0;
}

/* the true parse's symbols, for the driver to compare with the oracle's (zo_lz77_symbols): sym[i] = (start << 9) | len, len 0 = literal */
int true_symbols(const uint8_t *d, int n, uint32_t *sym) {
    St s; memset(&s, 0, sizeof s);
    s.d = d; s.n = n; s.K = 128; s.prev = calloc(n + 8, 2); s.exact = calloc(n + 8, 1);
    int *miss = malloc(sizeof(int) * (n + 8)), nmiss, fm;
    build_links(&s);
    int ns = parse(&s, TRUE_, NULL, miss, &nmiss, sym, &fm);
    free(s.prev); free(s.exact); free(miss);
    return ns;
}
