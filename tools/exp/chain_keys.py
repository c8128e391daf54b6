#!/usr/bin/env python3
"""Offline study behind lz_match's work order: how well does a cheap key predict the number of candidates the
chain walk visits?  A wave's trip lasts as long as its longest chain, so lane utilisation = work / (64 x sum of per-wave
maxima) when a tile's positions are sorted by the key and taken 64 at a time.  Run: python tools/exp/chain_keys.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, corpus

def analyse(seed):
    d = np.frombuffer(corpus.text_like(seed, 65535), dtype=np.uint8).astype(np.int64)
    h = ((d[:-2] << 10) ^ (d[1:-1] << 5) ^ d[2:]) & 0x7fff
    n = len(h)
    order = np.lexsort((np.arange(n), h)); hs = h[order]; ps = np.arange(n)[order]
    start = np.r_[True, hs[1:] != hs[:-1]]; grp = np.cumsum(start) - 1; first_idx = np.nonzero(start)[0]
    idx = np.arange(n)
    lo = np.empty(n, dtype=np.int64)
    for g in range(len(first_idx)):
        a = first_idx[g]; b = first_idx[g + 1] if g + 1 < len(first_idx) else n
        p = ps[a:b]; lo[a:b] = a + np.searchsorted(p, p - 32505, side="left")
    cand_s = np.minimum(idx - lo, 128)            # candidates in range, zlib's cap (early exits ignored)
    def unsort(a):
        o = np.empty(n, dtype=a.dtype); o[ps] = a; return o
    def pred(k):
        out = np.full(n, -1, dtype=np.int64); ok = idx - k >= first_idx[grp]; out[ok] = ps[idx[ok] - k]; return unsort(out)
    cand, pos = unsort(cand_s), np.arange(n)
    P = {k: pred(k) for k in (1, 2, 4, 8, 16)}
    def steps_for(key):
        tot = 0
        for t in range(4):
            sl = slice(t * 16384, min(n, (t + 1) * 16384))
            c2 = cand[sl][np.argsort(key[sl], kind="stable")]
            tot += sum(int(c2[w:w + 64].max()) for w in range(0, len(c2), 64))
        return tot
    res = {}
    key = np.full(n, 7)
    ok = (P[1] > 0) & (pos - P[1] <= 32506) & (P[2] > 0) & (pos - P[2] <= 32506)
    lg = np.floor(np.log2(np.maximum(pos - P[2], 1))).astype(int)
    key[ok] = np.minimum(lg[ok], 13) >> 1
    res["round 1: log2(dist to 2nd predecessor)/2, 8 buckets"] = steps_for(key)
    for k in (2, 4, 8, 16):
        win = np.minimum(pos, 32506).astype(float); Dk = (pos - P[k]).astype(float)
        e = np.where((P[k] > 0) & (Dk <= 32506), np.minimum(128.0, win * k / np.maximum(Dk, 1)), -1.0)
        for kk in (8, 4, 2, 1):
            if kk < k:
                small = (e < 0) & (P[kk] > 0) & (pos - P[kk] <= 32506); e[small] = np.maximum(e[small], kk)
        e[e < 0] = 0
        res["density from the %2d-th predecessor, 15 half-octave buckets" % k] = steps_for(np.floor(np.log2(np.maximum(e, 1)) * 2).astype(int))
    res["true chain length"] = steps_for(cand)
    return cand.sum(), res

if __name__ == "__main__":
    tot, W = {}, 0
    for seed in (2_000_000, 2_000_001, 2_000_007):
        w, r = analyse(seed); W += w
        for k, v in r.items(): tot[k] = tot.get(k, 0) + v
    for k, v in tot.items(): print("%-62s wave-steps/chunk %7d  lane utilisation %.3f" % (k, v / 3, W / (64 * v)))
