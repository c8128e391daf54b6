#!/usr/bin/env python3
"""Offline study for lz_match_band's first pass: how many candidates does an entry of the sorted array have on the
text corpus, and how many of them could a second sort level -- sub-buckets by the byte(s) behind the trigram -- leave out?  A candidate that does
not share the fourth byte decides nothing once the nearest candidate is known (its length is 3).   Run: python tools/exp/band_candidates.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, corpus, workloads

for seed in range(2):
    data = corpus.text_like(workloads.TEXT_SEED0 + seed, 262144)[:65535]
    d = np.frombuffer(data, dtype=np.uint8).astype(np.int64)
    n = len(d) - 2
    h = ((d[:-2] << 10) ^ (d[1:-1] << 5) ^ d[2:]) & 0x7fff
    order = np.lexsort((np.arange(n), h)); pos = order; hs = h[order]
    pad = np.concatenate([d, np.zeros(16, dtype=np.int64)])
    b3, b4 = pad[pos + 3], pad[pos + 4]
    tot = s1 = s2 = has1 = ents = start = 0
    for u in range(n):
        if u and hs[u] != hs[u - 1]: start = u
        lo = max(start, u - 128); c = 0
        for v in range(u - 1, lo - 1, -1):                 # zlib's candidates: same bucket, nearest first, <= 128, within MAX_DIST, not position 0
            if pos[u] - pos[v] >= 32506 or pos[v] == 0: break
            c += 1
        if c == 0: continue
        ents += 1
        sl = slice(u - c, u)
        e1 = b3[sl] == b3[u]; e2 = e1 & (b4[sl] == b4[u])
        tot += c; s1 += int(e1.sum()); s2 += int(e2.sum()); has1 += int(e1.any())
    print("chunk %d: %d entries with candidates, %.1f candidates each; sharing the 4th byte %.3f, the 4th and 5th %.3f; entries with a 4th-byte sharer %.3f"
          % (seed, ents, tot / ents, s1 / tot, s2 / tot, has1 / ents))
