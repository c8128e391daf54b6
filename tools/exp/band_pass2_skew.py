#!/usr/bin/env python3
"""Offline study of lz_match_band's second pass (csrc/zwz_band.hip): a flagged entry walks its SHARERS -- the candidates of its band
that agree with it on the trigram and the eight bytes behind it -- and a wave's trip lasts as long as its longest walk.  How long are
the walks, and how much of a wave is busy when the flagged entries are taken 64 at a time in sorted-array order (what the kernel does),
sorted by walk length, or with finished lanes refilled?   Run: python tools/exp/band_pass2_skew.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, corpus, workloads

def study(data, tile=5632):
    d = np.frombuffer(data, dtype=np.uint8).astype(np.int64)
    n = len(d) - 2
    h = ((d[:-2] << 10) ^ (d[1:-1] << 5) ^ d[2:]) & 0x7fff
    order = np.lexsort((np.arange(n), h))                 # the sorted array: (bucket, position)
    pos = order; hs = h[order]
    pad = np.concatenate([d, np.zeros(16, dtype=np.int64)])
    key = np.zeros(n, dtype=object)
    gram = [bytes(pad[p:p + 11].astype(np.uint8)) for p in range(n)]   # trigram + the eight bytes behind it
    visits = np.zeros(n, dtype=np.int64)                  # per sorted index u: sharers among its candidates
    last = {}
    # candidates of u: u-1 .. u-128 while same bucket, position != 0, distance < 32506
    start = 0
    for u in range(n):
        if u and hs[u] != hs[u - 1]: start = u
        lo = max(start, u - 128)
        g = gram[pos[u]]
        c = 0
        for v in range(u - 1, lo - 1, -1):
            if pos[u] - pos[v] >= 32506 or pos[v] == 0: break
            if gram[pos[v]] == g: c += 1
        visits[u] = c
    flagged = np.nonzero(visits)[0]
    tot = int(visits.sum())
    out = {"flagged": len(flagged), "visits": tot, "mean": tot / max(1, len(flagged)), "max": int(visits.max())}
    def trips(seq):                                       # groups of 64 in this order: a trip lasts as long as its longest walk
        t = 0
        for i in range(0, len(seq), 64): t += int(seq[i:i + 64].max())
        return t
    per_tile_order, per_tile_sorted, refill = 0, 0, 0
    for a in range(0, n, tile):
        f = flagged[(flagged >= a) & (flagged < a + tile)]
        v = visits[f]
        per_tile_order += trips(v)
        per_tile_sorted += trips(np.sort(v)[::-1])
        refill += -(-int(v.sum()) // 64) + (int(v.max()) if len(v) else 0)   # lanes refilled: total work / 64, plus the last walk's tail
    out.update({"wave_trips_array_order": per_tile_order, "wave_trips_sorted": per_tile_sorted, "wave_trips_refilled": refill,
                "busy_array_order": tot / 64 / max(1, per_tile_order), "busy_sorted": tot / 64 / max(1, per_tile_sorted)})
    return out

for i in range(2):
    data = corpus.text_like(workloads.TEXT_SEED0 + i, 262144)[:65535]
    print("text chunk", i, study(data))
sizes = workloads.small_file_sizes(40)
print("image-like 64 KB", study(corpus.gradient(77, 65535)))
