#!/usr/bin/env python3
"""What lz_dense_list's sample sees on the corpora, beside what the two searches would have to walk (round 5: the image-like small files are sent to
sort + band by the sample's rule -- one in five of the first 2 048 trigrams falls into a bucket taken already -- and run 19 ms a 370 000-file pass faster on
the chain walk; text-like chunks of every size from 4 KB run 1.4 - 3.5x faster on the band: tools/exp/dense_crossover.sh).  Per chunk: the sample's repeat
share, and the mean number of chain candidates a position has in the whole chunk (same bucket, in front of it, at most 128)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, corpus, workloads

def stats(c):
    a = np.frombuffer(c, dtype=np.uint8).astype(np.uint32)
    if len(a) < 3: return None
    h = ((a[:-2] << 10) ^ (a[1:-1] << 5) ^ a[2:]) & 0x7fff
    s = h[:2048]
    _, first = np.unique(s, return_index=True)
    repeats = len(s) - len(first)
    # candidates a position: rank inside its bucket, capped at 128
    order = np.argsort(h, kind="stable")
    hs = h[order]
    start = np.r_[0, np.flatnonzero(hs[1:] != hs[:-1]) + 1]
    rank = np.arange(len(hs)) - np.repeat(start, np.diff(np.r_[start, len(hs)]))
    # runs of second-level repeats in the sample: positions whose bucket had been hit TWICE before
    cnt = np.zeros(32768, dtype=np.int32); twice = 0
    for x in s:
        twice += cnt[x] >= 2; cnt[x] += 1
    return repeats / len(s), float(np.minimum(rank, 128).mean()), twice / len(s), len(c)

def show(name, chunks):
    r = [stats(c) for c in chunks]; r = [x for x in r if x]
    a = np.array(r)
    print("%-44s chunks %4d  mean %6.0f B  sample repeat share: mean %.3f (min %.3f, max %.3f)  third-or-later hits: %.3f  candidates a position: %.1f"
          % (name, len(r), a[:, 3].mean(), a[:, 0].mean(), a[:, 0].min(), a[:, 0].max(), a[:, 2].mean(), a[:, 1].mean()))

if __name__ == "__main__":
    show("text-like 64 KB", [corpus.text_like(workloads.TEXT_SEED0 + i, 65535) for i in range(12)])
    show("text-like 8 KB", [corpus.text_like(workloads.TEXT_SEED0 + 50 + i, 8192) for i in range(40)])
    sizes = workloads.small_file_sizes(600)
    show("image-like small files (configs[3])", [workloads.small_file_bytes(i, sizes[i]) for i in range(600)])
    show("image-like 64 KB (gradient)", [corpus.gradient(77 + i, 65535) for i in range(8)])
    show("incompressible 64 KB", [corpus.random_bytes(5 + i, 65535) for i in range(8)])
    for kind in ("lowent", "skewed", "lz_heavy"):
        f = getattr(corpus, kind, None)
        if f:
            try: show(kind + " 64 KB", [f(9 + i, 65535) for i in range(6)])
            except TypeError: pass
