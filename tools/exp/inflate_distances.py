#!/usr/bin/env python3
"""Offline study for inflate's LDS history ring: where do the back-references of the two corpora point?
For every match of zlib's level-6 parse (the oracle's symbol list: the same symbols inflate decodes), the distance, weighted
by matches and by copied bytes; cumulative shares at ring sizes 256 B ... 32 KiB.  Also: the share of OUTPUT bytes that come
from literals, from matches inside a ring of R bytes, from further back.   Run: python tools/exp/inflate_distances.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, corpus, oracle_binding, workloads

o = oracle_binding.load()
RINGS = [256, 512, 1024, 2048, 4096, 8192, 16384, 32768]

def study(name, chunks):
    dist_all, len_all, n_lit, n_bytes = [], [], 0, 0
    for c in chunks:
        d, l = o.symbols(c)          # d == 0: literal (l = the byte); else match of length l + 3 ... see zo_lz77_symbols
        m = d != 0
        dist_all.append(d[m].astype(np.int64)); len_all.append(l[m].astype(np.int64) + 3)
        n_lit += int((~m).sum()); n_bytes += len(c)
    d, l = np.concatenate(dist_all), np.concatenate(len_all)
    assert n_lit + int(l.sum()) == n_bytes, (n_lit, int(l.sum()), n_bytes)
    print("== %s: %d chunks, %d bytes, %d literals, %d matches (mean length %.1f, mean distance %.0f)" % (name, len(chunks), n_bytes, n_lit, len(d), l.mean(), d.mean()))
    print("   ring   matches<=ring   match bytes<=ring   output bytes NOT served by ring (of all output)")
    for r in RINGS:
        inside = d <= r
        print("  %6d      %5.1f %%          %5.1f %%              %5.2f %%" % (r, 100.0 * inside.mean(), 100.0 * l[inside].sum() / l.sum(), 100.0 * l[~inside].sum() / n_bytes))
    # a ring whose bytes are flushed to HBM in 1 KiB pieces holds between R - 1024 and R bytes of history: the guaranteed window is R - 1024
    return d, l

text = [corpus.text_like(workloads.TEXT_SEED0 + i, 262144)[j * 65535:(j + 1) * 65535] for i in range(6) for j in range(4)]
study("text-like (BASELINE configs[2])", text)
sizes = workloads.small_file_sizes(3000)
img = [workloads.small_file_bytes(i, sizes[i]) for i in range(3000)]
study("image-like small files (configs[3], 3 000 files)", img)
study("image-like, full 65 535-byte chunks", [corpus.gradient(77 + i, 65535) for i in range(12)])
