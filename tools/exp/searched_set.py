#!/usr/bin/env python3
"""Offline study for lz_match_band: zlib's deflate_slow hands only the positions its lazy parse
visits to longest_match; the band kernel searches every position.  How large is the searched set on the corpora, and how well does a CHEAP
predictor -- the lazy parse run over records from the nearest K candidates only -- cover it?  Then the scheme's cost: cheap records everywhere ->
predicted set -> full search there -> a parse that checks an "exact" bit on every record it consumes, misses repaired and the parse re-run
from the first miss until it consumed exact records only (bit-exact by that check, whatever the predictor did).
Run: python tools/exp/searched_set.py [chunks per corpus]      (builds tools/exp/searched_set.c into /tmp)"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import numpy as np, corpus, workloads, oracle_binding

so = "/tmp/searched_set.so"
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "searched_set.c")])
L = ctypes.CDLL(so)
L.study_chunk.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
L.true_symbols.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]

def check_against_oracle(data):
    o = oracle_binding.load()
    dist, lc = o.symbols(data)
    sym = np.zeros(len(data) + 8, dtype=np.uint32)
    ns = L.true_symbols(data, len(data), sym.ctypes.data)
    assert ns == len(dist), (ns, len(dist))
    p = 0
    for i in range(ns):
        if dist[i] == 0:
            assert sym[i] == (p << 9), i; p += 1
        else:
            ln = int(lc[i]) + 3
            assert sym[i] == (((p - int(dist[i])) << 9) | ln), i; p += ln
    assert p == len(data)

def corpora(nchunks):
    text = [corpus.text_like(workloads.TEXT_SEED0 + i, 262144)[j * 65535:(j + 1) * 65535] for i in range((nchunks + 3) // 4) for j in range(4)][:nchunks]
    sizes = workloads.small_file_sizes(3000)
    small = [workloads.small_file_bytes(i, int(sizes[i])) for i in range(0, 3000, max(1, 3000 // (nchunks * 8)))]
    big_img = [workloads.small_file_bytes(900000 + i, 65535) for i in range(max(2, nchunks // 4))]
    return {"text-like (configs[2])": text, "image-like small files": small, "image-like 64 KB": big_img,
            "lz-heavy": [corpus.lz_heavy(50 + i, 65535) for i in range(max(2, nchunks // 4))]}

def run(name, chunks, K, widen):
    tot = np.zeros(24, dtype=np.int64); maxr = 0; rounds_hist = {}
    for c in chunks:
        if len(c) < 4: continue
        out = np.zeros(24, dtype=np.int64)
        L.study_chunk(c, len(c), K, widen, out.ctypes.data)
        assert out[16] == 1, "the checked parse did not reproduce zlib's symbols"
        r = int(out[9]); out[9] = 0; maxr = max(maxr, r); rounds_hist[r] = rounds_hist.get(r, 0) + 1
        out[18] *= out[19]
        tot += out
    n = tot[0]
    print("%-26s K=%-3d widen=%d | searched %.3f of positions (long chain %.3f) | zlib walks %.1f cand/pos, all-positions band %.1f | cheap exact by construction %.3f of positions with candidates"
          % (name, K, widen, tot[2] / n, tot[3] / n, tot[4] / n, tot[5] / n, tot[17] / max(1, tot[6])))
    print("    predicted set %.3f of positions: precision %.3f recall %.3f; divergences %.0f a chunk, resync after %.1f positions"
          % (tot[7] / n, tot[8] / max(1, tot[7]), tot[8] / max(1, tot[2]), tot[19] / len(chunks), tot[18] / 100 / max(1, tot[19])))
    print("    checked parse: rounds %s (max %d); misses round 2: %.4f of positions, round 3: %.4f; full searches issued %.3f of positions, their candidates %.1f/pos + cheap %.1f/pos = %.1f (band today %.1f); re-parsed %.2f chunk-lengths on top of predict + first checked parse"
          % (sorted(rounds_hist.items()), maxr, tot[14] / n, tot[15] / n, tot[10] / n, tot[11] / n, tot[12] / n, (tot[11] + tot[12]) / n, tot[5] / n, tot[13] / n))

if __name__ == "__main__":
    nch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cs = corpora(nch)
    for c in cs["text-like (configs[2])"][:2] + cs["lz-heavy"][:1] + cs["image-like small files"][:3]: check_against_oracle(c)
    print("restated parse == oracle's symbols on 6 chunks")
    for name, chunks in cs.items():
        for K, widen in ((4, 0), (8, 0), (16, 0), (8, 1), (8, 2)):
            run(name, chunks, K, widen)
