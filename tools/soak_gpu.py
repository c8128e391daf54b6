#!/usr/bin/env python3
"""One-off parity soak on the GPU box: N chunks of mixed kinds / sizes / stitched segments through the C ABI,
every payload compared with the oracle (CPU restatement), every decode with the oracle's decode.
usage: python tools/soak_gpu.py [N] [seed]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import corpus, oracle_binding

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
zwz = importlib.import_module("parallel-data-compression-and-decompression_amd")
codec = zwz.Codec(0, 4096)
o = oracle_binding.load()
rs = corpus.splitmix64(seed, 6 * N)
kinds = [k for k in corpus.KINDS if k != "lz"]
chunks = []
for i in range(N):
    r = int(rs[6 * i] % 100)
    if r < 60:
        kind = kinds[int(rs[6 * i + 1] % len(kinds))]
        n = int(rs[6 * i + 2] % 65536) if r < 45 else 65535 - int(rs[6 * i + 2] % 700)
        chunks.append(corpus.make(kind, seed * 100000 + i, n))
    else:
        parts, total = [], 0
        for j in range(2 + int(rs[6 * i + 1] % 4)):
            kind = kinds[int((rs[6 * i + 3] >> (7 * j)) % len(kinds))]
            n = 200 + int((rs[6 * i + 4] >> (11 * j)) % 30000)
            n = min(n, 65535 - total)
            if n <= 0:
                break
            parts.append(corpus.make(kind, seed * 100000 + 7 * i + j, n)); total += n
        chunks.append(b"".join(parts))
t0 = time.time()
got = codec.deflate_chunks(chunks)
back, _ = codec.inflate_chunks(got)
t1 = time.time()
bad = 0
for i, (c, g, b) in enumerate(zip(chunks, got, back)):
    if g != o.payload(c):
        bad += 1; print("DEFLATE MISMATCH", i, len(c)); open("/tmp/soak_bad_%d.bin" % i, "wb").write(c)
    elif b != o.inflate(g, 70000)[0]:
        bad += 1; print("INFLATE MISMATCH", i, len(c))
# corrupted and truncated streams: the decoder must stop exactly where the reference's ignored-return-code inflate does
import zlib
rs2 = corpus.splitmix64(seed + 77, 4 * N)
bad_payloads = []
for i in range(min(N, 4000)):
    z = bytearray(got[i]) if len(got[i]) > 8 else bytearray(zlib.compress(b"abcabcabc" * 50, 6))
    mode = int(rs2[4 * i] % 4)
    if mode == 0:
        z[2 + int(rs2[4 * i + 1] % (len(z) - 2))] ^= 1 << int(rs2[4 * i + 2] % 8)
    elif mode == 1:
        z = z[:2 + int(rs2[4 * i + 1] % (len(z) - 2))]
    elif mode == 2:
        for j in range(3):
            z[2 + int((rs2[4 * i + 1] >> (11 * j)) % (len(z) - 2))] ^= 0xff
    else:
        z += bytes(min(int(rs2[4 * i + 1] % 7), 65535 - len(z)))
    bad_payloads.append(bytes(z))
t2 = time.time()
gb, _ = codec.inflate_chunks(bad_payloads)
nbad = 0
for i, (pl, g) in enumerate(zip(bad_payloads, gb)):
    w, total, st = o.inflate(pl, 1 << 20)
    if total <= 65535 and g != w:
        nbad += 1; bad += 1
        if nbad < 5: print("CORRUPT-STREAM MISMATCH", i, len(pl), len(g), len(w))
print("corrupt streams: %d checked, %d mismatches" % (len(bad_payloads), nbad))
print("soak: %d chunks, %d bytes, gpu %.1f s, oracle %.1f s, mismatches %d" % (N, sum(map(len, chunks)), t1 - t0, time.time() - t1, bad))
sys.exit(1 if bad else 0)
