#!/usr/bin/env python3
"""Offline study for a subsequence-parallel inflate (round 5): a lane that starts decoding at an arbitrary bit of a dynamic-Huffman block, with the
block's tables, is on a false path until it lands on a bit where a true symbol starts -- from there its symbols are the stream's.  How many bits
(and symbols) does that take on the corpora?  For every multiple of S bits inside the first block of zlib level-6 streams (zlib.compress = the
oracle's payload, tests/test_oracle.py): decode on until a true symbol start is hit; report the distribution of bits and symbols decoded before
that, and the share of starts that never get there within 4 S bits (or run into an invalid code).   Run: python tools/exp/inflate_sync.py"""
import os, sys, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import corpus, workloads

LBASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DEXT = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
ORDER = [16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]

class Bits:
    def __init__(self, data): self.d = int.from_bytes(data, "little"); self.n = 8 * len(data)
    def get(self, pos, k): return (self.d >> pos) & ((1 << k) - 1)

def build(lens):
    """canonical codes -> dict (length, code as read LSB-first) -> symbol"""
    bl = [0] * 16
    for l in lens: bl[l] += 1
    bl[0] = 0
    code, nxt = 0, [0] * 16
    for b in range(1, 16): code = (code + bl[b - 1]) << 1; nxt[b] = code
    table = {}
    for s, l in enumerate(lens):
        if l:
            c = nxt[l]; nxt[l] += 1
            rev = int(format(c, "0%db" % l)[::-1], 2)
            table[(l, rev)] = s
    return table

def dec(bits, pos, table):
    for l in range(1, 16):
        s = table.get((l, bits.get(pos, l)))
        if s is not None: return s, pos + l
    return None, pos

def first_block(payload):
    """-> (bits, first symbol bit, lit table, dist table, set of true symbol starts, end of block bit) of a dynamic first block, or None"""
    b = Bits(payload)
    pos = 16
    last, typ = b.get(pos, 1), b.get(pos + 1, 2); pos += 3
    if typ != 2: return None
    hlit, hdist, hclen = b.get(pos, 5) + 257, b.get(pos + 5, 5) + 1, b.get(pos + 10, 4) + 4; pos += 14
    cl = [0] * 19
    for i in range(hclen): cl[ORDER[i]] = b.get(pos, 3); pos += 3
    ct = build(cl)
    lens = []
    while len(lens) < hlit + hdist:
        s, pos = dec(b, pos, ct)
        if s < 16: lens.append(s)
        elif s == 16: r = 3 + b.get(pos, 2); pos += 2; lens += [lens[-1]] * r
        elif s == 17: r = 3 + b.get(pos, 3); pos += 3; lens += [0] * r
        else: r = 11 + b.get(pos, 7); pos += 7; lens += [0] * r
    lt, dt = build(lens[:hlit]), build(lens[hlit:])
    starts, p0 = set(), pos
    while True:
        starts.add(pos)
        nxt, kind = step(b, pos, lt, dt)
        if kind == "eob": break
        pos = nxt
    return b, p0, lt, dt, starts, pos

def step(b, pos, lt, dt):
    s, p = dec(b, pos, lt)
    if s is None: return pos, "bad"
    if s < 256: return p, "lit"
    if s == 256: return p, "eob"
    if s > 285: return pos, "bad"
    p += LEXT[s - 257]
    d, p = dec(b, p, dt)
    if d is None or d > 29: return pos, "bad"
    return p + DEXT[d], "match"

def study(name, chunks, S):
    bits_to, syms_to, never, n = [], [], 0, 0
    sym_bits = []
    for c in chunks:
        fb = first_block(zlib.compress(c, 6))
        if fb is None: continue
        b, p0, lt, dt, starts, end = fb
        sym_bits.append((end - p0) / max(1, len(starts)))
        for a in range(p0 + S - (p0 % S), end - 64, S):
            n += 1
            pos, k = a, 0
            while pos not in starts:
                nxt, kind = step(b, pos, lt, dt)
                if kind in ("bad", "eob") or nxt - a > 4 * S or nxt >= end: pos = None; break
                pos, k = nxt, k + 1
            if pos is None: never += 1
            else: bits_to.append(pos - a); syms_to.append(k)
    bits_to.sort(); syms_to.sort()
    q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
    print("== %s, subsequences of %d bits: %d starts, %.1f bits a symbol; on a true start at once: %.1f %%; bits until the first true start: mean %.0f, median %d, 90 %% %d, 99 %% %d; symbols decoded on the false path: mean %.1f, 90 %% %d; not within %d bits: %.2f %%"
          % (name, S, n, sum(sym_bits) / len(sym_bits), 100.0 * sum(1 for x in bits_to if x == 0) / n, sum(bits_to) / len(bits_to), q(bits_to, .5), q(bits_to, .9), q(bits_to, .99),
             sum(syms_to) / len(syms_to), q(syms_to, .9), 4 * S, 100.0 * never / n))

if __name__ == "__main__":
    text = [corpus.text_like(workloads.TEXT_SEED0 + i, 262144)[j * 65535:(j + 1) * 65535] for i in range(3) for j in range(2)]
    sizes = workloads.small_file_sizes(400)
    img = [workloads.small_file_bytes(i, sizes[i]) for i in range(400) if sizes[i] > 2000][:120]
    for S in (128, 256):
        study("text-like (configs[2])", text, S)
        study("image-like small files (configs[3])", img, S)


def long_codes(name, chunks):
    """share of TRUE symbols whose literal/length code is longer than the kernel's 10-bit fast table (or whose distance code is longer than 8),
    and the share of ARBITRARY bit offsets at which a decode runs into one (or into no code at all)"""
    n = long_true = n_any = long_any = 0
    for c in chunks:
        fb = first_block(zlib.compress(c, 6))
        if fb is None: continue
        b, p0, lt, dt, starts, end = fb
        def is_long(pos):
            for l in range(1, 16):
                s = lt.get((l, b.get(pos, l)))
                if s is not None:
                    if l > 10: return True
                    if s <= 256: return False
                    p = pos + l + LEXT[s - 257]
                    for dl in range(1, 16):
                        d = dt.get((dl, b.get(p, dl)))
                        if d is not None: return dl > 8
                    return True
            return True
        for pos in starts: n += 1; long_true += is_long(pos)
        for pos in range(p0, min(end, p0 + 40000), 7): n_any += 1; long_any += is_long(pos)
    print("== %s: true symbols with a code beyond the fast tables: %.2f %% (one in %.0f); arbitrary offsets: %.2f %%" % (name, 100.0 * long_true / n, n / max(1, long_true), 100.0 * long_any / n_any))

if __name__ == "__main__":
    long_codes("text-like", text)
    long_codes("image-like small files", img)
