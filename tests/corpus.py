"""Deterministic synthetic corpora for parity tests and bench (own PRNG: splitmix64, vectorised).

Kinds mirror SURVEY.md §8(d): uniform random bytes (BASELINE config 1/2), Zipf "text-like" words
(config 3, zlib-6 ratio ~0.36), plus LZ-heavy / periodic / low-entropy / skewed shapes used to
reach the Huffman and lazy-match corner cases of SURVEY.md Appendix B.
No numpy Generator is used, so bytes are identical on every box and numpy version.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n pseudo-random uint64 values, element i depends only on (seed, i)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(0x9E3779B97F4A7C15)
             * (np.arange(1, n + 1, dtype=np.uint64)))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_bytes(seed: int, n: int) -> bytes:
    if n == 0:
        return b""
    return splitmix64(seed, (n + 7) // 8).view(np.uint8)[:n].tobytes()


def _u01(seed: int, n: int) -> np.ndarray:
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def text_like(seed: int, n: int, vocab: int = 4096) -> bytes:
    """Zipf(1/rank) over `vocab` random lowercase words of length 2-8, space separated."""
    if n == 0:
        return b""
    wl = (splitmix64(seed ^ 0x1111, vocab) % np.uint64(7)).astype(np.int64) + 2
    letters = (splitmix64(seed ^ 0x2222, int(wl.sum())) % np.uint64(26)).astype(np.uint8) + 97
    starts = np.concatenate([[0], np.cumsum(wl)[:-1]])
    p = 1.0 / np.arange(1, vocab + 1)
    cdf = np.cumsum(p / p.sum())
    nwords = n // 3 + 8
    idx = np.minimum(np.searchsorted(cdf, _u01(seed ^ 0x3333, nwords)), vocab - 1)
    lens = wl[idx] + 1
    total = int(lens.sum())
    out = np.full(total, 32, dtype=np.uint8)
    pos = np.concatenate([[0], np.cumsum(lens)[:-1]])
    for k in range(8):
        m = wl[idx] > k
        out[pos[m] + k] = letters[starts[idx[m]] + k]
    while total < n:  # very unlikely
        out = np.concatenate([out, out]); total *= 2
    return out[:n].tobytes()


def low_entropy(seed: int, n: int, k: int = 4) -> bytes:
    return (splitmix64(seed, n) % np.uint64(k)).astype(np.uint8).tobytes()


def periodic(seed: int, n: int) -> bytes:
    per = int(splitmix64(seed ^ 0x77, 1)[0] % np.uint64(300)) + 1
    unit = random_bytes(seed, per)
    return (unit * (n // per + 1))[:n]


def skewed(seed: int, n: int, nsym: int = 19) -> bytes:
    """Fibonacci-weighted symbol frequencies: drives Huffman depth past the 15-bit limit."""
    f = [1, 1]
    while len(f) < nsym:
        f.append(f[-1] + f[-2])
    cdf = np.cumsum(np.array(f, dtype=np.float64))
    cdf /= cdf[-1]
    return (np.minimum(np.searchsorted(cdf, _u01(seed, n)), nsym - 1) * 7 + 1).astype(np.uint8).tobytes()


def lz_heavy(seed: int, n: int) -> bytes:
    """Random literals interleaved with back-references at distances up to 40000."""
    r = splitmix64(seed, 3 * (n // 4 + 16))
    out = bytearray(random_bytes(seed ^ 0x99, min(n, 64)))
    i = 0
    while len(out) < n:
        a, b, c = int(r[i]), int(r[i + 1]), int(r[i + 2]); i += 3
        if a % 10 < 7 and len(out) > 8:
            d = b % min(len(out), 40000) + 1
            ln = c % 298 + 3
            for _ in range(ln):
                out.append(out[-d])
        else:
            out += random_bytes(b, c % 20 + 1)
    return bytes(out[:n])


def gradient(seed: int, n: int) -> bytes:
    """'Image-like': smooth ramp + small noise."""
    x = np.arange(n, dtype=np.int64)
    noise = (splitmix64(seed, n) % np.uint64(5)).astype(np.int64) - 2
    return ((x // 7 + (x % 251) // 3 + noise) & 255).astype(np.uint8).tobytes()


KINDS = {
    "random": random_bytes, "text": text_like, "lowent": low_entropy, "periodic": periodic,
    "skewed": skewed, "lz": lz_heavy, "gradient": gradient,
    "zeros": lambda seed, n: bytes(n),
}


def make(kind: str, seed: int, n: int) -> bytes:
    return KINDS[kind](seed, n)


def golden_tree():
    """The small directory tree behind tests/golden/tree_N*/ (SURVEY.md §8c item iii)."""
    files = {
        "empty.bin": b"",
        "hello.txt": b"hello world, hello zwz, hello MI355X\n",
        "exact.bin": random_bytes(101, 65535),           # exact multiple: extra empty chunk + truncation
        "text100k.txt": text_like(102, 100000),
        "rand140k.bin": random_bytes(103, 140000),
        "sub/dir/nested.txt": text_like(104, 70000),
        "sub/grad.raw": gradient(105, 30000),
    }
    for i in range(7):                                   # equal sizes: order = readdir + introsort
        files["same/eq%d.dat" % i] = low_entropy(200 + i, 1234)
    return files


_DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 32769]


def deep_distance_tree(seed: int, k: int = 17, first_code: int = 13, margin: float = 0.02) -> bytes:
    """A chunk whose last block's DISTANCE tree is deeper than 15 bits, so that zlib's gen_bitlen has to repair it: 32 KiB of
    random bytes (two blocks of literals), then 4-byte copies of earlier 4-byte slots, each slot copied from at most once (the
    match zlib finds is then the one intended), the distance codes drawn with counts w[i+2] = w[0] + ... + w[i] + 1 + 2 % --
    a little steeper than Fibonacci, so the Huffman tree stays a chain when a handful of matches come out differently.
    (tests/test_emu.py checks on the CPU that the repair is reached for the seeds the GPU test uses.)"""
    import math
    import numpy as np
    rs = np.random.RandomState(seed)
    w = [1, 1]
    while len(w) < k:
        s = sum(w[:-1])
        w.append(s + 1 + math.ceil(margin * s))
    codes = np.repeat(np.arange(first_code, first_code + k), w)
    rs.shuffle(codes)
    out = bytearray(rs.randint(0, 256, 32768).astype(np.uint8).tobytes())
    used = np.zeros(16384, dtype=bool)
    last = -1
    for c in codes:
        c = int(c); p = len(out)
        lo = _DIST_BASE[c]; hi = min(_DIST_BASE[c + 1] - 1, 32500, p)
        slots = np.arange((p - hi + 3) // 4, (p - lo) // 4 + 1)
        slots = slots[(~used[slots]) & (slots != last + 1)]          # never the slot right behind the last source: the two copies would be one match
        rs.shuffle(slots)
        for sl in slots:
            sl = int(sl)
            if last < 0 or out[4 * sl] != out[4 * last + 4]:        # ... and the previous match must not run on into this copy
                break
        else:
            raise AssertionError((c, p))
        used[sl] = True; last = sl
        out += out[4 * sl:4 * sl + 4]
    return bytes(out)
