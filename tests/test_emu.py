"""The product's portable codec cores (csrc/lz_core.h, csrc/huff_core.h), host-compiled and run in
the kernels' decomposition, against the oracle.  Guards the reformulation itself (per-position
match tables + table walk) independently of the GPU."""
import zlib

import pytest

import corpus
import emu_binding


@pytest.fixture(scope="module")
def emu():
    return emu_binding.load()


SIZES = [0, 1, 2, 3, 4, 9, 258, 259, 300, 4097, 16383, 16384, 16385, 32768, 40000, 65274, 65509, 65534, 65535]


@pytest.mark.parametrize("kind", list(corpus.KINDS))
def test_cores_match_oracle(emu, oracle, kind):
    for i, n in enumerate(SIZES):
        if kind == "lz" and 40000 < n < 65535:
            continue
        data = corpus.make(kind, 3000 + i, n)
        assert emu_binding.chunk_stream(emu, data) == oracle.deflate6(data), (kind, n)


def test_window_slide_corner(emu, oracle):
    for t in range(12):
        n = 65300 + 19 * t
        a = bytearray(corpus.random_bytes(900 + t, n))
        a[65274:65274 + 5] = a[32768:32768 + 5]
        assert emu_binding.chunk_stream(emu, bytes(a)) == oracle.deflate6(bytes(a))


def test_block_boundary_counts(emu, oracle):
    # symbol counts straddling 16383 / 32766: literal-only inputs of exactly those sizes
    for n in [16382, 16383, 16384, 32765, 32766, 32767, 49149, 49150]:
        data = corpus.skewed(77 + n, n, nsym=12)
        assert emu_binding.chunk_stream(emu, data) == oracle.deflate6(data), n


@pytest.mark.skipif(zlib.ZLIB_RUNTIME_VERSION != "1.2.11", reason="live libz is not the pinned 1.2.11")
def test_cores_match_libz_fuzz(emu):
    rs = corpus.splitmix64(2024, 3 * 120)
    kinds = list(corpus.KINDS)
    for i in range(120):
        kind = kinds[int(rs[3 * i] % len(kinds))]
        n = int(rs[3 * i + 2] % 65536)
        if kind == "lz":
            n //= 3
        data = corpus.make(kind, 7000 + i, n)
        assert emu_binding.chunk_stream(emu, data) == zlib.compress(data, 6), (kind, n)


def _records(emu, data):
    import ctypes
    n = len(data)
    a = (ctypes.c_uint32 * (n + 1))(); b = (ctypes.c_uint32 * (n + 1))()
    out = ctypes.create_string_buffer(70000)
    emu.emu_chunk_stream(data, n, out, 70000, a, b)
    return list(a)[:n], list(b)[:n]


def _band(emu, data, tile, fmt):
    import ctypes
    n = len(data)
    a = (ctypes.c_uint32 * (n + 1))(); b = (ctypes.c_uint32 * (n + 1))()
    pure = emu.emu_band_records(data, n, tile, fmt, a, b)
    return list(a)[:n], list(b)[:n], pure


@pytest.mark.parametrize("kind", list(corpus.KINDS))
def test_banded_search_gives_lz_search_records(emu, kind):
    """csrc/lz_band.h (sorted positions, banded keys, sharer chains) against csrc/lz_core.h's chain walk, record by record:
    both 8-byte formats, tiles small enough that chains cross halos."""
    for i, n in enumerate([0, 1, 2, 3, 4, 11, 12, 13, 300, 4097, 20000, 40000, 65274 + 7, 65535]):
        if kind == "lz" and 20000 < n < 65535:
            continue
        data = corpus.make(kind, 5000 + i, n)
        want = _records(emu, data)
        for tile, fmt in ((6016, -1), (6016, 0), (5632, -1), (192, -1), (64, 0)):
            if tile < 1000 and n > 20000:
                continue
            got = _band(emu, data, tile, fmt)
            assert got[0] == want[0], (kind, n, tile, fmt, "e128")
            assert got[1] == want[1], (kind, n, tile, fmt, "e32")


def test_banded_search_corners(emu):
    # a first candidate at exactly MAX_DIST (allowed), at MAX_DIST once zlib's window has slid (not allowed), position 0 as the
    # only earlier occurrence (NIL), long runs that stop at `nice`, the chunk's last positions
    base = bytearray(corpus.random_bytes(4242, 65535))
    cases = []
    a = bytearray(base); a[32506 + 100:32506 + 120] = a[100:120]; cases.append(bytes(a))
    a = bytearray(base); a[65274:65274 + 12] = a[32768:32768 + 12]; cases.append(bytes(a))
    a = bytearray(base); a[65275:65275 + 12] = a[32769:32769 + 12]; cases.append(bytes(a))
    a = bytearray(base); a[5000:5040] = a[0:40]; cases.append(bytes(a))
    a = bytearray(base); a[65535 - 40:] = a[1000:1040]; cases.append(bytes(a))
    a = bytearray(base); a[65535 - 9:] = a[1000:1009]; a[65535 - 300:65535 - 291] = a[1000:1009]; cases.append(bytes(a))
    cases.append(bytes(b"abcabcabd" * 7000)[:65535])
    cases.append(corpus.text_like(77, 65535, vocab=64))
    cases.append(bytes(corpus.text_like(78, 30000)) + corpus.random_bytes(79, 20000) + bytes(15535))
    for idx, data in enumerate(cases):
        want = _records(emu, data)
        for tile, fmt in ((6016, -1), (6016, 0), (5632, 0), (1024, -1)):
            got = _band(emu, data, tile, fmt)
            assert got[0] == want[0] and got[1] == want[1], (idx, tile, fmt)


@pytest.mark.parametrize("kind", list(corpus.KINDS))
def test_lazy_parse_on_speculative_segments_gives_the_table_walk(emu, kind):
    """csrc/lz_lazy.h -- zlib's lazy parse with its searches on demand, lanes starting at segment heads on the assumption that nothing
    is pending, stopping where they meet an owner's mark, the true chain of segments replayed over the marks (the lz_lazy kernel's
    decomposition) -- against lz_core.h's records of every position + table walk: symbol starts, match starts, chosen records.  Segment
    sizes from the kernel's (128; 32 in its ring form) down to 16, lanes advanced one after the other, last first and in random orders."""
    import ctypes
    st = (ctypes.c_uint64 * 4)()
    for i, n in enumerate([0, 1, 2, 3, 4, 5, 11, 130, 300, 4097, 20000, 65274 + 7, 65535]):
        if kind == "lz" and 20000 < n < 65535:
            continue
        data = corpus.make(kind, 7000 + i, n)
        for seg, seeds in ((128, (0, 1, 2, 3)), (32, (1, 2)), (16, (3,))):
            for seed in seeds:
                r = emu.emu_lazy_check(data, n, seg, seed, st)
                assert r == 0, (kind, n, seg, seed, hex(r))
        if n == 65535 and kind == "text":
            assert st[2] > 1000                      # most segments' owners end up on the chain of true lanes


def test_lazy_parse_corners(emu):
    # a first candidate at exactly MAX_DIST (allowed), at MAX_DIST once zlib's window has slid (not), position 0 as the only earlier
    # occurrence (NIL), matches of max_lazy bytes and more (no search behind them), nice matches, matches that end with the data
    base = bytearray(corpus.random_bytes(4242, 65535))
    cases = []
    a = bytearray(base); a[32506 + 100:32506 + 120] = a[100:120]; cases.append(bytes(a))
    a = bytearray(base); a[65274:65274 + 12] = a[32768:32768 + 12]; cases.append(bytes(a))
    a = bytearray(base); a[65275:65275 + 12] = a[32769:32769 + 12]; cases.append(bytes(a))
    a = bytearray(base); a[5000:5040] = a[0:40]; cases.append(bytes(a))
    a = bytearray(base); a[65535 - 40:] = a[1000:1040]; cases.append(bytes(a))
    a = bytearray(base); a[65535 - 9:] = a[1000:1009]; a[65535 - 300:65535 - 291] = a[1000:1009]; cases.append(bytes(a))
    cases.append(bytes(b"abcabcabd" * 7000)[:65535])
    cases.append(corpus.text_like(77, 65535, vocab=64))
    cases.append(bytes(corpus.text_like(78, 30000)) + corpus.random_bytes(79, 20000) + bytes(15535))
    for idx, data in enumerate(cases):
        for seg, seed in ((128, 1), (128, 2), (32, 3)):
            assert emu.emu_lazy_check(data, len(data), seg, seed, None) == 0, (idx, seg, seed)


def test_plan_split_matches_plan_block_on_histograms(emu):
    """csrc/zwz_plan.hip's decomposition of zlib's block flush (heap a lane per tree, depths by pointer jumping over the
    merges, capped lengths + overflow repair, codes by rank, code-length runs one by one) against huff_core.h's plan_block on
    histograms no corpus produces: Fibonacci-like counts (trees deeper than 15 -> gen_bitlen's repair), all-equal counts
    (every tie-break is the heap's), one or two used symbols (forced nodes), long runs of equal lengths."""
    import numpy as np
    rs = np.random.RandomState(31337)
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])

    def check(lf, df, stored_len=60000, stored_ok=1, last=1):
        lf = np.asarray(lf, dtype=np.uint16).copy(); df = np.asarray(df, dtype=np.uint16).copy()
        lf[256] = 1
        assert lf.size == 286 and df.size == 30
        rc = emu.emu_plan_split_check(lf.ctypes.data, df.ctypes.data, stored_len, stored_ok, last)
        assert rc == 0, (rc, lf.tolist(), df.tolist())

    z286, z30 = np.zeros(286, np.int64), np.zeros(30, np.int64)
    hits0 = emu.emu_overflow_hits()
    check(z286, z30, stored_len=0)                                   # an empty block: EOB alone, both trees forced
    for sym in (0, 1, 2, 97, 255, 285):
        a = z286.copy(); a[sym] = 5; check(a, z30)
    for d in (0, 1, 2, 29):
        a = z286.copy(); a[65:91] = 3; b = z30.copy(); b[d] = 7; check(a, b)
    for k in (3, 8, 17, 20, 22):                                      # skewed: deeper than 15 from k = 17 on (with the EOB's count of 1, fib[1:] makes the chain)
        a = z286.copy(); a[10:10 + k] = fib[1:k + 1]; check(a, z30)
        a = z286.copy(); a[np.arange(k) * 7] = fib[:k][::-1]; b = z30.copy(); b[:min(k, 20)] = fib[:min(k, 20)]; check(a, b)
    for c in (1, 2, 57, 58):                                          # all counts equal: full, ragged
        check(np.full(286, c), np.full(30, c))
        a = z286.copy(); a[:200:3] = c; check(a, z30)
    for _ in range(400):
        used = rs.randint(1, 287)
        a = z286.copy()
        idx = rs.choice(286, used, replace=False)
        shape = rs.randint(0, 4)
        vals = [rs.randint(1, 4, used), rs.randint(1, 60, used), (rs.pareto(1.1, used) * 3 + 1).astype(np.int64), rs.geometric(0.02, used)][shape]
        a[idx] = np.minimum(vals, 16383 // max(used, 1) + 1)
        b = z30.copy()
        nd = rs.randint(0, 31)
        if nd:
            b[rs.choice(30, nd, replace=False)] = np.minimum((rs.pareto(1.0, nd) * 2 + 1).astype(np.int64), 500)
        check(a, b, stored_len=int(rs.randint(0, 65536)), stored_ok=int(rs.randint(0, 2)), last=int(rs.randint(0, 2)))
    assert emu.emu_overflow_hits() - hits0 >= 6          # the over-long trees (literal and distance) really went through the repair


def test_deep_distance_tree_chunks_reach_the_repair(emu, oracle):
    """corpus.deep_distance_tree: real chunks on which zlib's gen_bitlen must shorten an over-long distance tree -- the corner
    tests/test_gpu_codec.py::test_plan_stage_skewed_histograms sends through the device's plan stage.  Here: the repair is
    really reached (the split form counts it), and the cores give the oracle's stream."""
    for seed in (2, 3, 4):
        data = corpus.deep_distance_tree(seed)
        h0 = emu.emu_overflow_hits()
        assert emu_binding.chunk_stream(emu, data) == oracle.deflate6(data)
        assert emu.emu_overflow_hits() - h0 >= 1, seed


def test_packed_window_decode_agrees_with_the_classic_decoder(emu):
    """inflate's fast tables in their packed form (csrc/inflate_core.h: an entry = kind + bits to skip) and the two things the kernel does
    with them -- a slot's (kind, bits) and the once-per-round value pass -- against decode_symbol on the classic tables, at every symbol
    start of streams of every kind and level, whole, cut and damaged (decompress_chunk(), decompression.cpp:11-37, ignores zlib's
    return codes: whatever precedes the damage must decode the same)."""
    import ctypes
    total = slow = 0
    rs = corpus.splitmix64(4242, 64)
    for i, kind in enumerate(corpus.KINDS):
        for j, n in enumerate((300, 5000, 65535)):
            data = corpus.make(kind, 5100 + 7 * i + j, n)
            for level in (6, 1, 9):
                z = zlib.compress(data, level)[:65535]
                variants = [z, z[:len(z) * 2 // 3]]
                b = bytearray(z); b[len(b) // 2] ^= 1 << int(rs[(i + j) % 64] % 8); variants.append(bytes(b))
                for v in variants:
                    a, c = ctypes.c_uint64(), ctypes.c_uint64()
                    assert emu.emu_packed_window_check(v, len(v), ctypes.byref(a), ctypes.byref(c)) == 0, (kind, n, level, len(v))
                    total += a.value; slow += c.value
    assert total > 200000 and 0 < slow < total // 20


def test_which_chunks_go_to_sort_and_band(emu):
    """lz_dense_list's rule (csrc/lz_band.h: sample_is_dense) pinned on the corpora it was measured on (tools/exp/dense_crossover.sh, DESIGN.md section 4
    round 5): text-like chunks from 4 KB up and full image-like chunks are chain-heavy (sort + band is 1.4 - 3.5x faster there), incompressible chunks
    and the 7 KB image-like files of BASELINE configs[3] are not (the chain walk is 1.5x faster).  Either path gives the same records -- the GPU tests
    force each -- so this is about speed only, and a change of the rule should be a decision, not an accident."""
    import workloads
    def dense(data):
        return emu.emu_chunk_is_dense(data, len(data)) == 1
    for n in (4096, 8192, 16384, 32768, 65535):
        votes = [dense(corpus.text_like(900 + i, n)) for i in range(8)]
        assert all(votes), (n, votes)
    assert all(dense(corpus.gradient(77 + i, 65535)) for i in range(4))
    assert not any(dense(corpus.random_bytes(5 + i, 65535)) for i in range(4))
    sizes = workloads.small_file_sizes(300)
    small = [workloads.small_file_bytes(i, sizes[i]) for i in range(300) if 2000 <= sizes[i] <= 9000]
    assert len(small) > 100 and sum(dense(c) for c in small) == 0
    big = [workloads.small_file_bytes(1000 + i, 48000) for i in range(6)]           # the same generator at 48 KB: the band's side of the crossing
    assert all(dense(c) for c in big)
