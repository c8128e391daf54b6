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
        for tile, fmt in ((5632, -1), (5632, 0), (192, -1), (64, 0)):
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
        for tile, fmt in ((5632, -1), (5632, 0), (1024, -1)):
            got = _band(emu, data, tile, fmt)
            assert got[0] == want[0] and got[1] == want[1], (idx, tile, fmt)
