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
