"""Synthetic BASELINE workloads as device-resident chunk batches (bench.py and the full-batch GPU tests).

Every byte comes from tests/corpus.py's own PRNG (splitmix64), so the GPU batch, the files the CPU baseline compresses and
the chunks the oracle checks are the same bytes on every box and torch version.  The incompressible
workload is generated on the device: splitmix64 restated in wrapping int64 arithmetic, bit-identical to
corpus.random_bytes(seed, n) (checked by tests/test_workloads.py on the CPU).

A batch is the reference's chunking of n_files files (compression.cpp:52-64): 65 535-byte reads until a short one, each
chunk in its own 65 536-byte slot.
"""
import os
import numpy as np

import corpus

CHUNK, STRIDE = 65535, 65536
RANDOM_SEED0, TEXT_SEED0, SMALL_SEED0 = 1_000_000, 2_000_000, 3_000_000   # file i of rank r: seed0 + r * 100 000 + i


def _i64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(torch, z, k):           # logical shift right of int64
    return (z >> k) & ((1 << (64 - k)) - 1)


def splitmix64_device(torch, seeds, n, dev):
    """corpus.splitmix64(seed, n) for every seed of the int64 tensor `seeds` -> (len(seeds), n) int64 (bit patterns)."""
    i = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
    z = seeds.to(dev).view(-1, 1) * _i64(0xD1342543DE82EF95) + i.view(1, -1) * _i64(0x9E3779B97F4A7C15)
    z = (z ^ _lsr(torch, z, 30)) * _i64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(torch, z, 27)) * _i64(0x94D049BB133111EB)
    return z ^ _lsr(torch, z, 31)


def random_files_device(torch, seeds, file_bytes, dev):
    """(len(seeds), file_bytes) uint8: corpus.random_bytes(seed, file_bytes) per row."""
    z = splitmix64_device(torch, seeds, (file_bytes + 7) // 8, dev)
    return z.view(torch.uint8).view(len(seeds), -1)[:, :file_bytes]


def chunk_layout(file_sizes):
    """Per-chunk lengths of files cut the reference's way, and each file's first chunk."""
    sizes = np.asarray(file_sizes, dtype=np.int64)
    nchunks = sizes // CHUNK + 1                      # a short (possibly empty) read ends the file
    first = np.concatenate([[0], np.cumsum(nchunks)[:-1]])
    total = int(nchunks.sum())
    lens = np.full(total, CHUNK, dtype=np.int64)
    lens[first + nchunks - 1] = sizes % CHUNK
    return lens, first, nchunks


def _place_equal_files(torch, d_in, rows, f0, file_bytes):
    """rows: (k, file_bytes) uint8 device tensor -> slots of files f0..f0+k-1 (all files have the same chunk count)."""
    full, tail = divmod(file_bytes, CHUNK)
    per_file = full + 1
    view = d_in.view(-1, per_file, STRIDE)
    k = rows.shape[0]
    for c in range(full):
        view[f0:f0 + k, c, :CHUNK] = rows[:, c * CHUNK:(c + 1) * CHUNK]
    if tail:
        view[f0:f0 + k, full, :tail] = rows[:, full * CHUNK:]


def build_equal_files(torch, dev, workload, n_files, file_bytes, rank=0, distinct_text=256):
    """BASELINE configs 1-3: n_files files of file_bytes each.  -> (d_in, d_off, d_len, n_chunks, raw_bytes, host_file)
    host_file(i) returns file i's bytes on the host (for the oracle sample and the CPU baseline)."""
    full, tail = divmod(file_bytes, CHUNK)
    per_file = full + 1
    n = n_files * per_file
    d_in = torch.zeros(n * STRIDE, dtype=torch.uint8, device=dev)
    lens = np.full((n_files, per_file), CHUNK, dtype=np.int32)
    lens[:, -1] = tail
    if workload == "random":
        seed0 = RANDOM_SEED0 + rank * 100_000
        step = 256
        for f0 in range(0, n_files, step):
            f1 = min(n_files, f0 + step)
            seeds = torch.arange(seed0 + f0, seed0 + f1, dtype=torch.int64)
            _place_equal_files(torch, d_in, random_files_device(torch, seeds, file_bytes, dev), f0, file_bytes)

        def host_file(i):
            return corpus.random_bytes(seed0 + i, file_bytes)
    elif workload == "text":
        seed0 = TEXT_SEED0 + rank * 100_000
        distinct = min(n_files, distinct_text)       # text generation is host work: the batch repeats `distinct` files
        host = np.zeros((distinct, file_bytes), dtype=np.uint8)
        for i in range(distinct):
            host[i] = np.frombuffer(corpus.text_like(seed0 + i, file_bytes), dtype=np.uint8)
        t = torch.from_numpy(host).to(dev)
        for f0 in range(0, n_files, distinct):
            f1 = min(n_files, f0 + distinct)
            _place_equal_files(torch, d_in, t[:f1 - f0], f0, file_bytes)

        def host_file(i):
            return host[i % distinct].tobytes()
    else:
        raise ValueError(workload)
    d_len = torch.from_numpy(lens.reshape(-1)).to(dev)
    d_off = torch.arange(n, dtype=torch.int64, device=dev) * STRIDE
    return d_in, d_off, d_len, n, int(n_files) * int(file_bytes), host_file


def small_file_sizes(n_files, seed=4):
    """BASELINE config 4 (reference README.md:12-13: ~370 000 images, ~2.5 GB): log-normal sizes, mean ~6.8 KB."""
    u = (corpus.splitmix64(seed, 2 * n_files) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    g = np.sqrt(-2.0 * np.log(np.maximum(u[0::2], 1e-300))) * np.cos(2.0 * np.pi * u[1::2])      # Box-Muller
    sigma = 0.9
    mean = float(os.environ.get("ZWZ_SMALL_MEAN", "6800"))      # (tools/exp/dense_crossover.sh varies it; the workload is the default)
    sizes = np.exp(np.log(mean) - sigma * sigma / 2.0 + sigma * g)
    return np.clip(sizes, 16, 400_000).astype(np.int64)


def small_file_bytes(i, size, rank=0):
    """File i of the small-files workload: "image-like" smooth ramp + noise (corpus.gradient)."""
    return corpus.gradient(SMALL_SEED0 + rank * 1_000_000 + i, int(size))


def build_small_files(torch, dev, n_files, rank=0):
    """BASELINE config 4 as a device batch (one chunk per file almost always).  -> as build_equal_files."""
    sizes = small_file_sizes(n_files)
    lens, first, nchunks = chunk_layout(sizes)
    n = len(lens)
    d_in = torch.zeros(n * STRIDE, dtype=torch.uint8, device=dev)
    view = d_in.view(n, STRIDE)
    # corpus.gradient restated on the device, a block of files at a time: ((x // 7 + (x % 251) // 3 + noise) & 255),
    # noise = splitmix64(seed, n)[x] % 5 - 2 (unsigned modulo: 2**64 % 5 == 1)
    step = 1024
    for f0 in range(0, n_files, step):
        f1 = min(n_files, f0 + step)
        width = int(sizes[f0:f1].max())
        seeds = torch.arange(SMALL_SEED0 + rank * 1_000_000 + f0, SMALL_SEED0 + rank * 1_000_000 + f1, dtype=torch.int64)
        z = splitmix64_device(torch, seeds, width, dev)
        noise = (torch.remainder(z, 5) + (z < 0).to(torch.int64)) % 5 - 2
        x = torch.arange(width, dtype=torch.int64, device=dev).view(1, -1)
        img = ((x // 7 + (x % 251) // 3 + noise) & 255).to(torch.uint8)
        one = np.nonzero(nchunks[f0:f1] == 1)[0]     # one chunk per file: placed in one indexed copy (bytes past a file's
        if len(one):                                 # size are junk the length array hides, as the API allows)
            w = min(width, CHUNK)
            view[torch.from_numpy(first[f0:f1][one]).to(dev), :w] = img[torch.from_numpy(one).to(dev), :w]
        for k in np.nonzero(nchunks[f0:f1] > 1)[0]:  # (files longer than one chunk are rare: per-file placement)
            k = int(k)
            size, c0 = int(sizes[f0 + k]), int(first[f0 + k])
            for c in range(int(nchunks[f0 + k])):
                a, b = c * CHUNK, min(size, (c + 1) * CHUNK)
                if b > a:
                    view[c0 + c, :b - a] = img[k, a:b]
    d_len = torch.from_numpy(lens.astype(np.int32)).to(dev)
    d_off = torch.arange(n, dtype=torch.int64, device=dev) * STRIDE
    return d_in, d_off, d_len, n, int(sizes.sum()), lambda i: small_file_bytes(i, sizes[i], rank)


def text_rows_device(torch, seeds, n_bytes, dev, vocab_seed=TEXT_SEED0, vocab=4096):
    """corpus.text_like restated on the device for many rows at once: row r = the first n_bytes bytes of Zipf(1/rank) words over ONE
    vocabulary (vocab_seed's: the words corpus.text_like(vocab_seed, .) draws from), the sequence of words drawn from seeds[r] -- every
    row a different text of the same language.  -> (len(seeds), n_bytes) uint8.  (bench.py --workload one_file: distinct text per record,
    where rounds 3-4 tiled 1 024 rows.)"""
    wl = (corpus.splitmix64(vocab_seed ^ 0x1111, vocab) % np.uint64(7)).astype(np.int64) + 2
    letters = (corpus.splitmix64(vocab_seed ^ 0x2222, int(wl.sum())) % np.uint64(26)).astype(np.uint8) + 97
    starts = np.concatenate([[0], np.cumsum(wl)[:-1]])
    p = 1.0 / np.arange(1, vocab + 1)
    cdf = np.cumsum(p / p.sum())
    t_wl = torch.from_numpy(wl).to(dev)
    t_letters = torch.from_numpy(letters).to(dev)
    t_starts = torch.from_numpy(starts).to(dev)
    t_cdf = torch.from_numpy(cdf).to(dev)
    k = len(seeds)
    nwords = n_bytes // 3 + 8                                   # words are >= 3 bytes with their space
    z = splitmix64_device(torch, torch.as_tensor(seeds, dtype=torch.int64) ^ 0x3333, nwords, dev)
    u = _lsr(torch, z, 11).to(torch.float64) / float(1 << 53)
    idx = torch.clamp(torch.searchsorted(t_cdf, u.reshape(-1)).view(k, nwords), max=vocab - 1)
    del z, u
    wlen = t_wl[idx]
    pos = torch.cumsum(wlen + 1, dim=1) - (wlen + 1)            # where each word starts in its row
    out = torch.full((k, n_bytes + 16), 32, dtype=torch.uint8, device=dev)
    rows = torch.arange(k, device=dev).view(-1, 1).expand(k, nwords)
    for j in range(8):
        m = (wlen > j) & (pos + j < n_bytes)
        out[rows[m], (pos + j)[m]] = t_letters[(t_starts[idx] + j)[m]]
    return out[:, :n_bytes]
