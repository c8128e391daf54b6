"""The device-side generators of tests/workloads.py produce corpus.py's bytes (checked here on the CPU device)."""
import numpy as np
import pytest

import corpus
import workloads


def test_device_splitmix_is_corpus_splitmix():
    torch = pytest.importorskip("torch")
    dev = torch.device("cpu")
    seeds = torch.tensor([0, 1, 12345, workloads.RANDOM_SEED0 + 77, (1 << 40) + 3], dtype=torch.int64)
    rows = workloads.random_files_device(torch, seeds, 1001, dev)
    for s, row in zip(seeds.tolist(), rows):
        assert row.numpy().tobytes() == corpus.random_bytes(s, 1001)


def test_equal_files_batch_layout():
    torch = pytest.importorskip("torch")
    dev = torch.device("cpu")
    for workload in ("random", "text"):
        d_in, d_off, d_len, n, raw, host_file = workloads.build_equal_files(torch, dev, workload, 5, 140000, distinct_text=3)
        assert n == 5 * 3 and raw == 5 * 140000
        lens = d_len.numpy().tolist()
        assert lens == [65535, 65535, 140000 - 2 * 65535] * 5
        slots = d_in.view(n, workloads.STRIDE).numpy()
        for f in range(5):
            data = host_file(f)
            got = b"".join(slots[3 * f + c, :lens[3 * f + c]].tobytes() for c in range(3))
            assert got == data and len(data) == 140000


def test_small_files_batch_layout():
    torch = pytest.importorskip("torch")
    dev = torch.device("cpu")
    n_files = 300
    d_in, d_off, d_len, n, raw, host_file = workloads.build_small_files(torch, dev, n_files)
    sizes = workloads.small_file_sizes(n_files)
    lens, first, nchunks = workloads.chunk_layout(sizes)
    assert n == len(lens) and raw == int(sizes.sum()) and 3000 < sizes.mean() < 12000
    slots = d_in.view(n, workloads.STRIDE).numpy()
    for f in (0, 1, 17, 123, 299, int(np.argmax(sizes))):
        got = b"".join(slots[first[f] + c, :lens[first[f] + c]].tobytes() for c in range(nchunks[f]))
        assert got == host_file(f)
