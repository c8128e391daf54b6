"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/zwz.h declares; host helpers that need no GPU behave like the reference's; without a GPU
the product fails loudly instead of falling back."""
import ctypes
import hashlib
import importlib
import os
import re

import pytest

import corpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def zwz():
    import __graft_entry__ as g
    z = importlib.import_module(g.PKG)
    if not os.path.exists(z.LIB_PATH):
        g.build()
    return z


def test_every_declared_symbol_is_exported(zwz):
    header = open(os.path.join(ROOT, "include", "zwz.h")).read()
    names = set(re.findall(r"\b(zwz_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 18
    L = ctypes.CDLL(zwz.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_no_gpu_means_loud_failure_not_fallback(zwz):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(zwz.ZwzError):
        zwz.Codec(0)


def test_product_does_not_link_the_oracle_or_zlib(zwz):
    import subprocess
    needed = subprocess.run(["readelf", "-d", zwz.LIB_PATH], capture_output=True, text=True).stdout
    assert "libz.so" not in needed and "oracle" not in needed          # direct dependencies only
    undef = subprocess.run(["nm", "-D", "--undefined-only", zwz.LIB_PATH], capture_output=True, text=True).stdout
    for sym in ("deflate", "inflate", "zo_", "MD5_"):
        assert not re.search(r"\b%s" % sym, undef), sym


def test_host_helpers_match_reference_semantics(zwz, tmp_path):
    src = tmp_path / "top" / "src"
    files = corpus.golden_tree()
    for rel, data in files.items():
        p = src / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(data)
    rec = zwz.sort_files_by_size(str(src))
    assert rec == str(tmp_path / "top" / "sorted_files_by_size.txt")      # file_sort.cpp:33
    lines = open(rec).read().split("\n")
    assert lines[-1] == "" and sorted(lines[:-1]) == sorted(files)
    sizes = [len(files[l]) for l in lines[:-1]]
    assert sizes == sorted(sizes, reverse=True)                           # file_sort.cpp:30
    assert zwz.count_non_empty_lines(rec) == len(files)
    (tmp_path / "blank.txt").write_text("a\n\n  \t\nb\n")
    assert zwz.count_non_empty_lines(str(tmp_path / "blank.txt")) == 2    # file_tools.cpp:17
    assert zwz.count_non_empty_lines(str(tmp_path / "missing.txt")) == -1  # file_tools.cpp:8-11
    for rel in files:
        assert zwz.md5_of_file(str(src / rel)) == hashlib.md5(files[rel]).hexdigest()
    for n in [55, 56, 57, 63, 64, 65, 119, 120, 121, (1 << 20) + 3]:
        (tmp_path / "m.bin").write_bytes(corpus.random_bytes(n, n))
        assert zwz.md5_of_file(str(tmp_path / "m.bin")) == hashlib.md5(corpus.random_bytes(n, n)).hexdigest()
