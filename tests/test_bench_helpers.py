"""bench.py's CPU-side helpers (no GPU): the reference-binary baseline on a small tree, the guarded time-out, the top-level text fields."""
import importlib.util
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("zwz_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def test_cpu_baseline_tree_times_the_reference_on_a_small_tree(bench):
    """The baseline `--workload small_files | one_file` put on the line: the reference binary itself on a bounded sample; without
    the binary (the GPU box of a checkout that never built it) the entry says so instead of substituting anything."""
    import corpus

    def populate(src):
        total = 0
        for i, n in enumerate((70000, 1000, 0, 131070)):
            b = corpus.text_like(900 + i, n)
            total += len(b)
            with open(os.path.join(src, "f%d.txt" % i), "wb") as f:
                f.write(b)
        return total

    r = bench.cpu_baseline_tree(populate, "four small files", [1, 2])
    assert r["kind"] == "reference" and r["unit"] == "GB/s" and r["sample"] == "four small files"
    if os.path.exists(bench.REF_MAIN):
        assert r["value"] > 0 and r["compress_GBps"] > 0 and r["decompress_GBps"] > 0 and r["ranks"]["1"]["bytes"] == 202070
        assert r["ranks"]["1"]["compress_banner_s"] is not None          # the reference's own "Time Taken" was found on its stdout
        if "2" in r["ranks"] and "error" not in r["ranks"]["2"]:
            assert r["ranks"]["2"]["compress_banner_s"] is not None      # ... and with two ranks sharing a stdout: from rank 0's own file
    else:
        assert r["value"] is None and "absent" in r["note"]


def test_run_timed_kills_the_whole_process_group(bench):
    """A reference run that overstays is killed with everything it started (mpiexec's ranks hold the stdout pipe: killing the launcher alone left
    read() waiting) and reported as a failed command, not as a measurement."""
    t0 = time.perf_counter()
    with pytest.raises(subprocess.CalledProcessError):
        bench._run_timed(["bash", "-c", "sleep 30 & sleep 30"], limit_s=0.5)
    assert time.perf_counter() - t0 < 10
