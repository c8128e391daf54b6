#!/usr/bin/env python3
"""bench.py -- compress + decompress throughput of the chunk codec on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch resident in HBM: deflate every <= 65535-byte chunk of a synthetic
directory image, then inflate the resulting payloads.  `value` = raw bytes / (deflate + inflate time), whole job over
all ranks (max over ranks of the timed region).  Ranks own disjoint files (shards partition one-per-rank, SURVEY.md
section 8e): weak scaling, no data-path collective.

With no --workload the run covers BOTH headline configurations and prints ONE JSON line:
  top level   BASELINE configs[2]: 10 000 x 256 KiB text-like files (50 000 chunks; the LZ77 + Huffman kernels) -- the line's `value`
  "random"    BASELINE configs[1]: 10 000 x 256 KiB incompressible files (stored-block path), same fields
  "e2e"       the drop-in CLI (`main compress` + `main decompress`) on the same two directories and on a 4x larger text one: banner seconds,
              GB/s, steady state / fixed cost -- what a user of the reference's command line sees; never `value`
Other workloads on request: --workload small_files (configs[3]: ~370 000 image-like files, ~2.5 GB) and
--workload one_file --decompress-only (configs[4] scaled to fit: ONE file, its records split over the ranks).

Every line also carries
  roofline      dominant kernel's (raw + payload) bytes per launch / its mean launch duration, measured with HIP events on
                the codec's own stream, against the 8 TB/s HBM peak; plus whole-direction fractions against 8.0 and 6.29 TB/s
  verified      outside the timed region: every decoded byte compared with the input on the device (not lengths), and
                >= 512 sampled chunks per workload compared with the CPU oracle (payload bytes and decoded bytes)
  cpu_baseline  the REFERENCE binary (oracle/_ref/main) on this box's host cores, 1/2/4/8 MPI ranks, on a bounded sample of
                the same files (same PRNG, same seeds) and on BASELINE configs[0]'s shape (100 x 1 MiB)

`--gpus N` without a launcher starts N ranks itself (fresh child processes, before this process touches the GPU); under a
launcher (WORLD_SIZE set) a mismatch between --gpus and WORLD_SIZE is an error.
"""
import argparse
import importlib
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHUNK, STRIDE = 65535, 65536
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0    # ... 6.29 TB/s measured-achievable
MPIEXEC = "/opt/conda/bin/mpiexec"
REF_MAIN = os.path.join(ROOT, "oracle", "_ref", "main")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="both", choices=["both", "random", "text", "small_files", "one_file"])
    ap.add_argument("--files", type=int, default=None, help="files per GPU (default 10000; small_files: 370000)")
    ap.add_argument("--file-bytes", type=int, default=262144)
    ap.add_argument("--one-file-bytes", type=int, default=8 << 30, help="one_file: size of the single file (BASELINE: 64 GiB)")
    ap.add_argument("--decompress-only", action="store_true", help="time inflate only (one_file: BASELINE configs[4])")
    ap.add_argument("--max-batch", type=int, default=51200)   # whole config-2 batch in one launch per kernel: ~48 GB of workspace, sized for 288 GB of HBM
    ap.add_argument("--oracle-sample", type=int, default=512)
    ap.add_argument("--cpu-sample-files", type=int, default=400)   # x 256 KiB = 105 MB: ~4 s of reference CPU work at 1 rank
    ap.add_argument("--cpu-sample-small-files", type=int, default=40000, help="small_files: files of the CPU baseline's sample (~270 MB: ~6 s at 1 rank)")
    ap.add_argument("--cpu-sample-one-file-bytes", type=int, default=512 << 20, help="one_file: bytes of the CPU baseline's single file (~12 s to compress at 1 rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end legs (the CLI `main compress|decompress` on the same directories)")
    ap.add_argument("--e2e-scale", type=int, default=4, help="also run the text directory this many times larger (steady state visible); 0: skip")
    ap.add_argument("--e2e-tmp", default=None, help="where the e2e directories are built (default: $ZWZ_E2E_TMP or /tmp)")
    ap.add_argument("--cpu-baseline-port", action="store_true", help="time the oracle restatement if the reference binary is absent (labelled kind=port)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 ranks share cuda:0 over gloo: exercises the multi-rank code path on a 1-GPU box (not a measurement)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(args):
    """--gpus N with no launcher around us: start N ranks as children (this process has not touched the GPU) and
    return the launcher's exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_facts(path):
    facts = {"cpu_model": None, "logical_cpus": os.cpu_count(), "physical_cores": None, "sockets": None, "filesystem": None}
    try:
        model, cores = None, set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        facts["cpu_model"] = model
        facts["physical_cores"] = len(cores) or None
        facts["sockets"] = len({p for p, _ in cores}) or None
    except OSError:
        pass
    try:
        best = ""
        for line in open("/proc/mounts"):
            parts = line.split()
            if len(parts) >= 3 and (path == parts[1] or path.startswith(parts[1].rstrip("/") + "/")) and len(parts[1]) >= len(best):
                best, facts["filesystem"] = parts[1], parts[2]
    except OSError:
        pass
    try:
        facts["cpus_allowed"] = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    return facts


def _run_timed(cmd, limit_s=900, banner_file=None):
    """Run a command to its end; wall time between the spawn and the child's exit as wait() sees it (no polling: a
    subprocess.run(timeout=...) wakes every 50 ms and quantised every figure of round 2), and the interval the reference
    prints itself (main.cpp:148-155, MPI_Wtime around the operation -- what BASELINE.md section 3 asks for)."""
    import re
    import threading
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, start_new_session=True)

    def kill_group():               # mpiexec's ranks hold the stdout pipe: killing the launcher alone would leave read() waiting for them
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
    killer = threading.Timer(limit_s, kill_group)
    killer.start()
    try:
        out = p.stdout.read()          # (returns at EOF: the child and its ranks have closed stdout)
        rc = p.wait()
    finally:
        killer.cancel()
    wall = time.perf_counter() - t0
    if rc != 0:
        raise subprocess.CalledProcessError(rc, cmd)
    if banner_file is not None:          # several MPI ranks: rank 0's own stdout (mpiexec -outfile-pattern), where no other rank's lines cut into the banner
        try:
            out = open(banner_file, "rb").read()
        except OSError:
            pass
    m = re.findall(rb"Time Taken: ([0-9.eE+-]+) seconds", out)
    return wall, (float(m[-1]) if m else None)


def _time_reference(src, work, total_bytes, ranks):
    """The reference's CLI (README.md:59) as `ranks` MPI ranks; decompression is one process, a thread per shard
    (decompression.cpp:174).  Warm page cache.  *_s is the wall clock around the whole command (process start-up, MPI
    bootstrap and the file sort included), *_banner_s the reference's own "Time Taken"; the GB/s figures use the wall clock."""
    dst, back = os.path.join(work, "zwz%d" % ranks), os.path.join(work, "back%d" % ranks)
    # (K ranks write one stdout: their lines used to cut into rank 0's banner and the "Time Taken" regex lost it -- every rank gets its own file)
    pat = os.path.join(work, "ref_out_%d" % ranks)
    cmd = [REF_MAIN] if ranks == 1 else [MPIEXEC, "-n", str(ranks), "-outfile-pattern", pat + ".%r", REF_MAIN]
    tc, bc = _run_timed(cmd + ["compress", src, dst], banner_file=(pat + ".0") if ranks > 1 else None)
    td, bd = _run_timed([REF_MAIN, "decompress", dst, back])
    shutil.rmtree(dst, ignore_errors=True)
    shutil.rmtree(back, ignore_errors=True)
    return {"bytes": total_bytes, "compress_s": round(tc, 4), "decompress_s": round(td, 4), "compress_banner_s": bc, "decompress_banner_s": bd,
            "compress_GBps": round(total_bytes / tc / 1e9, 5), "decompress_GBps": round(total_bytes / td / 1e9, 5),
            "roundtrip_GBps": round(total_bytes / (tc + td) / 1e9, 5)}


def cpu_baseline(workload, host_file, n_files, file_bytes, allow_port):
    """Reference CPU path on a bounded sample: the first n_files files of the GPU workload (same bytes)."""
    import corpus
    work = tempfile.mkdtemp(prefix="zwz_cpu_")
    try:
        facts = _cpu_facts(work)
        src = os.path.join(work, "data", "src")
        os.makedirs(src)
        for i in range(n_files):
            with open(os.path.join(src, "f%05d.bin" % i), "wb") as f:
                f.write(host_file(i))
        total = n_files * file_bytes
        sample = "first %d files of the workload (%d x %d B %s, same PRNG and seeds), warm page cache" % (n_files, n_files, file_bytes, workload)
        if not os.path.exists(REF_MAIN):
            if not allow_port:
                return {"value": None, "unit": "GB/s", "cores": 0, "kind": "reference", "sample": sample, "host": facts,
                        "note": "oracle/_ref/main is absent on this box (it is built from /root/reference by `make -C oracle ref` and is "
                                "git-ignored); no substitute was timed -- pass --cpu-baseline-port to time the oracle restatement instead"}
            import oracle_binding
            o = oracle_binding.load()
            rec = os.path.join(work, "rec.txt")
            with open(rec, "w") as f:
                f.write("".join("f%05d.bin\n" % i for i in range(n_files)))
            dst, back = os.path.join(work, "zwz"), os.path.join(work, "back")
            os.makedirs(dst)
            os.makedirs(back)
            t0 = time.perf_counter()
            o.compress_shard(src, dst, rec, 0, 1)
            t1 = time.perf_counter()
            o.decompress_shard(os.path.join(dst, "compressed_0.zwz"), back)
            t2 = time.perf_counter()
            return {"value": round(total / (t2 - t0) / 1e9, 5), "unit": "GB/s", "cores": 1, "kind": "port", "sample": sample, "host": facts,
                    "compress_GBps": round(total / (t1 - t0) / 1e9, 5), "decompress_GBps": round(total / (t2 - t1) / 1e9, 5)}
        # K = 1, 2, 4, 8 and the box's physical cores.  The sample grows with K (n_files per rank, capped)
        # so that no leg shrinks to a few tenths of a second; every entry states its own byte count.
        ranks = {}
        ks = [1]
        if os.path.exists(MPIEXEC):
            ks = [1, 2, 4, 8]
            pc = facts.get("physical_cores") or 0
            if pc > 8 and pc <= (facts.get("cpus_allowed") or pc):
                ks.append(pc)
        have = n_files
        for k in ks:
            want = min(n_files * k, 8 * n_files if k <= 8 else 16 * n_files)
            for i in range(have, want):
                with open(os.path.join(src, "f%05d.bin" % i), "wb") as f:
                    f.write(host_file(i))
            have = max(have, want)
            try:
                ranks[str(k)] = _time_reference(src, work, have * file_bytes, k)
            except (subprocess.CalledProcessError, OSError) as e:
                ranks[str(k)] = {"error": str(e)}
        sample = "first %d x K files of the workload for K ranks, at most %d (%d B %s files, same PRNG and seeds), warm page cache" % (n_files, have, file_bytes, workload)
        r1 = ranks.get("1", {})
        if "error" in r1 or not r1:
            return {"value": None, "unit": "GB/s", "cores": 2, "kind": "reference", "sample": sample, "host": facts, "ranks": ranks,
                    "note": "the reference's 1-rank run failed on this box: %s" % r1.get("error")}
        out = {"value": r1["roundtrip_GBps"], "unit": "GB/s", "cores": 2, "kind": "reference", "sample": sample, "host": facts,
               "compress_GBps": r1["compress_GBps"], "decompress_GBps": r1["decompress_GBps"],
               "cores_note": "value = 1 MPI rank = producer + consumer thread (compression.cpp:162); ranks[K] uses 2K threads to compress and K threads (one per shard) to decompress",
               "timer": "wall clock from spawn to the child's exit (Popen.wait, no polling); *_banner_s = the reference's own 'Time Taken' (main.cpp:148-155)",
               "ranks": ranks}
        if not os.path.exists(MPIEXEC):
            out["ranks_note"] = "no mpiexec on this box: the reference runs as an MPI singleton only"
        if workload == "random":           # BASELINE configs[0] as such: 100 x 1 MiB random files, 1 rank
            shutil.rmtree(src)
            os.makedirs(src)
            for i in range(100):
                with open(os.path.join(src, "m%03d.bin" % i), "wb") as f:
                    f.write(corpus.random_bytes(500_000 + i, 1 << 20))
            out["config1_100x1MiB"] = _time_reference(src, work, 100 << 20, 1)
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cpu_baseline_tree(populate, sample, ks):
    """Reference CPU path on a bounded sample that is not n equal files (small_files: a directory of image-like files;
    one_file: ONE text-like file -> one shard, which the reference decodes on one thread, decompression.cpp:165-178).
    populate(src) writes the sample and returns its byte count."""
    work = tempfile.mkdtemp(prefix="zwz_cpu_")
    try:
        facts = _cpu_facts(work)
        src = os.path.join(work, "data", "src")
        os.makedirs(src)
        total = populate(src)
        if not os.path.exists(REF_MAIN):
            return {"value": None, "unit": "GB/s", "cores": 0, "kind": "reference", "sample": sample, "host": facts,
                    "note": "oracle/_ref/main is absent on this box (built from /root/reference by `make -C oracle ref`, git-ignored); no substitute was timed"}
        ranks = {}
        for k in ([1] + [k for k in ks if k != 1 and os.path.exists(MPIEXEC)]):
            try:
                ranks[str(k)] = _time_reference(src, work, total, k)
            except (subprocess.CalledProcessError, OSError) as e:
                ranks[str(k)] = {"error": str(e)}
        r1 = ranks["1"]
        if "error" in r1:
            return {"value": None, "unit": "GB/s", "cores": 2, "kind": "reference", "sample": sample, "host": facts, "ranks": ranks,
                    "note": "the reference's 1-rank run failed on this box: %s" % r1["error"]}
        return {"value": r1["roundtrip_GBps"], "unit": "GB/s", "cores": 2, "kind": "reference", "sample": sample, "host": facts,
                "compress_GBps": r1["compress_GBps"], "decompress_GBps": r1["decompress_GBps"],
                "cores_note": "value = 1 MPI rank = producer + consumer thread to compress (compression.cpp:162), ONE thread per shard to decompress (decompression.cpp:174)",
                "timer": "wall clock from spawn to the child's exit; *_banner_s = the reference's own 'Time Taken' (main.cpp:148-155)", "ranks": ranks}
    finally:
        shutil.rmtree(work, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ end to end (the CLI)
def e2e_legs(args):
    """`main compress|decompress` on the two headline directories (+ the text one args.e2e_scale times larger): tools/e2e.py's measurement,
    summarised.  Failures of this leg never lose the GPU line: they come back as a note."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    out = {}
    try:
        import e2e as e2e_tool
        if args.e2e_tmp:
            os.environ["ZWZ_E2E_TMP"] = args.e2e_tmp
        n = args.files or 10000
        for nm in ("random", "text"):
            out[nm] = e2e_tool.summarize(e2e_tool.measure(nm, n, False))
        if args.e2e_scale and args.e2e_scale > 1:
            out["text_x%d" % args.e2e_scale] = e2e_tool.summarize(e2e_tool.measure("text", n * args.e2e_scale, False))
        # the same text directory on a memory file system, if the box has one: file creation there is not the overlay's (10 000 creates in
        # one directory cost `main decompress` 0.1 - 1.5 s on the box's overlay root, run to run)
        if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and not (args.e2e_tmp or os.environ.get("ZWZ_E2E_TMP", "")).startswith("/dev/shm"):
            keep = os.environ.get("ZWZ_E2E_TMP")
            os.environ["ZWZ_E2E_TMP"] = "/dev/shm"
            try:
                out["text_tmpfs"] = e2e_tool.summarize(e2e_tool.measure("text", n, False))
            finally:
                if keep is None:
                    os.environ.pop("ZWZ_E2E_TMP", None)
                else:
                    os.environ["ZWZ_E2E_TMP"] = keep
        out["note"] = ("the drop-in CLI end to end, one GPU, one process: raw bytes / the banner's seconds; steady_* = the pipelined slices alone, "
                       "fixed_s = process + HIP start-up, context self-tests, enumeration + sort, pinning, teardown (from a second, ZWZ_TIMELINE run)")
    except Exception as e:
        out["note"] = "e2e leg failed: %r" % (e,)
    return out


# ------------------------------------------------------------------------------------------------ verification
def verify_bytes(torch, d_in, d_len, d_olen, d_back, d_blen, d_stat):
    """Every chunk, on the device: the decoded bytes are the input's bytes.  A chunk either comes back whole, or it is one
    the reference truncates (payload at the 65535-byte cap, decoder ran out of input) and comes back as a PREFIX."""
    n = d_len.numel()
    ok_len = ((d_blen == d_len) | ((d_olen == CHUNK) & (d_stat == 1) & (d_blen < d_len)))
    a, b = d_in.view(n, STRIDE), d_back.view(n, STRIDE)
    col = torch.arange(STRIDE, device=d_in.device, dtype=torch.int32).view(1, -1)
    bad = 0
    for c0 in range(0, n, 2048):
        c1 = min(n, c0 + 2048)
        live = col < d_blen[c0:c1].view(-1, 1)
        bad += int(((a[c0:c1] != b[c0:c1]) & live).any(dim=1).sum().item())
    return {"chunks": n, "length_rule_ok": bool(ok_len.all().item()), "chunks_with_wrong_bytes": bad,
            "truncated_chunks": int((d_blen < d_len).sum().item())}


def verify_oracle_sample(torch, k, d_in, d_len, d_out, d_olen, d_back, d_blen):
    """k chunks spread over the batch: payload == oracle's payload, decoded bytes == oracle's decode of it."""
    import oracle_binding
    o = oracle_binding.load()
    n = d_len.numel()
    idx = sorted(set(int(i * (n - 1) / max(1, k - 1)) for i in range(min(k, n))))
    t = torch.tensor(idx, dtype=torch.int64, device=d_in.device)
    h_in = d_in.view(n, STRIDE)[t].cpu().numpy()
    h_out = d_out.view(n, STRIDE)[t].cpu().numpy()
    h_back = d_back.view(n, STRIDE)[t].cpu().numpy()
    lens, olens, blens = d_len[t].cpu().tolist(), d_olen[t].cpu().tolist(), d_blen[t].cpu().tolist()
    bad_payload = bad_decode = 0
    for j in range(len(idx)):
        chunk = h_in[j, :lens[j]].tobytes()
        want = o.payload(chunk)
        if h_out[j, :olens[j]].tobytes() != want:
            bad_payload += 1
        if h_back[j, :blens[j]].tobytes() != o.inflate(want, 70000)[0]:
            bad_decode += 1
    return {"sampled": len(idx), "payload_mismatches": bad_payload, "decode_mismatches": bad_decode}


# ------------------------------------------------------------------------------------------------ one workload
def run_workload(args, torch, dist, codec, dev, world, rank, name):
    import workloads
    cpu_dev = torch.device("cpu")
    decompress_only = args.decompress_only
    scaling = "weak"
    cpu_tree = None
    if name in ("random", "text"):
        n_files = args.files or 10000
        d_in, d_off, d_len, n, raw_bytes, host_file = workloads.build_equal_files(torch, dev, name, n_files, args.file_bytes, rank)
        desc = "%d x %d B %s files per GPU -> %d chunks of <=65535 B (BASELINE configs[%d])" % (
            n_files, args.file_bytes, "incompressible random" if name == "random" else "text-like", n, 1 if name == "random" else 2)
        cpu_files, cpu_file_bytes = min(args.cpu_sample_files, n_files), args.file_bytes
    elif name == "small_files":
        n_files = args.files or 370000
        d_in, d_off, d_len, n, raw_bytes, host_file = workloads.build_small_files(torch, dev, n_files, rank)
        desc = "%d image-like files per GPU, log-normal sizes, %.2f GB -> %d chunks (BASELINE configs[3], reference README.md:12-13)" % (n_files, raw_bytes / 1e9, n)
        cpu_files = 0
        k_cpu = min(n_files, args.cpu_sample_small_files)

        def populate(src):          # the first k_cpu files of the workload, 1 000 a directory (the reference's own data set is nested too)
            total = 0
            for i in range(k_cpu):
                if i % 1000 == 0:
                    os.makedirs(os.path.join(src, "d%04d" % (i // 1000)))
                b = host_file(i)
                total += len(b)
                with open(os.path.join(src, "d%04d" % (i // 1000), "img%06d.bin" % i), "wb") as f:
                    f.write(b)
            return total
        cpu_tree = (populate, "first %d files of the workload (same PRNG and seeds, ~%.0f MB), 1 000 a directory, warm page cache" % (k_cpu, k_cpu * 6.8e-3), [1, 8])
    else:   # one_file: ONE text-like file; its records are split over the ranks in contiguous ranges (strong scaling)
        total = args.one_file_bytes
        n_all = total // CHUNK + 1
        j0, j1 = n_all * rank // world, n_all * (rank + 1) // world
        n = j1 - j0
        # Every record its own text (Zipf words of one vocabulary, the sequence drawn from the record's seed: workloads.text_rows_device), generated
        # on the device 2 048 records at a time; the file's last record is ragged (64 GiB = 1 048 592 x 65 535 + 16 bytes).  Rounds 3-4 tiled
        # 1 024 distinct rows 1 024 times: equal lengths, perfect balance for a kernel that sorts its chunks by payload length.
        d_in = torch.zeros(n * STRIDE, dtype=torch.uint8, device=dev)
        view = d_in.view(n, STRIDE)
        seed0 = workloads.TEXT_SEED0 + 50_000_000
        for c0 in range(0, n, 2048):
            c1 = min(n, c0 + 2048)
            view[c0:c1, :CHUNK] = workloads.text_rows_device(torch, list(range(seed0 + j0 + c0, seed0 + j0 + c1)), CHUNK, dev)
        lens = torch.full((n,), CHUNK, dtype=torch.int32, device=dev)
        if j1 == n_all:
            lens[-1] = total % CHUNK
            view[-1, int(total % CHUNK):] = 0
        d_len, d_off = lens, torch.arange(n, dtype=torch.int64, device=dev) * STRIDE
        raw_bytes = int(d_len.sum().item())
        host_file, cpu_files = None, 0
        scaling = "strong"
        cpu_bytes = min(total, args.cpu_sample_one_file_bytes)

        cpu_rows = min(n, (cpu_bytes + CHUNK - 1) // CHUNK)
        head_host = view[:cpu_rows, :CHUNK].cpu().numpy()          # (rank 0's first records = the file's first bytes)

        def populate(src):          # the file's first cpu_bytes bytes: the same records, in the same order
            blob = head_host.tobytes()[:cpu_bytes]
            with open(os.path.join(src, "one.txt"), "wb") as f:
                f.write(blob)
            return len(blob)
        cpu_tree = (populate, "the file's first %d bytes (the same records) as ONE file -> one shard; the reference decodes a shard on one thread "
                              "(decompression.cpp:165-178), so decompress_GBps is its single-thread rate whatever the file's size" % cpu_bytes, [1])
        desc = "ONE %.1f GB text-like file = %d records in one shard, split into %d contiguous record ranges (BASELINE configs[4] scaled: 64 GiB there)" % (total / 1e9, n_all, world)
    d_out = torch.empty(n * STRIDE, dtype=torch.uint8, device=dev)
    d_olen = torch.zeros(n, dtype=torch.int32, device=dev)
    d_back = torch.empty(n * STRIDE, dtype=torch.uint8, device=dev)
    d_blen = torch.zeros(n, dtype=torch.int32, device=dev)
    d_stat = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def deflate():
        codec.deflate_dev(d_in, d_off, d_len, d_out, d_olen)

    def inflate():
        codec.inflate_dev(d_out, d_off, d_olen, d_back, d_blen, d_stat)

    def step():
        if not decompress_only:
            deflate()
        inflate()

    def fence():
        codec.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if decompress_only:
        deflate()                    # the shard is made once, outside every timed region
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    per_rank_ms = None
    if world > 1:
        # every rank's own time comes back to rank 0 beside the maximum: a SCALE record then shows which rank was slowest
        mine = torch.tensor([elapsed], dtype=torch.float64, device=cpu_dev if args.rehearse_on_one_gpu else dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [round(float(x.item()) / args.steps * 1e3, 3) for x in every]
        elapsed = max(float(x.item()) for x in every)

    # direction split + per-kernel times (untimed extra passes, HIP events on the codec's stream)
    fence()
    c0 = time.perf_counter(); deflate(); codec.sync(); c1 = time.perf_counter()
    inflate(); codec.sync(); c2 = time.perf_counter()
    codec.set_profiling(True)
    codec.stage_ms(reset=True)
    prof_passes = 2
    for _ in range(prof_passes):
        deflate()
        inflate()
    codec.sync()
    stage = codec.stage_ms(reset=True)
    codec.set_profiling(False)

    payload_bytes = int(d_olen.sum().item())
    back_bytes = int(d_blen.sum().item())
    verified = verify_bytes(torch, d_in, d_len, d_olen, d_back, d_blen, d_stat)
    verified["oracle"] = verify_oracle_sample(torch, args.oracle_sample, d_in, d_len, d_out, d_olen, d_back, d_blen)
    verified["ok"] = bool(verified["length_rule_ok"] and verified["chunks_with_wrong_bytes"] == 0 and
                          verified["oracle"]["payload_mismatches"] == 0 and verified["oracle"]["decode_mismatches"] == 0)
    raw_all = raw_bytes
    if world > 1:
        tot = torch.tensor([float(raw_bytes)], dtype=torch.float64, device=cpu_dev if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        raw_all = float(tot.item())
        flag = torch.tensor([0.0 if verified["ok"] else 1.0], dtype=torch.float64, device=cpu_dev if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM)
        verified["ranks_failed"] = int(flag.item())
        verified["ok"] = verified["ok"] and flag.item() == 0

    res = None
    if rank == 0:
        launches_per_pass = (n + args.max_batch - 1) // args.max_batch
        timed = {k: v for k, v in stage.items() if not (decompress_only and k != "inflate")}
        dom = max(timed, key=lambda k: timed[k])
        dom_launches = launches_per_pass * prof_passes if dom != "inflate" else prof_passes
        dom_ms = stage[dom] / dom_launches
        algo_bytes = (raw_bytes + payload_bytes) if dom != "inflate" else (payload_bytes + back_bytes)
        algo_per_launch = algo_bytes / (dom_launches / prof_passes)
        achieved = algo_per_launch / (dom_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % name)
        if os.path.exists(tpath) and name in ("random", "text") and (args.files or 10000) == 10000 and args.file_bytes == 262144:
            # PMC passes of this same workload, collected separately (tools/prof_gpu.sh); null if not measured
            # (a stage of the profile may be several kernels: stage lz_links = marks, links and the sort; stage lz_match = both searches)
            stage_kernels = {"lz_links": ["lz_dense_list", "lz_lists", "lz_links", "lz_sort", "lz_place"], "lz_match": ["lz_match", "lz_match_band"],
                             "plan": ["plan_probe", "plan_cost", "plan"], "encode": ["encode_stored", "encode"], "inflate": ["inflate_order", "inflate"]}
            per_kernel = json.load(open(tpath))
            got = [per_kernel[k]["hbm_bytes_per_launch"] for k in stage_kernels.get(dom, [dom]) if k in per_kernel]
            traffic = sum(got) if got else None
        comp_s, dec_s = c1 - c0, c2 - c1
        res = {
            "value": round(raw_all * args.steps / elapsed / 1e9, 3), "unit": "GB/s",
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "ms_per_step_by_rank": per_rank_ms, "scaling": scaling,
            "config": {"workload": desc, "chunk_bytes": CHUNK, "max_batch_chunks": args.max_batch,
                       "parallelism": ("record-range-per-gpu x%d" if name == "one_file" else "shard-per-gpu x%d") % world,
                       **({"scope_note": "device-resident inflate of the record ranges only.  Through the CLI (zwz_decompress_dir_ranked) ONE file is bounded by its "
                                         "single MD5 stream (~0.65 GB/s on one host core, the reference's verification.cpp:6-30 semantics): a 64 GiB file takes "
                                         "~105 s there whatever N is -- see DESIGN.md section 5"} if name == "one_file" else {}),
                       "timed": "inflate only" if decompress_only else "deflate + inflate"},
            "compress_GBps": round(raw_bytes / comp_s / 1e9, 3), "decompress_GBps": round(raw_bytes / dec_s / 1e9, 3),
            "payload_ratio": round(payload_bytes / raw_bytes, 4), "verified": verified,
            "stage_ms_per_pass": {k: round(v / prof_passes, 3) for k, v in stage.items()},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": "profiles/traffic_%s.json (a builder-run rocprofv3 PMC pass, 2*FETCH_SIZE + WRITE_SIZE per launch; not measured in this run)" % name if traffic is not None else None,
                         "algorithmic_bytes_per_launch": int(algo_per_launch), "launch_ms": round(dom_ms, 4),
                         # whole directions: algorithmic bytes (raw + payload) / wall time of the direction
                         "compress_frac_of_8TBps": round((raw_bytes + payload_bytes) / comp_s / 1e9 / HBM_PEAK_GBS, 5),
                         "compress_frac_of_6.29TBps": round((raw_bytes + payload_bytes) / comp_s / 1e9 / HBM_ACHIEVABLE_GBS, 5),
                         "decompress_frac_of_8TBps": round((payload_bytes + back_bytes) / dec_s / 1e9 / HBM_PEAK_GBS, 5),
                         "decompress_frac_of_6.29TBps": round((payload_bytes + back_bytes) / dec_s / 1e9 / HBM_ACHIEVABLE_GBS, 5)},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                if cpu_files:
                    res["cpu_baseline"] = cpu_baseline(name, host_file, cpu_files, cpu_file_bytes, args.cpu_baseline_port)
                elif cpu_tree is not None:
                    res["cpu_baseline"] = cpu_baseline_tree(*cpu_tree)
            except Exception as e:            # the GPU measurement above must not be lost to a failing CPU leg
                res["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "reference", "sample": None, "note": "CPU baseline failed: %r" % (e,)}
    del d_in, d_out, d_back
    torch.cuda.empty_cache()
    return res


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))          # children get WORLD_SIZE; this process never touches the GPU
    world = int(env_world or "1")
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s)\n" % (args.gpus, world))
        sys.exit(2)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the codec has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    zwz = importlib.import_module("parallel-data-compression-and-decompression_amd")
    if not os.path.exists(zwz.LIB_PATH) and rank == 0:      # a checkout that never ran build(): artefacts are git-ignored
        import __graft_entry__
        __graft_entry__.build()
    comm = None
    if world > 1:
        # First collective of the job, and a self-check of it: every rank contributes a one and its device index.  A SCALE
        # record then says how many ranks the backend (RCCL under "nccl") really joined and where each one ran.
        cdev = torch.device("cpu") if args.rehearse_on_one_gpu else dev
        ones = torch.ones(1, dtype=torch.int64, device=cdev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        mine = torch.tensor([local, torch.cuda.current_device()], dtype=torch.int64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        comm = {"backend": dist.get_backend(), "ranks_seen": int(ones.item()), "rccl_ranks_seen": int(ones.item()) if dist.get_backend() == "nccl" else None,
                "devices_by_rank": [int(x[1].item()) for x in every], "device_name": torch.cuda.get_device_name(local)}
        if comm["ranks_seen"] != world:
            raise SystemExit("bench.py: the all-reduce saw %d of %d ranks" % (comm["ranks_seen"], world))
        dist.barrier()
    codec = zwz.Codec(local, args.max_batch)

    # (both: the text-like configuration first -- it is the line's `value` -- then the incompressible one)
    names = ["text", "random"] if args.workload == "both" else [args.workload]
    results = [run_workload(args, torch, dist, codec, dev, world, rank, nm) for nm in names]
    if rank == 0:
        head = results[0]
        line = {"metric": "compress+decompress GB/s (raw bytes / (deflate + inflate time)), .zwz bit-exact",
                "value": head["value"], "unit": head["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": head["scaling"],
                "vs_baseline": None, "dtype": "u8", "data": "synthetic (own splitmix64 PRNG, seeds in tests/workloads.py)"}
        if comm is not None:
            line["comm"] = comm
            line["rccl_ranks_seen"] = comm["rccl_ranks_seen"]
        for k, v in head.items():
            line.setdefault(k, v)
        for nm, r in zip(names[1:], results[1:]):
            line[nm] = r
        if "random" in line and isinstance(line["random"], dict):
            # `value` is BASELINE configs[2] (text-like: the LZ77 + Huffman kernels the north star is about).  configs[1] (incompressible:
            # DEFLATE's stored-block path, a copy) is nested under "random", its headline figures lifted to the top level too.  (Rounds 1-4
            # had it the other way round: value = configs[1], value_text = configs[2].)
            t = line["random"]
            line["value_random"] = t["value"]
            line["ms_per_step_random"] = t["ms_per_step"]
            line["compress_GBps_random"], line["decompress_GBps_random"] = t["compress_GBps"], t["decompress_GBps"]
            line["roofline_random_frac"] = t["roofline"]["frac"]
            line["roofline_random_kernel"] = t["roofline"]["kernel"]
            # (the names rounds 1-4 used for the text figures, kept so that records compare across rounds)
            line["value_text"], line["compress_GBps_text"], line["decompress_GBps_text"] = line["value"], line["compress_GBps"], line["decompress_GBps"]
            line["roofline_text_frac"] = line["roofline"]["frac"]
            line["value_note"] = ("value = BASELINE configs[2] (10 000 x 256 KiB text-like files: dynamic-Huffman blocks, the LZ77 + Huffman kernels); "
                                  "value_random = configs[1] (incompressible files: stored blocks) -- same size, same step.  Until round 4 `value` was configs[1].")
    ok = all(r is None or r["verified"]["ok"] for r in results)
    codec.close()
    e2e = None
    if rank == 0 and world == 1 and args.workload == "both" and not args.no_e2e:
        # What a user of the reference's command line sees (main.cpp:78-159): the product's `main compress` + `main decompress` on the same
        # 10 000 x 256 KiB directories, on this box's file system, page cache warm.  NOT `value`: file I/O, PCIe and the process's fixed
        # costs are in it.  The GPU workspace of the runs above is released first (two processes share the card for a moment).
        del codec
        torch.cuda.empty_cache()
        e2e = e2e_legs(args)
    if rank == 0:
        if e2e is not None:
            line["e2e"] = e2e
            ok = ok and all(v.get("ok", True) for v in e2e.values() if isinstance(v, dict))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        sys.stderr.write("bench.py: verification FAILED (see \"verified\" in the JSON line)\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
