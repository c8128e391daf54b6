#!/usr/bin/env python3
"""End-to-end figures of the drop-in CLI (`main compress|decompress <src> <dst>`, reference main.cpp:78-159) on the headline workloads:
what a user of the reference's command line sees, beside bench.py's kernel-scope figures.

  tools/e2e.py <random|text|small_files> [n_files] [out.json] [--scale K] [--ref]

Builds the directory on this box's file system (same PRNG and seeds as bench.py's device batches; text: the 256 distinct files bench.py
tiles), then runs the product's `main compress` and `main decompress` twice each: once plain (the reference's own "Time Taken" banner
and the wall clock around the process), once with ZWZ_TIMELINE=1 for the pipeline's timeline, from which the fixed cost (context +
self-tests, pinned staging, enumeration + sort) and the steady state (slices) are split.  A sample of the shard's records is compared with
the CPU oracle's payloads, and the decoded tree with the source (text, small_files: byte for byte; random: the reference's lossy chunks
come back short, so sizes only).  --scale K repeats it on a K times larger directory (steady state visible).  --ref also times the
reference binary (oracle/_ref/main) at one rank.  Page cache warm (the files were just written); file system stated."""
import json, os, re, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import corpus, workloads

MAIN = os.path.join(ROOT, "parallel-data-compression-and-decompression_amd", "main")
REF = os.path.join(ROOT, "oracle", "_ref", "main")
CHUNK = 65535


def filesystem_of(path):
    best, fs = "", None
    try:
        for line in open("/proc/mounts"):
            parts = line.split()
            if len(parts) >= 3 and (path == parts[1] or path.startswith(parts[1].rstrip("/") + "/")) and len(parts[1]) >= len(best):
                best, fs = parts[1], parts[2]
    except OSError:
        pass
    return fs


def populate(workload, src, n_files, file_bytes=262144):
    """-> (total bytes, file_of(i) -> (relative path, bytes))"""
    os.makedirs(src)
    if workload == "small_files":
        sizes = workloads.small_file_sizes(n_files)

        def file_of(i):
            return os.path.join("d%03d" % (i % 997), "s%02d" % (i % 13), "img_%06d.raw" % i), workloads.small_file_bytes(i, sizes[i])
    elif workload == "random":
        def file_of(i):
            return "f%05d.bin" % i, corpus.random_bytes(workloads.RANDOM_SEED0 + i, file_bytes)
    else:
        distinct = [corpus.text_like(workloads.TEXT_SEED0 + i, file_bytes) for i in range(min(n_files, 256))]

        def file_of(i):
            return "f%05d.txt" % i, distinct[i % len(distinct)]
    total, dirs = 0, set()
    for i in range(n_files):
        rel, data = file_of(i)
        d = os.path.dirname(rel)
        if d and d not in dirs:
            os.makedirs(os.path.join(src, d), exist_ok=True)
            dirs.add(d)
        with open(os.path.join(src, rel), "wb") as f:
            f.write(data)
        total += len(data)
    return total, file_of


def run(cmd, verbose=False):
    env = dict(os.environ)
    if verbose:
        env["ZWZ_TIMELINE"] = "1"        # the phase timelines without ZWZ_VERBOSE's per-file messages
    t = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    wall = time.perf_counter() - t
    m = re.findall(r"Time Taken: ([0-9.eE+-]+) seconds", r.stdout)
    out = {"rc": r.returncode, "banner_s": float(m[-1]) if m else None, "wall_s": round(wall, 3),
           "md5_mismatch_lines": r.stderr.count("MD5 mismatch for file:") + r.stdout.count("MD5 mismatch for file:")}
    if r.returncode:
        out["stderr_tail"] = r.stderr[-400:]
    if verbose:
        out["timeline"] = [l for l in r.stderr.splitlines() if l.startswith("zwz:")][:60]
    return out


def split_timeline(lines):
    """fixed cost / steady state out of the pipeline's own timeline (csrc/zwz_pipeline.cpp, ZWZ_VERBOSE)."""
    marks = {}
    for l in lines:
        m = re.match(r"zwz: \[([0-9.]+) s\] (.*)", l)
        if m:
            marks[m.group(2)] = float(m.group(1))
    return marks


def summarize(res):
    """The figures bench.py puts on its JSON line: per direction the reference's own banner (main.cpp:148-155), the wall clock, GB/s of raw bytes,
    and -- from the ZWZ_TIMELINE run's timeline -- the split into the steady state (the slices: read / copy in / kernels / copy out / write,
    pipelined) and everything around it (process and HIP start-up, the context's self-tests, enumeration + sort, pinning, teardown)."""
    out = {"workload": res["workload"], "n_files": res["n_files"], "bytes": res["bytes"], "filesystem": res["filesystem"], "page_cache": res["page_cache"], "ok": res.get("ok")}
    for d in ("compress", "decompress"):
        r, v = res[d], res.get(d + "_verbose", {})
        o = {"banner_s": r["banner_s"], "wall_s": r["wall_s"], "GBps": r["GBps_banner"]}
        steady = None
        for l in v.get("timeline", []):
            m = re.search(r"slices of <= \d+ chunks in ([0-9.]+) s: waiting for reads ([0-9.]+) s, for the GPU ([0-9.]+) s, writing ([0-9.]+) s", l)
            if m:
                steady = float(m.group(1)); o["waits_s"] = {"reads": float(m.group(2)), "gpu": float(m.group(3)), "shard_writer": float(m.group(4))}
            m = re.search(r"decode: .* slices done at ([0-9.]+) s \(waited ([0-9.]+) s for the GPU, ([0-9.]+) s for the writers\)", l)
            if m:
                steady = float(m.group(1)); o["waits_s"] = {"gpu": float(m.group(2)), "file_writers": float(m.group(3))}
        if steady:
            o["steady_s"] = steady
            o["steady_GBps"] = round(res["bytes"] / steady / 1e9, 3)
            if v.get("banner_s"):
                o["fixed_s"] = round(max(0.0, v["banner_s"] - steady), 3)       # (of the verbose run: its slices and its banner)
        out[d] = o
    out["oracle_sample"] = res.get("oracle_sample")
    return out


def check_shard_sample(shard, src, k_files):
    """The shard's first records against the CPU oracle: payload == oracle.payload(chunk) for the chunks of the first k_files files."""
    import oracle_binding
    o = oracle_binding.load()
    bad = checked = 0
    with open(shard, "rb") as f:
        files_done = 0
        while files_done < k_files:
            head = f.read(8)
            if len(head) < 8:
                break
            total, plen = int.from_bytes(head[:4], "little", signed=True), int.from_bytes(head[4:], "little", signed=True)
            path = f.read(plen).decode()
            seq = int.from_bytes(f.read(4), "little", signed=True)
            last = f.read(1)[0]
            payload = f.read(total - (4 + plen + 4 + 1))
            if last:
                f.read(32)
                files_done += 1
            with open(os.path.join(src, path), "rb") as g:
                g.seek(seq * CHUNK)
                chunk = g.read(CHUNK)
            checked += 1
            bad += payload != o.payload(chunk)
    return {"records_checked": checked, "payload_mismatches": bad}


def measure(workload, n_files, with_ref):
    base = os.environ.get("ZWZ_E2E_TMP", "/tmp")
    work = tempfile.mkdtemp(prefix="zwz_e2e_", dir=base)
    res = {"workload": workload, "n_files": n_files, "filesystem": filesystem_of(work), "page_cache": "warm (files written just before)", "host_cpus": os.cpu_count()}
    try:
        t0 = time.perf_counter()
        src = os.path.join(work, "data", "src")
        os.makedirs(os.path.dirname(src))
        total, _ = populate(workload, src, n_files)
        res["bytes"] = total
        res["generate_s"] = round(time.perf_counter() - t0, 1)
        zwz, back = os.path.join(work, "zwz"), os.path.join(work, "back")
        for tag, verbose in (("", False), ("_verbose", True)):
            shutil.rmtree(zwz, ignore_errors=True)
            shutil.rmtree(back, ignore_errors=True)
            res["compress" + tag] = run([MAIN, "compress", src, zwz], verbose)
            res["decompress" + tag] = run([MAIN, "decompress", zwz, back], verbose)
        for d in ("compress", "decompress"):
            r = res[d]
            secs = r["banner_s"] or r["wall_s"]
            r["GBps_banner"] = round(total / secs / 1e9, 3)
            r["GBps_wall"] = round(total / r["wall_s"] / 1e9, 3)
        shard = os.path.join(zwz, "compressed_0.zwz")
        res["shard_bytes"] = os.path.getsize(shard)
        res["oracle_sample"] = check_shard_sample(shard, src, 24 if workload != "small_files" else 400)
        if workload == "random":        # full incompressible chunks come back 22 bytes short, as from the reference (compression.cpp:127-132)
            n_back = sum(len(fs) for _, _, fs in os.walk(back))
            res["decoded_files"] = n_back
            res["decoded_ok"] = n_back == n_files
        else:
            res["decoded_equals_source"] = subprocess.call(["diff", "-rq", back, src], stdout=subprocess.DEVNULL) == 0
            res["decoded_ok"] = res["decoded_equals_source"]
        res["ok"] = bool(res["compress"]["rc"] == 0 and res["decompress"]["rc"] == 0 and res["oracle_sample"]["payload_mismatches"] == 0 and res["decoded_ok"])
        if with_ref and os.path.exists(REF):
            rz, rb = os.path.join(work, "rzwz"), os.path.join(work, "rback")
            res["reference_compress_1rank"] = run([REF, "compress", src, rz])
            res["reference_decompress"] = run([REF, "decompress", rz, rb])
            res["shards_identical"] = subprocess.call(["cmp", "-s", shard, os.path.join(rz, "compressed_0.zwz")]) == 0
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return res


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    workload = args[0] if args else "text"
    n = int(args[1]) if len(args) > 1 else (370000 if workload == "small_files" else 10000)
    out = args[2] if len(args) > 2 else os.path.join(ROOT, "gpurun_out", "e2e_%s.json" % workload)
    scale = int(sys.argv[sys.argv.index("--scale") + 1]) if "--scale" in sys.argv else 0
    res = measure(workload, n, "--ref" in sys.argv)
    if scale > 1:
        res["scaled_x%d" % scale] = measure(workload, n * scale, False)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    brief = {k: res[k] for k in ("workload", "n_files", "bytes", "generate_s", "ok")}
    for d in ("compress", "decompress"):
        brief[d] = {k: res[d][k] for k in ("banner_s", "wall_s", "GBps_banner")}
    print(json.dumps(brief))
    for d in ("compress_verbose", "decompress_verbose"):
        print(d, *res[d].get("timeline", []), sep="\n  ")
    if scale > 1:
        s = res["scaled_x%d" % scale]
        print("x%d:" % scale, {d: {k: s[d][k] for k in ("banner_s", "wall_s", "GBps_banner")} for d in ("compress", "decompress")}, s["ok"])
        for d in ("compress_verbose", "decompress_verbose"):
            print(d, *s[d].get("timeline", []), sep="\n  ")
