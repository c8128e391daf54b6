#!/usr/bin/env python3
"""BASELINE configs[3] at full size, end to end through the command lines (the reference's one published workload:
README.md:12-13, ~370 000 images / ~2.5 GB; charts README.md:201-204): `main compress` + `main decompress` of the product on
one GPU beside the reference binary (oracle/_ref/main, 1 MPI rank and, if mpiexec is there, 8) on the same box, same
files (tests/workloads.py: log-normal sizes, image-like content, own PRNG).  Shards and decoded trees are compared byte
for byte.  Writes a JSON summary (copy it to profiles/).   usage: tools/config4_e2e.py [n_files] [out.json]"""
import json, os, re, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import workloads

N = int(sys.argv[1]) if len(sys.argv) > 1 else 370000
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "config4_e2e.json")
MAIN = os.path.join(ROOT, "parallel-data-compression-and-decompression_amd", "main")
REF = os.path.join(ROOT, "oracle", "_ref", "main")
MPIEXEC = "/opt/conda/bin/mpiexec"
work = tempfile.mkdtemp(prefix="zwz_c4_", dir=os.environ.get("ZWZ_E2E_TMP", "/tmp"))
res = {"n_files": N}
try:
    t0 = time.time()
    src = os.path.join(work, "data", "src")
    sizes = workloads.small_file_sizes(N)
    dirs = set()
    for i in range(N):
        d = os.path.join(src, "d%03d" % (i % 997), "s%02d" % (i % 13))
        if d not in dirs:
            os.makedirs(d, exist_ok=True); dirs.add(d)
        with open(os.path.join(d, "img_%06d.raw" % i), "wb") as f:
            f.write(workloads.small_file_bytes(i, sizes[i]))
        if i % 50000 == 0:
            print("generated", i, "files", flush=True)
    res["bytes"] = int(sizes.sum())
    res["generate_s"] = round(time.time() - t0, 1)
    print("tree: %d files, %.3f GB, %.0f s" % (N, res["bytes"] / 1e9, res["generate_s"]), flush=True)

    def run(cmd, env=None):
        t = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        m = re.search(r"Time Taken: ([0-9.e+-]+) seconds", r.stdout)
        return {"rc": r.returncode, "time_taken_s": float(m.group(1)) if m else None, "wall_s": round(time.time() - t, 2),
                "md5_mismatch_lines": r.stderr.count("MD5 mismatch for file:"), "stderr_tail": r.stderr[-300:] if r.returncode else ""}

    listing = os.path.join(work, "data", "sorted_files_by_size.txt")
    res["product_compress"] = run([MAIN, "compress", src, os.path.join(work, "zwz")])
    if os.path.exists(listing):
        shutil.copy(listing, os.path.join(work, "list_product.txt"))
    res["product_decompress"] = run([MAIN, "decompress", os.path.join(work, "zwz"), os.path.join(work, "back")])
    print("product:", res["product_compress"], res["product_decompress"], flush=True)
    if os.environ.get("ZWZ_E2E_VERBOSE"):      # the pipeline's own timeline (where the host side spends its time)
        shutil.rmtree(os.path.join(work, "zwz")); shutil.rmtree(os.path.join(work, "back"))
        r = subprocess.run([MAIN, "compress", src, os.path.join(work, "zwz")], capture_output=True, text=True, env=dict(os.environ, ZWZ_VERBOSE="1"))
        res["product_compress_timeline"] = [l for l in r.stderr.splitlines() if l.startswith("zwz:")][:40]
        r = subprocess.run([MAIN, "decompress", os.path.join(work, "zwz"), os.path.join(work, "back")], capture_output=True, text=True, env=dict(os.environ, ZWZ_VERBOSE="1"))
        res["product_decompress_timeline"] = [l for l in r.stderr.splitlines() if l.startswith("zwz:")][:40]
    if os.path.exists(REF) and not os.environ.get("ZWZ_E2E_SKIP_REF"):
        res["reference_compress_1rank"] = run([REF, "compress", src, os.path.join(work, "rzwz")])
        res["lists_identical"] = os.path.exists(os.path.join(work, "list_product.txt")) and open(listing, "rb").read() == open(os.path.join(work, "list_product.txt"), "rb").read()
        res["reference_decompress"] = run([REF, "decompress", os.path.join(work, "rzwz"), os.path.join(work, "rback")])
        print("reference:", res["reference_compress_1rank"], res["reference_decompress"], flush=True)
        res["shards_identical"] = subprocess.call(["cmp", "-s", os.path.join(work, "zwz", "compressed_0.zwz"), os.path.join(work, "rzwz", "compressed_0.zwz")]) == 0
        res["decoded_trees_identical"] = subprocess.call(["diff", "-rq", os.path.join(work, "back"), os.path.join(work, "rback")], stdout=subprocess.DEVNULL) == 0
        res["decoded_equals_source"] = subprocess.call(["diff", "-rq", os.path.join(work, "back"), src], stdout=subprocess.DEVNULL) == 0
        res["shard_bytes"] = os.path.getsize(os.path.join(work, "zwz", "compressed_0.zwz"))
        if os.path.exists(MPIEXEC):
            shutil.rmtree(os.path.join(work, "rzwz")); shutil.rmtree(os.path.join(work, "rback"))
            res["reference_compress_8ranks"] = run([MPIEXEC, "-n", "8", REF, "compress", src, os.path.join(work, "rzwz8")])
    else:
        res["reference"] = "oracle/_ref/main absent"
    res["host_cpus"] = os.cpu_count()
finally:
    shutil.rmtree(work, ignore_errors=True)
os.makedirs(os.path.dirname(OUT), exist_ok=True)
json.dump(res, open(OUT, "w"), indent=1)
print(json.dumps(res, indent=1))
