#!/usr/bin/env python3
"""BASELINE configs[4]'s shape through the command line: ONE big text-like file -> `main compress` (one shard) -> `main decompress`, with the
pipeline's timeline.  The decode of one file is bounded by its single MD5 stream (verification.cpp:6-30 semantics: one sequential hash per
file, ~0.65 GB/s on a host core) -- this run makes that bound a measurement.   usage: tools/one_file_e2e.py [GiB=8] [out.json]"""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus, workloads, e2e

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "one_file_e2e.json")
work = tempfile.mkdtemp(prefix="zwz_one_", dir=os.environ.get("ZWZ_E2E_TMP", "/tmp"))
res = {"file_GiB": gib, "filesystem": e2e.filesystem_of(work), "page_cache": "warm", "host_cpus": os.cpu_count()}
try:
    src = os.path.join(work, "data", "src"); os.makedirs(src)
    total = int(gib * (1 << 30))
    blocks = [corpus.text_like(workloads.TEXT_SEED0 + 7000 + i, 1 << 20) for i in range(64)]      # 64 distinct MiB of text, repeated
    t0 = time.perf_counter()
    with open(os.path.join(src, "one.txt"), "wb") as f:
        left, i = total, 0
        while left > 0:
            b = blocks[i % 64][:left]; f.write(b); left -= len(b); i += 1
    res["bytes"] = total; res["generate_s"] = round(time.perf_counter() - t0, 1)
    zwz, back = os.path.join(work, "zwz"), os.path.join(work, "back")
    res["compress"] = e2e.run([e2e.MAIN, "compress", src, zwz], True)
    res["decompress"] = e2e.run([e2e.MAIN, "decompress", zwz, back], True)
    for d in ("compress", "decompress"):
        res[d]["GBps_banner"] = round(total / (res[d]["banner_s"] or res[d]["wall_s"]) / 1e9, 3)
    res["shard_bytes"] = os.path.getsize(os.path.join(zwz, "compressed_0.zwz"))
    res["decoded_equals_source"] = os.path.getsize(os.path.join(back, "one.txt")) == total and res["decompress"]["md5_mismatch_lines"] == 0
finally:
    shutil.rmtree(work, ignore_errors=True)
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("file_GiB", "bytes", "generate_s", "shard_bytes", "decoded_equals_source")}))
for d in ("compress", "decompress"):
    print(d, res[d]["banner_s"], "s", res[d]["GBps_banner"], "GB/s"); print("  " + "\n  ".join(res[d].get("timeline", [])))
