#!/usr/bin/env python3
"""Mint the golden fixtures in this directory.  Run ONLY in the build container, where
/root/reference exists and oracle/_ref/main has been built from it (make -C oracle ref).

Sources of truth:
  * chunk payloads: the system libz the reference binary links (must report 1.2.11), through
    Python's zlib module -- payload = zlib.compress(chunk, 6)[:65535]   (compression.cpp:119-132)
  * shards / decompressed trees / file ordering: the reference binary itself, run here as
    1, 2 and 3 MPI ranks (README.md:59) on the tree from tests/corpus.golden_tree().
Outputs: chunks.json, micro.json, tree.json, tree_N1/compressed_0.zwz, edges.json
"""
import hashlib, json, os, shutil, subprocess, sys, tempfile, zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import corpus  # noqa: E402

REF_MAIN = os.path.join(HERE, "..", "..", "oracle", "_ref", "main")
MPIEXEC = "/opt/conda/bin/mpiexec"
CHUNK = 65535


def sha(b):
    return hashlib.sha256(b).hexdigest()


def payload(chunk):
    return zlib.compress(chunk, 6)[:CHUNK]


def mint_micro():
    cases = [b"", b"a", b"ab", b"abc", b"abcd", b"a" * 10, b"hello world, hello world, hello world",
             bytes(1000), b"abcabcabcabcabcabcabcabc", bytes(range(256)), b"\xff" * 258, b"\x00\x01" * 700]
    return [{"in_hex": c.hex(), "payload_hex": payload(c).hex()} for c in cases]


def mint_chunks():
    rows = []
    sizes = [0, 1, 2, 3, 7, 100, 4096, 16383, 16384, 32768, 40000, 65509, 65510, 65534, 65535]
    seed = 1000
    for kind in ["random", "text", "lowent", "periodic", "skewed", "lz", "gradient", "zeros"]:
        for n in sizes:
            if kind in ("lz",) and n > 40000 and n != 65535:
                continue
            seed += 1
            data = corpus.make(kind, seed, n)
            p = payload(data)
            full = zlib.compress(data, 6)
            # what the reference's decompress_chunk() would write for this payload
            back = zlib.decompressobj().decompress(p)
            rows.append({"kind": kind, "seed": seed, "n": n, "in_sha256": sha(data),
                         "payload_len": len(p), "payload_sha256": sha(p), "stream_len": len(full),
                         "decoded_len": len(back), "decoded_sha256": sha(back)})
    return rows


def run_ref(args, nranks, cwd):
    cmd = [REF_MAIN] + args if nranks == 1 else [MPIEXEC, "-n", str(nranks), REF_MAIN] + args
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("reference failed: %s\n%s" % (cmd, r.stderr[-2000:]))
    return r.stdout


def mint_tree():
    files = corpus.golden_tree()
    out = {"zlib_version": zlib.ZLIB_RUNTIME_VERSION, "files": {k: {"size": len(v), "sha256": sha(v)} for k, v in files.items()},
           "runs": {}}
    work = tempfile.mkdtemp(prefix="zwz_golden_")
    try:
        src = os.path.join(work, "src")
        for rel, data in files.items():
            p = os.path.join(src, rel)
            os.makedirs(os.path.dirname(p), exist_ok=True)
            with open(p, "wb") as f:
                f.write(data)
        for n in (1, 2, 3):
            dst = os.path.join(work, "out%d" % n)
            run_ref(["compress", src, dst], n, work)
            listing = open(os.path.join(work, "sorted_files_by_size.txt")).read()
            shards = {}
            for name in sorted(os.listdir(dst)):
                b = open(os.path.join(dst, name), "rb").read()
                shards[name] = {"size": len(b), "sha256": sha(b)}
                if n == 1:
                    os.makedirs(os.path.join(HERE, "tree_N1"), exist_ok=True)
                    with open(os.path.join(HERE, "tree_N1", name), "wb") as f:
                        f.write(b)
            back = os.path.join(work, "back%d" % n)
            stdout = run_ref(["decompress", dst, back], 1, work)
            decoded = {}
            for rel in files:
                b = open(os.path.join(back, rel), "rb").read()
                decoded[rel] = {"size": len(b), "sha256": sha(b)}
            out["runs"][str(n)] = {"sorted_list": listing, "shards": shards, "decoded": decoded,
                                   "md5_mismatches": stdout.count("Expected MD5:")}
    finally:
        shutil.rmtree(work)
    return out


def mint_edges():
    """What the REFERENCE's decoder makes of damaged / reordered variants of the golden shard (tests/zwz_records.py derives
    them, byte for byte reproducibly): decoded files and the number of "MD5 mismatch" lines."""
    import zwz_records
    good = open(os.path.join(HERE, "tree_N1", "compressed_0.zwz"), "rb").read()
    out = {}
    for name, blob in zwz_records.edge_shards(good).items():
        work = tempfile.mkdtemp(prefix="zwz_edge_")
        try:
            os.makedirs(os.path.join(work, "in"))
            with open(os.path.join(work, "in", "compressed_0.zwz"), "wb") as f:
                f.write(blob)
            r = subprocess.run([REF_MAIN, "decompress", os.path.join(work, "in"), os.path.join(work, "out")],
                               capture_output=True, text=True, timeout=120)
            decoded = {}
            for root, _, names in os.walk(os.path.join(work, "out")):
                for n in names:
                    p = os.path.join(root, n)
                    rel = os.path.relpath(p, os.path.join(work, "out"))
                    if all(32 <= ord(ch) < 127 for ch in rel):     # (a cut header makes the reference try a garbage path)
                        b = open(p, "rb").read()
                        decoded[rel] = {"size": len(b), "sha256": sha(b)}
            out[name] = {"shard_size": len(blob), "shard_sha256": sha(blob), "exit": r.returncode, "decoded": decoded,
                         "md5_mismatches": r.stderr.count("MD5 mismatch for file:")}
        finally:
            shutil.rmtree(work)
    return out


if __name__ == "__main__":
    assert zlib.ZLIB_RUNTIME_VERSION == "1.2.11", zlib.ZLIB_RUNTIME_VERSION
    assert os.path.exists(REF_MAIN), "build oracle/_ref/main first: make -C oracle ref"
    json.dump(mint_micro(), open(os.path.join(HERE, "micro.json"), "w"), indent=0)
    json.dump(mint_chunks(), open(os.path.join(HERE, "chunks.json"), "w"), indent=0)
    json.dump(mint_tree(), open(os.path.join(HERE, "tree.json"), "w"), indent=1)
    json.dump(mint_edges(), open(os.path.join(HERE, "edges.json"), "w"), indent=1)
    print("golden fixtures written")
