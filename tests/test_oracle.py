"""The CPU oracle against the golden fixtures minted from the reference binary + libz 1.2.11
(tests/golden/make_golden.py), and live against the box's libz when it is 1.2.11."""
import hashlib
import json
import os
import zlib

import pytest

import corpus


def sha(b):
    return hashlib.sha256(b).hexdigest()


def test_micro_vectors(oracle, golden_dir):
    for row in json.load(open(os.path.join(golden_dir, "micro.json"))):
        data = bytes.fromhex(row["in_hex"])
        assert oracle.payload(data).hex() == row["payload_hex"]
        back, n, st = oracle.inflate(bytes.fromhex(row["payload_hex"]))
        assert back == data and st == 0


def test_empty_chunk_stream(oracle):
    # empty input -> one static block holding only EOB
    assert oracle.payload(b"") == bytes.fromhex("789c030000000001")


def test_chunk_vectors(oracle, golden_dir):
    rows = json.load(open(os.path.join(golden_dir, "chunks.json")))
    assert len(rows) > 100
    for row in rows:
        data = corpus.make(row["kind"], row["seed"], row["n"])
        assert sha(data) == row["in_sha256"], "corpus generator drifted"
        p = oracle.payload(data)
        assert (len(p), sha(p)) == (row["payload_len"], row["payload_sha256"]), row
        assert len(oracle.deflate6(data)) == row["stream_len"]
        back, n, st = oracle.inflate(p)
        assert (n, sha(back)) == (row["decoded_len"], row["decoded_sha256"]), row


def test_truncation_is_lossy_like_reference(oracle):
    # a full random chunk deflates to 65561 B, stored as 65535, decodes to 65513
    data = corpus.random_bytes(5, 65535)
    assert len(oracle.deflate6(data)) == 65561
    p = oracle.payload(data)
    assert len(p) == 65535
    back, n, st = oracle.inflate(p)
    assert n == 65513 and back == data[:65513] and st == 1


def test_md5_and_adler(oracle):
    for n in [0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 1000, 70000]:
        d = corpus.random_bytes(n + 1, n)
        assert oracle.md5_hex(d) == hashlib.md5(d).hexdigest()
        assert oracle.adler32(d) == zlib.adler32(d)


def _write_tree(root):
    files = corpus.golden_tree()
    for rel, data in files.items():
        p = os.path.join(root, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "wb") as f:
            f.write(data)
    return files


@pytest.mark.parametrize("nranks", [1, 2, 3])
def test_shards_match_reference(oracle, golden_dir, tmp_path, nranks):
    tree = json.load(open(os.path.join(golden_dir, "tree.json")))
    run = tree["runs"][str(nranks)]
    src = tmp_path / "src"
    _write_tree(str(src))
    rec = tmp_path / "sorted_files_by_size.txt"
    rec.write_text(run["sorted_list"])          # the reference's own ordering (readdir + introsort ties)
    dst = tmp_path / "dst"
    dst.mkdir()
    for r in range(nranks):
        assert oracle.compress_shard(str(src), str(dst), str(rec), r, nranks) == 0
    got = {n: open(dst / n, "rb").read() for n in sorted(os.listdir(dst))}
    assert {n: {"size": len(b), "sha256": sha(b)} for n, b in got.items()} == run["shards"]
    if nranks == 1:
        assert got["compressed_0.zwz"] == open(os.path.join(golden_dir, "tree_N1", "compressed_0.zwz"), "rb").read()


def test_idle_rank_writes_no_shard(oracle, tmp_path):
    # main.cpp:47-51: ranks >= number of listed files do not call do_compression
    (tmp_path / "src").mkdir()
    (tmp_path / "src" / "a.txt").write_bytes(b"aaa")
    (tmp_path / "src" / "b.txt").write_bytes(b"b")
    rec = tmp_path / "rec.txt"
    rec.write_text("a.txt\nb.txt\n")
    (tmp_path / "dst").mkdir()
    for r in range(4):
        assert oracle.compress_shard(str(tmp_path / "src"), str(tmp_path / "dst"), str(rec), r, 4) == 0
    assert sorted(os.listdir(tmp_path / "dst")) == ["compressed_0.zwz", "compressed_1.zwz"]


def test_decompress_golden_shard(oracle, golden_dir, tmp_path):
    tree = json.load(open(os.path.join(golden_dir, "tree.json")))
    run = tree["runs"]["1"]
    out = tmp_path / "back"
    out.mkdir()
    bad = oracle.decompress_shard(os.path.join(golden_dir, "tree_N1", "compressed_0.zwz"), str(out))
    assert bad == run["md5_mismatches"] == 2    # exact.bin and rand140k.bin come back short
    for rel, want in run["decoded"].items():
        b = open(out / rel, "rb").read()
        assert {"size": len(b), "sha256": sha(b)} == want, rel


@pytest.mark.skipif(zlib.ZLIB_RUNTIME_VERSION != "1.2.11", reason="live libz is not the pinned 1.2.11")
def test_live_differential_vs_libz(oracle):
    seed = 9000
    for kind in corpus.KINDS:
        for n in [0, 5, 300, 5000, 33000, 65535, 70000, 140000]:
            seed += 1
            data = corpus.make(kind, seed, n)
            ref = zlib.compress(data, 6)
            assert oracle.deflate6(data) == ref, (kind, n)
            cut = (seed * 7919) % (len(ref) + 1)
            want = zlib.decompressobj().decompress(ref[:cut])
            back, total, st = oracle.inflate(ref[:cut])
            assert back == want, (kind, n, cut)


def test_window_slide_corner(oracle):
    # position 65274 may not use a candidate at 32768: zlib's window slides there and the rebased
    # position becomes NIL
    if zlib.ZLIB_RUNTIME_VERSION != "1.2.11":
        pytest.skip("needs libz 1.2.11")
    for t in range(20):
        n = 65300 + t
        a = bytearray(corpus.random_bytes(777 + t, n))
        a[65274:65274 + 6] = a[32768:32768 + 6]
        assert oracle.deflate6(bytes(a)) == zlib.compress(bytes(a), 6)


def _tree_of(root):
    out = {}
    for d, _, names in os.walk(root):
        for n in names:
            b = open(os.path.join(d, n), "rb").read()
            out[os.path.relpath(os.path.join(d, n), root)] = {"size": len(b), "sha256": sha(b)}
    return out


def test_edge_shards_match_reference(oracle, golden_dir, tmp_path):
    """Reordered and damaged shards (tests/zwz_records.py) against what the reference binary itself made of them
    (edges.json, minted by make_golden.py): records out of order incl. a last chunk that arrives early, a shard cut at a
    record boundary, inside an MD5, inside a payload, inside a header, and an empty shard (decompression.cpp:65-162)."""
    import zwz_records
    edges = json.load(open(os.path.join(golden_dir, "edges.json")))
    good = open(os.path.join(golden_dir, "tree_N1", "compressed_0.zwz"), "rb").read()
    shards = zwz_records.edge_shards(good)
    assert set(shards) == set(edges)
    for name, blob in shards.items():
        want = edges[name]
        assert (len(blob), sha(blob)) == (want["shard_size"], want["shard_sha256"]), "edge derivation drifted: " + name
        out = tmp_path / name
        out.mkdir()
        shard = tmp_path / (name + ".zwz")
        shard.write_bytes(blob)
        bad = oracle.decompress_shard(str(shard), str(out))
        assert bad == want["md5_mismatches"], name
        assert _tree_of(str(out)) == want["decoded"], name
