import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, workloads, oracle_binding
zwz = importlib.import_module("parallel-data-compression-and-decompression_amd")
if os.environ.get("ZWZ_OLD_LIB"):
    zwz.LIB_PATH = os.path.join(ROOT, "tools", "dbg", "old_libzwz_hip.so")
    print("using", zwz.LIB_PATH)
o = oracle_binding.load()
dev = torch.device("cuda", 0)
STRIDE = 65536
for nfiles, mb in ((2000, 51200), (10000, 51200), (10000, 8192)):
    d_in, d_off, d_len, n, raw, host_file = workloads.build_equal_files(torch, dev, "random", nfiles, 262144)
    c = zwz.Codec(0, mb)
    d_out = torch.zeros(n * STRIDE, dtype=torch.uint8, device=dev); d_olen = torch.zeros(n, dtype=torch.int32, device=dev)
    d_back = torch.zeros(n * STRIDE, dtype=torch.uint8, device=dev); d_blen = torch.zeros(n, dtype=torch.int32, device=dev); d_stat = torch.zeros(n, dtype=torch.int32, device=dev)
    c.deflate_dev(d_in, d_off, d_len, d_out, d_olen); c.sync()
    c.inflate_dev(d_out, d_off, d_olen, d_back, d_blen, d_stat); c.sync()
    lens, olens, blens, stats = d_len.cpu().tolist(), d_olen.cpu().tolist(), d_blen.cpu().tolist(), d_stat.cpu().tolist()
    bad = [i for i in range(n) if lens[i] < 100 and blens[i] != lens[i]]
    print("files", nfiles, "max_batch", mb, "n", n, "bad tails", len(bad), "of", nfiles)
    if bad:
        import collections
        print("  blen histogram of bad tails:", collections.Counter(blens[i] for i in bad).most_common(6), "stat:", collections.Counter(stats[i] for i in bad).most_common(4),
              "first bad idx:", bad[:8], "last:", bad[-3:])
    for i in bad[:4]:
        chunk = d_in.view(n, STRIDE)[i, :lens[i]].cpu().numpy().tobytes()
        pay = d_out.view(n, STRIDE)[i, :olens[i]].cpu().numpy().tobytes()
        want = o.payload(chunk)
        alone = c.deflate_chunks([chunk])[0]
        back_alone, st_alone = c.inflate_chunks([want])
        print("  chunk", i, chunk.hex(), "payload", pay.hex(), "oracle", want.hex(), "same" if pay == want else "DIFF", "blen", blens[i], "stat", stats[i],
              "| alone deflate", "ok" if alone == want else alone.hex(), "alone inflate", len(back_alone[0]), st_alone)
    c.close()
    del d_in, d_out, d_back
