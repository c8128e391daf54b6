#!/usr/bin/env python3
"""bench.py -- compress + decompress throughput of the chunk codec on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: deflate every 65535-byte chunk of a
synthetic directory image resident in HBM, then inflate the resulting payloads (also resident).
Default workload = BASELINE.json configs[1]: 10 000 x 256 KiB incompressible random files
(50 000 chunks: 4 full + one 4-byte chunk per file).  `value` = raw bytes / (deflate + inflate
time), whole job over all ranks (max over ranks of the timed region).  Ranks own disjoint files
(shards partition one-per-rank): weak scaling, no data-path collective.

Extra objects on the JSON line:
  roofline      dominant kernel's (raw + payload) bytes per launch / its mean launch duration,
                measured with HIP events on the codec's own stream, against the 8 TB/s HBM peak
  cpu_baseline  the reference binary (oracle/_ref/main, kind "reference") or, if it cannot run on
                this box, the oracle restatement (kind "port"), timed on a bounded sample
"""
import argparse
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHUNK, STRIDE = 65535, 65536
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="random", choices=["random", "text"])
    ap.add_argument("--files", type=int, default=10000)
    ap.add_argument("--file-bytes", type=int, default=262144)
    ap.add_argument("--max-batch", type=int, default=51200)   # whole config-2 batch in one launch per kernel: ~48 GB of workspace, sized for 288 GB of HBM
    ap.add_argument("--cpu-sample-files", type=int, default=2400)   # ~12-17 s of reference CPU work
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 ranks share cuda:0 over gloo: exercises the multi-rank code path on a 1-GPU box (not a measurement)")
    return ap.parse_args()


def build_batch(torch, dev, workload, n_files, file_bytes, seed):
    """Device image of n_files files cut the reference's way (compression.cpp:52-64)."""
    import numpy as np
    full, tail = divmod(file_bytes, CHUNK)
    per_file = full + 1                       # a short (possibly empty) read ends the file
    n = n_files * per_file
    lens = np.full((n_files, per_file), CHUNK, dtype=np.uint32)
    lens[:, -1] = tail
    d_in = torch.zeros(n * STRIDE, dtype=torch.uint8, device=dev)
    view = d_in.view(n_files, per_file, STRIDE)
    if workload == "random":
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        step = 500
        for f0 in range(0, n_files, step):
            f1 = min(n_files, f0 + step)
            r = torch.randint(0, 256, (f1 - f0, file_bytes), dtype=torch.uint8, device=dev, generator=g)
            for c in range(full):
                view[f0:f1, c, :CHUNK] = r[:, c * CHUNK:(c + 1) * CHUNK]
            if tail:
                view[f0:f1, full, :tail] = r[:, full * CHUNK:]
    else:
        import corpus
        distinct = min(n_files, 256)
        host = np.zeros((distinct, file_bytes), dtype=np.uint8)
        for i in range(distinct):
            host[i] = np.frombuffer(corpus.text_like(seed * 1000 + i, file_bytes), dtype=np.uint8)
        t = torch.from_numpy(host).to(dev)
        for f0 in range(0, n_files, distinct):
            f1 = min(n_files, f0 + distinct)
            r = t[:f1 - f0]
            for c in range(full):
                view[f0:f1, c, :CHUNK] = r[:, c * CHUNK:(c + 1) * CHUNK]
            if tail:
                view[f0:f1, full, :tail] = r[:, full * CHUNK:]
    d_len = torch.from_numpy(lens.reshape(-1).astype(np.int32)).to(dev)
    d_off = (torch.arange(n, dtype=torch.int64, device=dev) * STRIDE)
    return d_in, d_off, d_len, n, int(n_files) * int(file_bytes)


def cpu_baseline(workload, file_bytes, n_files):
    """Reference CPU path on a bounded sample of the same workload."""
    import corpus
    work = tempfile.mkdtemp(prefix="zwz_cpu_")
    try:
        src = os.path.join(work, "data", "src")
        os.makedirs(src)
        for i in range(n_files):
            data = corpus.random_bytes(90000 + i, file_bytes) if workload == "random" else corpus.text_like(90000 + i, file_bytes)
            with open(os.path.join(src, "f%05d.bin" % i), "wb") as f:
                f.write(data)
        total = n_files * file_bytes
        dst, back = os.path.join(work, "zwz"), os.path.join(work, "back")
        ref = os.path.join(ROOT, "oracle", "_ref", "main")
        kind, cores = None, 1
        if os.path.exists(ref):
            try:
                t0 = time.perf_counter()
                subprocess.run([ref, "compress", src, dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
                t1 = time.perf_counter()
                subprocess.run([ref, "decompress", dst, back], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
                t2 = time.perf_counter()
                kind, cores = "reference", 2     # producer + consumer threads (compression.cpp:162)
            except Exception:
                kind = None
        if kind is None:
            import oracle_binding
            o = oracle_binding.load()
            rec = os.path.join(work, "rec.txt")
            with open(rec, "w") as f:
                f.write("".join("f%05d.bin\n" % i for i in range(n_files)))
            os.makedirs(dst, exist_ok=True)
            os.makedirs(back, exist_ok=True)
            t0 = time.perf_counter()
            o.compress_shard(src, dst, rec, 0, 1)
            t1 = time.perf_counter()
            o.decompress_shard(os.path.join(dst, "compressed_0.zwz"), back)
            t2 = time.perf_counter()
            kind, cores = "port", 1
        return {"value": round(total / (t2 - t0) / 1e9, 5), "unit": "GB/s", "cores": cores, "kind": kind,
                "sample": "%d x %d B %s files, 1 rank: compress %.2f s + decompress %.2f s (warm page cache)"
                          % (n_files, file_bytes, workload, t1 - t0, t2 - t1),
                "compress_GBps": round(total / (t1 - t0) / 1e9, 5), "decompress_GBps": round(total / (t2 - t1) / 1e9, 5)}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    if world > 1:
        dist.barrier()
    codec = zwz.Codec(local, args.max_batch)

    d_in, d_off, d_len, n, raw_bytes = build_batch(torch, dev, args.workload, args.files, args.file_bytes, 1234 + rank)
    d_out = torch.empty(n * STRIDE, dtype=torch.uint8, device=dev)
    d_olen = torch.zeros(n, dtype=torch.int32, device=dev)
    d_back = torch.empty(n * STRIDE, dtype=torch.uint8, device=dev)
    d_blen = torch.zeros(n, dtype=torch.int32, device=dev)
    d_stat = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step():
        codec.deflate_dev(d_in, d_off, d_len, d_out, d_olen)
        codec.inflate_dev(d_out, d_off, d_olen, d_back, d_blen, d_stat)

    def fence():
        codec.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cpu") if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # direction split + per-kernel times (untimed extra passes, HIP events on the codec's stream)
    fence()
    c0 = time.perf_counter(); codec.deflate_dev(d_in, d_off, d_len, d_out, d_olen); codec.sync(); c1 = time.perf_counter()
    codec.inflate_dev(d_out, d_off, d_olen, d_back, d_blen, d_stat); codec.sync(); c2 = time.perf_counter()
    codec.set_profiling(True)
    codec.stage_ms(reset=True)
    prof_passes = 2
    for _ in range(prof_passes):
        step()
    codec.sync()
    stage = codec.stage_ms(reset=True)
    codec.set_profiling(False)

    payload_bytes = int(d_olen.sum().item())
    back_bytes = int(d_blen.sum().item())
    # parity property on the full batch: every chunk either round-trips or is a reference-truncated one
    ok = bool(((d_blen == d_len) | ((d_olen == CHUNK) & (d_stat == 1))).all().item())

    if rank == 0:
        launches_per_pass = (n + args.max_batch - 1) // args.max_batch
        dom = max(stage, key=lambda k: stage[k])
        dom_launches = launches_per_pass * prof_passes if dom != "inflate" else prof_passes
        dom_ms = stage[dom] / dom_launches
        algo_bytes = (raw_bytes + payload_bytes) if dom != "inflate" else (payload_bytes + back_bytes)
        algo_per_launch = algo_bytes / (dom_launches / prof_passes)
        achieved = algo_per_launch / (dom_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath) and args.files == 10000 and args.file_bytes == 262144:
            # PMC pass of this same command, collected separately (tools_prof.sh); null if not measured
            traffic = json.load(open(tpath)).get(dom.replace("lz_", "lz_"), {}).get("hbm_bytes_per_launch")
        line = {
            "metric": "compress+decompress GB/s (raw bytes / (deflate + inflate time)), .zwz bit-exact",
            "value": round(world * raw_bytes * args.steps / elapsed / 1e9, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d x %d B %s files per GPU -> %d chunks of <=65535 B (BASELINE configs[%d])"
                                   % (args.files, args.file_bytes, args.workload, n, 1 if args.workload == "random" else 2),
                       "chunk_bytes": CHUNK, "max_batch_chunks": args.max_batch, "parallelism": "shard-per-gpu x%d" % world},
            "compress_GBps": round(raw_bytes / (c1 - c0) / 1e9, 3), "decompress_GBps": round(raw_bytes / (c2 - c1) / 1e9, 3),
            "payload_ratio": round(payload_bytes / raw_bytes, 4), "roundtrip_property_ok": ok,
            "stage_ms_per_pass": {k: round(v / prof_passes, 3) for k, v in stage.items()},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(algo_per_launch), "launch_ms": round(dom_ms, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.file_bytes, args.cpu_sample_files)
        print(json.dumps(line), flush=True)
    codec.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
