#!/usr/bin/env python3
"""Experiment (round 4): do the deflate stages of two batches overlap on one GPU when they come from two contexts (two HIP streams)?
lz_match_band keeps the vector unit at ~0.77 instructions a cycle and holds 154 KB of LDS and 480 of a SIMD's 512 registers; the other stages are
small kernels.  If a second stream's lz_parse / encode waves could run in the band kernel's shadow, two staggered streams would finish sooner than
twice one.  Run on the GPU box: python tools/exp/overlap_two_contexts.py [files]"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, workloads
zwz = importlib.import_module("parallel-data-compression-and-decompression_amd")
files = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
dev = torch.device("cuda", 0)
jobs = []
for r in range(2):
    d_in, d_off, d_len, n, raw, _ = workloads.build_equal_files(torch, dev, "text", files, 262144, r)
    codec = zwz.Codec(0, 51200)
    d_out = torch.empty(n * 65536, dtype=torch.uint8, device=dev); d_olen = torch.zeros(n, dtype=torch.int32, device=dev)
    jobs.append((codec, d_in, d_off, d_len, d_out, d_olen, raw))
torch.cuda.synchronize()
def run(j, reps):
    codec, d_in, d_off, d_len, d_out, d_olen, _ = j
    for _ in range(reps): codec.deflate_dev(d_in, d_off, d_len, d_out, d_olen)
    codec.sync()
for j in jobs: run(j, 1)                                  # warm-up (workspace allocation)
t0 = time.perf_counter(); run(jobs[0], 3); t1 = time.perf_counter(); run(jobs[1], 3); t2 = time.perf_counter()
ths = [threading.Thread(target=run, args=(j, 3)) for j in jobs]
t3 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]; t4 = time.perf_counter()
raw = jobs[0][6]
print("one context at a time: %.1f + %.1f ms per 3 passes; both at once: %.1f ms  (%.1f %% of the sum); %d chunks each"
      % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 1e3, 100.0 * (t4 - t3) / (t2 - t0), jobs[0][3].numel()))
