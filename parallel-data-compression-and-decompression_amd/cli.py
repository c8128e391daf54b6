"""`python -m <package>.cli compress|decompress <src> <dst>` -- the reference's command line
(main.cpp:78-159) for a one-process-per-GPU launch under torchrun.

What the reference does with MPI maps onto torch.distributed (backend "nccl" = RCCL
over xGMI when GPUs are present, "gloo" otherwise):

    MPI_Bcast of the record-file *path* (main.cpp:27,35), contents via a shared file system
        -> broadcast of the sorted list *contents* from rank 0; every rank writes its own copy, so
           ranks need no shared file system for the list
    rank -> files: line i of the list goes to rank i mod N (compression.cpp:38-41), one shard per rank
        -> unchanged; no data-path collective
    MPI_Barrier + MPI_Wtime banner (main.cpp:144-155)
        -> dist.barrier() + the same banner text
    (addition, --gather) per-shard .zwz blobs collected on rank 0 for hosts without a shared output
        directory; the reference has no gather (every rank writes compressed_<rank>.zwz itself).  The protocol is
        the library's zwz_gather_shards -- the code csrc/main.cpp drives over RCCL -- with torch.distributed as its transport

    (addition) decompression is a single-rank job in the reference (main.cpp:61-68); here shard j goes to rank j mod N,
        and fewer shards than ranks are split by record ranges with an all_gather of decoded byte counts
"""
import argparse
import os
import sys
import tempfile
import time


def _dist():
    import torch.distributed as dist
    return dist


def broadcast_bytes(data, src=0):
    """Rank src's bytes on every rank (length first, then payload)."""
    import torch
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return data
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    n = torch.tensor([len(data) if dist.get_rank() == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if dist.get_rank() == src:
        buf.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
    if buf.numel():
        dist.broadcast(buf, src)
    return bytes(buf.cpu().numpy().tobytes())


def gather_shards(my_shard_path, out_dir, piece_bytes=0):
    """Every rank's shard file on rank 0 as <out_dir>/compressed_<r>.zwz: the library's protocol (include/zwz.h: zwz_gather_shards --
    the same code csrc/main.cpp drives over RCCL; sizes and readiness by all-gather, bounded pieces, one send for one receive, an I/O
    failure never leaves a send unmatched) with torch.distributed's send / recv / all_gather as its hooks.  my_shard_path: "" = this
    rank contributes nothing.  True only if every rank saw every transfer and every write succeed."""
    import ctypes
    import torch
    from . import lib
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return True
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    AG = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32)
    SEND = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int)
    RECV = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int)
    PREP = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64)
    REL = ctypes.CFUNCTYPE(None, ctypes.c_void_p)

    class Hooks(ctypes.Structure):
        _fields_ = [("user", ctypes.c_void_p), ("allgather_u64", AG), ("send", SEND), ("recv", RECV), ("prepare", PREP), ("release", REL)]

    def ag(_, mine, out, count):
        try:
            got = allgather_u64([mine[i] for i in range(count)])
            for i, v in enumerate(got):
                out[i] = v
            return 0
        except Exception:
            return -1

    def send(_, buf, n, to):
        try:
            t = torch.frombuffer((ctypes.c_ubyte * n).from_address(buf), dtype=torch.uint8).clone().to(dev)
            dist.send(t, to)
            return 0
        except Exception:
            return -1

    def recv(_, buf, n, src):
        try:
            t = torch.empty(n, dtype=torch.uint8, device=dev)
            dist.recv(t, src)
            h = t.cpu().contiguous()
            ctypes.memmove(buf, h.data_ptr(), n)
            return 0
        except Exception:
            return -1

    hooks = Hooks(None, AG(ag), SEND(send), RECV(recv), PREP(lambda _, n: 0), REL(lambda _: None))
    fn = lib().zwz_gather_shards
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.POINTER(Hooks)]
    return fn(rank, world, (my_shard_path or "").encode(), out_dir.encode(), piece_bytes, ctypes.byref(hooks)) == 1


def allgather_u64(values):
    """The exchange zwz_decompress_dir_ranked asks for (include/zwz.h: zwz_allgather_u64_fn): every rank's list of unsigned
    64-bit values on every rank, rank-major.  Doubles as a barrier."""
    import torch
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(values)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    signed = [v - (1 << 64) if v >= (1 << 63) else v for v in values]           # torch has no uint64 collectives
    mine = torch.tensor(signed, dtype=torch.int64, device=dev)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [int(x) & ((1 << 64) - 1) for p in parts for x in p.cpu().tolist()]


def run(operation, source_path, output_path, *, compress_fn=None, decompress_fn=None, sort_fn=None, count_fn=None,
        gather=False, out=sys.stdout):
    """The reference's main() flow.  *_fn default to the GPU codec; tests inject stand-ins to
    exercise the orchestration without a GPU."""
    start = time.perf_counter()
    dist = _dist()
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if compress_fn is None or decompress_fn is None or sort_fn is None or count_fn is None:
        from . import count_non_empty_lines, do_compression, do_decompression, sort_files_by_size
        compress_fn = compress_fn or do_compression
        decompress_fn = decompress_fn or do_decompression
        sort_fn = sort_fn or sort_files_by_size
        count_fn = count_fn or count_non_empty_lines

    source_path, output_path = source_path.rstrip("/") or source_path, output_path.rstrip("/") or output_path  # main.cpp:72-76
    print("source_path: %s\noutput_path: %s" % (source_path, output_path), file=out)
    if operation not in ("compress", "decompress"):
        print("Invalid operation: %s. Please use 'compress' or 'decompress'." % operation, file=sys.stderr)
        return 1
    if rank == 0:                                                   # main.cpp:105-129
        if not os.path.exists(source_path):
            print("Source path does not exist.", file=sys.stderr)
            return 1
        if not os.path.exists(output_path):
            os.mkdir(output_path, 0o777)
        elif not os.path.isdir(output_path):
            print("Output path is not a directory.", file=sys.stderr)
            return 1
    if world > 1:
        dist.barrier()

    rc = 0
    if operation == "compress":
        listing = b""
        if rank == 0:
            print("Compressing folder: %s" % source_path, file=out)
            record = os.environ.get("ZWZ_FILE_RECORD") or sort_fn(source_path)
            print("File record saved location: %s" % record, file=out)
            with open(record, "rb") as f:
                listing = f.read()
        listing = broadcast_bytes(listing, 0)
        with tempfile.NamedTemporaryFile(prefix="zwz_list_r%d_" % rank, suffix=".txt", delete=False) as f:
            f.write(listing)
            local_record = f.name
        failure = None
        try:
            shard_dir = output_path
            if gather and world > 1 and rank != 0:
                shard_dir = tempfile.mkdtemp(prefix="zwz_shard_r%d_" % rank)
            try:
                if rank < count_fn(local_record):                   # main.cpp:44-51
                    compress_fn(source_path, shard_dir, local_record, rank, world)
                else:
                    print("Rank: %d - No file to compress" % rank, file=out)
            except Exception as e:      # a failed rank still joins the collectives below: its peers have no time-out there
                failure = e
            if gather and world > 1:
                path = os.path.join(shard_dir, "compressed_%d.zwz" % rank)
                # (a rank that failed hands over nothing -- a partially written shard must not reach <dst> under a good name; csrc/main.cpp does the same)
                mine = path if rank != 0 and failure is None and os.path.exists(path) else ""
                if not gather_shards(mine, output_path):
                    # the shard stays where the rank wrote it: nothing is lost, and the message says where it is
                    if failure is None:
                        failure = RuntimeError("the gather of the shards failed" + ("; this rank's shard is kept at %s" % path if mine else ""))
                elif mine and shard_dir != output_path:
                    os.unlink(path)
                    os.rmdir(shard_dir)
        finally:
            os.unlink(local_record)
    else:
        # The reference decodes on rank 0 only (main.cpp:61-68); here every rank takes its share: whole shards round-robin,
        # or record ranges of a shard when there are fewer shards than ranks.
        failure = None
        try:
            decompress_fn(source_path, output_path, rank, world, allgather_u64 if world > 1 else None)
        except Exception as e:          # (zwz_decompress_dir_ranked itself carries a rank's failure through its exchanges)
            failure = e

    if world > 1:
        # MPI_Barrier (main.cpp:144) and the job's status in one: every rank learns whether any rank failed
        bad = allgather_u64([1 if failure is not None else 0])
        if failure is None and any(bad):
            failure = RuntimeError("rank(s) %s failed" % ", ".join(str(r) for r, b in enumerate(bad) if b))
    if failure is not None:
        print("zwz: %s" % failure, file=sys.stderr)
        rc = 2
    if rank == 0:                                                    # main.cpp:148-155
        print("========================================\nOperation: %s\nProcessor Count: %d\nTime Taken: %g seconds\n"
              "========================================" % (operation, world, time.perf_counter() - start), file=out)
    return rc


def main(argv=None):
    ap = argparse.ArgumentParser(prog="main", usage="%(prog)s <compress/decompress> <source directory path> <output directory path>")
    ap.add_argument("operation")
    ap.add_argument("source")
    ap.add_argument("output")
    ap.add_argument("--gather", action="store_true", help="collect every rank's shard on rank 0 (RCCL send/recv)")
    args = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch
        dist = _dist()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl")
        else:
            dist.init_process_group("gloo")
    try:
        return run(args.operation, args.source, args.output, gather=args.gather)
    finally:
        if world > 1:
            _dist().destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
