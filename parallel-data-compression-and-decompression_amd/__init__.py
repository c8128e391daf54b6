"""MI355X-native chunk codec: Python host mirror of the reference's C++ seam (process.hpp:37-42)
over the C ABI in include/zwz.h (libzwz_hip.so, hand-written HIP for gfx950).

Names and argument meaning follow the reference:
    sort_files_by_size(path)                                   file_sort.cpp:24
    count_non_empty_lines(file_path)                           file_tools.cpp:6
    do_compression(input_dir, output_dir, file_record, rank)   compression.cpp:161
    do_decompression(input_dir, output_dir)                    decompression.cpp:165
    md5_of_file(file_path)                                     verification.cpp:6
plus the batch form of the two zlib call sites (compression.cpp:119-134, decompression.cpp:16-36):
    Codec.deflate_chunks / Codec.inflate_chunks                (host bytes)
    Codec.deflate_dev / Codec.inflate_dev                      (device-resident torch tensors)

There is no CPU fallback: importing works anywhere (so the build can be checked), but every codec
call needs the HIP library and a GPU and raises ZwzError otherwise.
"""
import ctypes
import os

CHUNK_SIZE = 65535          # process.hpp:12
DEV_STRIDE = 65536
NUM_STAGES = 7
STAGE_NAMES = ("lz_links", "lz_match", "lz_parse", "blockify", "plan", "encode", "inflate")

LOSSLESS_CHUNK_SIZE = 65509  # opt-in, never default (include/zwz.h: zwz_ctx_set_chunk_size)
# zwz_allgather_u64_fn: int (*)(void *user, const uint64_t *mine, uint64_t *all, uint32_t count)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                ctypes.c_uint32)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ZWZ_LIB: another build of the same library -- tools/gpu.sh's instrumented libzwz_hip_exp.so; such a build announces itself on stderr)
LIB_PATH = os.environ.get("ZWZ_LIB") or os.path.join(_HERE, "libzwz_hip.so")
_lib = None


class ZwzError(RuntimeError):
    """A non-zero status from the C ABI.  .status is the zwz_status code; a failed do_decompression also carries
    .md5_mismatches (a damaged shard is decoded up to the damage before ZWZ_E_FORMAT is returned)."""
    status = None
    md5_mismatches = None


def lib():
    """The C-ABI library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZwzError("libzwz_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or make -C %s); this package has no CPU fallback" % _HERE)
        L = ctypes.CDLL(LIB_PATH)
        c = ctypes
        vp, u32, u64 = c.c_void_p, c.c_uint32, c.c_uint64
        L.zwz_strerror.restype = c.c_char_p
        L.zwz_strerror.argtypes = [c.c_int]
        L.zwz_last_error.restype = c.c_char_p
        L.zwz_device_count.argtypes = [c.POINTER(c.c_int)]
        L.zwz_ctx_create.argtypes = [c.c_int, u32, c.POINTER(vp)]
        L.zwz_ctx_destroy.argtypes = [vp]
        L.zwz_ctx_destroy.restype = None
        L.zwz_ctx_stream.argtypes = [vp]
        L.zwz_ctx_stream.restype = vp
        L.zwz_ctx_sync.argtypes = [vp]
        L.zwz_deflate_batch_dev.argtypes = [vp, vp, vp, vp, u32, vp, u64, vp]
        L.zwz_inflate_batch_dev.argtypes = [vp, vp, vp, vp, u32, vp, u64, vp, vp]
        L.zwz_deflate_batch.argtypes = [vp, vp, vp, vp, u32, vp, vp]
        L.zwz_inflate_batch.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp]
        L.zwz_ctx_set_profiling.argtypes = [vp, c.c_int]
        L.zwz_ctx_stage_ms.argtypes = [vp, c.POINTER(c.c_float), c.c_int]
        L.zwz_sort_files_by_size.argtypes = [c.c_char_p, c.c_char_p, c.c_size_t]
        L.zwz_count_non_empty_lines.argtypes = [c.c_char_p]
        L.zwz_md5_of_file.argtypes = [c.c_char_p, c.c_char_p]
        L.zwz_md5_files_dev.argtypes = [vp, vp, vp, vp, vp, u32, vp]
        L.zwz_compress_dir.argtypes = [vp, c.c_char_p, c.c_char_p, c.c_char_p, c.c_int, c.c_int]
        L.zwz_decompress_dir.argtypes = [vp, c.c_char_p, c.c_char_p, c.POINTER(c.c_int)]
        L.zwz_decompress_dir_ranked.argtypes = [vp, c.c_char_p, c.c_char_p, c.c_int, c.c_int, ALLGATHER_FN, vp, c.POINTER(c.c_int)]
        L.zwz_ctx_set_chunk_size.argtypes = [vp, u32]
        L.zwz_ctx_set_option.argtypes = [vp, c.c_char_p, c.c_char_p]
        _lib = L
    return _lib


E_FORMAT = -6


def _check(rc, what, **extra):
    if rc != 0:
        L = lib()
        err = ZwzError("%s: %s (%s)" % (what, L.zwz_strerror(rc).decode(), L.zwz_last_error().decode()))
        err.status = rc
        for k, v in extra.items():
            setattr(err, k, v)
        raise err


def device_count():
    n = ctypes.c_int(0)
    lib().zwz_device_count(ctypes.byref(n))
    return n.value


class Codec:
    """One GPU, one HIP stream, one workspace (zwz_ctx)."""

    def __init__(self, device=0, max_batch_chunks=0):
        self._h = ctypes.c_void_p()
        _check(lib().zwz_ctx_create(device, max_batch_chunks, ctypes.byref(self._h)), "zwz_ctx_create")
        self.device = device

    def close(self):
        if self._h:
            lib().zwz_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def stream(self):
        return lib().zwz_ctx_stream(self._h)

    def sync(self):
        _check(lib().zwz_ctx_sync(self._h), "zwz_ctx_sync")

    def set_profiling(self, on):
        _check(lib().zwz_ctx_set_profiling(self._h, int(on)), "zwz_ctx_set_profiling")

    def stage_ms(self, reset=True):
        arr = (ctypes.c_float * NUM_STAGES)()
        _check(lib().zwz_ctx_stage_ms(self._h, arr, int(reset)), "zwz_ctx_stage_ms")
        return dict(zip(STAGE_NAMES, list(arr)))

    # ---- host bytes ---------------------------------------------------------------------------
    def deflate_chunks(self, chunks):
        """[bytes <= 65535] -> [payload bytes]: what consumer() stores per Chunk (compression.cpp:118-134)."""
        import numpy as np
        n = len(chunks)
        blob = b"".join(chunks)
        lens = np.array([len(c) for c in chunks], dtype=np.uint32)
        offs = np.zeros(n, dtype=np.uint64)
        if n:
            offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        src = np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(1, dtype=np.uint8)
        out = np.empty(max(n, 1) * CHUNK_SIZE, dtype=np.uint8)
        olen = np.zeros(max(n, 1), dtype=np.uint32)
        _check(lib().zwz_deflate_batch(self._h, src.ctypes.data, offs.ctypes.data, lens.ctypes.data, n, out.ctypes.data,
                                       olen.ctypes.data), "zwz_deflate_batch")
        return [out[i * CHUNK_SIZE:i * CHUNK_SIZE + int(olen[i])].tobytes() for i in range(n)]

    def inflate_chunks(self, payloads):
        """[payload bytes] -> ([decoded bytes], [status]): decompress_chunk() per record (decompression.cpp:11-37)."""
        import numpy as np
        n = len(payloads)
        blob = b"".join(payloads)
        lens = np.array([len(c) for c in payloads], dtype=np.uint32)
        offs = np.zeros(n, dtype=np.uint64)
        if n:
            offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        src = np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(1, dtype=np.uint8)
        out = np.empty(max(n, 1) * CHUNK_SIZE, dtype=np.uint8)
        olen = np.zeros(max(n, 1), dtype=np.uint32)
        st = np.zeros(max(n, 1), dtype=np.uint32)
        _check(lib().zwz_inflate_batch(self._h, src.ctypes.data, offs.ctypes.data, lens.ctypes.data, n, out.ctypes.data,
                                       olen.ctypes.data, st.ctypes.data), "zwz_inflate_batch")
        return ([out[i * CHUNK_SIZE:i * CHUNK_SIZE + int(olen[i])].tobytes() for i in range(n)], [int(s) for s in st[:n]])

    def md5_files_dev(self, d_in, d_off, d_len, d_files, d_digests):
        """MD5 of whole files from their chunk slots in device memory: d_files = int32 pairs (first slot, slots),
        d_digests = 16 bytes per file (md5_of_file(), verification.cpp:6-30, batched).  Asynchronous."""
        n_files = d_files.numel() // 2
        _check(lib().zwz_md5_files_dev(self._h, d_in.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), d_files.data_ptr(), n_files,
                                       d_digests.data_ptr()), "zwz_md5_files_dev")

    # ---- device-resident torch tensors (asynchronous on self.stream) ---------------------------
    # self.stream is the context's own non-blocking HIP stream, NOT torch's current stream: work torch has queued on the
    # tensors (a torch.zeros fill, a copy) must be complete before these calls -- torch.cuda.synchronize(), or an event --
    # and self.sync() must precede any torch read of the results.
    def deflate_dev(self, d_in, d_off, d_len, d_out, d_out_len, out_stride=DEV_STRIDE):
        n = d_len.numel()
        _check(lib().zwz_deflate_batch_dev(self._h, d_in.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n, d_out.data_ptr(),
                                           out_stride, d_out_len.data_ptr()), "zwz_deflate_batch_dev")

    def inflate_dev(self, d_in, d_off, d_len, d_out, d_out_len, d_status, out_stride=DEV_STRIDE):
        n = d_len.numel()
        _check(lib().zwz_inflate_batch_dev(self._h, d_in.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n, d_out.data_ptr(),
                                           out_stride, d_out_len.data_ptr(), d_status.data_ptr()), "zwz_inflate_batch_dev")

    # ---- directory level -----------------------------------------------------------------------
    def do_compression(self, input_dir, output_dir, file_record, world_rank, world_size=1):
        _check(lib().zwz_compress_dir(self._h, os.fsencode(input_dir), os.fsencode(output_dir), os.fsencode(file_record),
                                      world_rank, world_size), "zwz_compress_dir")

    def do_decompression(self, input_dir, output_dir, world_rank=0, world_size=1, allgather=None):
        """do_decompression(), decompression.cpp:165-178, shared by world_size ranks (one GPU each): whole shards round-robin,
        or record ranges of a shard when there are fewer shards than ranks.  `allgather(values) -> list of every rank's
        values, rank-major` is the launcher's exchange (cli.py passes torch.distributed's); returns the number of files
        this rank found with a wrong MD5."""
        bad = ctypes.c_int(0)

        def _cb(_user, mine, out, count):
            try:
                flat = allgather([int(mine[i]) for i in range(count)])
                for i, v in enumerate(flat):
                    out[i] = int(v)
                return 0
            except Exception:          # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return -1

        cb = ALLGATHER_FN(_cb) if allgather is not None else ctypes.cast(None, ALLGATHER_FN)
        rc = lib().zwz_decompress_dir_ranked(self._h, os.fsencode(input_dir), os.fsencode(output_dir), world_rank, world_size,
                                             cb, None, ctypes.byref(bad))
        _check(rc, "zwz_decompress_dir", md5_mismatches=bad.value)
        return bad.value

    def set_option(self, name, value):
        """Test / experiment switches of this context (include/zwz.h: zwz_ctx_set_option): "match" = auto | walk | band | lazy |
        autoband | autolazy, "plan" = wave | serial, "inflate_header" = wave | serial.  Every choice produces the same bytes; a form that
        failed its self-test on this device is refused."""
        _check(lib().zwz_ctx_set_option(self._h, name.encode(), value.encode()), "zwz_ctx_set_option")

    def set_chunk_size(self, nbytes):
        """Raw bytes per Chunk for do_compression (0 = the reference's 65535).  Opt-in, not bit-exact with the reference's
        shards; LOSSLESS_CHUNK_SIZE never truncates."""
        _check(lib().zwz_ctx_set_chunk_size(self._h, nbytes), "zwz_ctx_set_chunk_size")


def sort_files_by_size(path):
    buf = ctypes.create_string_buffer(4096)
    _check(lib().zwz_sort_files_by_size(os.fsencode(path), buf, len(buf)), "zwz_sort_files_by_size")
    return buf.value.decode()


def count_non_empty_lines(file_path):
    return lib().zwz_count_non_empty_lines(os.fsencode(file_path))


def md5_of_file(file_path):
    buf = ctypes.create_string_buffer(33)
    lib().zwz_md5_of_file(os.fsencode(file_path), buf)
    return buf.value.decode()


_default = None


def _codec():
    global _default
    if _default is None:
        _default = Codec(int(os.environ.get("LOCAL_RANK", "0")))
    return _default


def do_compression(input_dir, output_dir, file_record, world_rank, world_size=1):
    _codec().do_compression(input_dir, output_dir, file_record, world_rank, world_size)


def do_decompression(input_dir, output_dir, world_rank=0, world_size=1, allgather=None):
    return _codec().do_decompression(input_dir, output_dir, world_rank, world_size, allgather)
