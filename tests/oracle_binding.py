"""ctypes binding of the CPU oracle (oracle/zwz_oracle.h).  Test infrastructure only."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
CHUNK = 65535


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        c = ctypes
        lib.zo_deflate6.restype = c.c_size_t
        lib.zo_deflate6.argtypes = [c.c_char_p, c.c_size_t, c.c_char_p, c.c_size_t]
        lib.zo_chunk_payload.restype = c.c_uint32
        lib.zo_chunk_payload.argtypes = [c.c_char_p, c.c_uint32, c.c_char_p]
        lib.zo_inflate.restype = c.c_size_t
        lib.zo_inflate.argtypes = [c.c_char_p, c.c_size_t, c.c_char_p, c.c_size_t, c.POINTER(c.c_int)]
        lib.zo_lz77_symbols.restype = c.c_size_t
        lib.zo_lz77_symbols.argtypes = [c.c_char_p, c.c_size_t, c.c_void_p, c.c_void_p]
        lib.zo_adler32.restype = c.c_uint32
        lib.zo_adler32.argtypes = [c.c_char_p, c.c_size_t]
        lib.zo_md5_hex.restype = None
        lib.zo_md5_hex.argtypes = [c.c_char_p, c.c_size_t, c.c_char_p]
        lib.zo_compress_shard.restype = c.c_int
        lib.zo_compress_shard.argtypes = [c.c_char_p, c.c_char_p, c.c_char_p, c.c_int, c.c_int]
        lib.zo_decompress_shard.restype = c.c_int
        lib.zo_decompress_shard.argtypes = [c.c_char_p, c.c_char_p]

    def deflate6(self, data: bytes) -> bytes:
        cap = len(data) + len(data) // 100 + 4096
        out = ctypes.create_string_buffer(cap)
        n = self.lib.zo_deflate6(data, len(data), out, cap)
        assert n > 0
        return out.raw[:n]

    def payload(self, chunk: bytes) -> bytes:
        assert len(chunk) <= CHUNK
        out = ctypes.create_string_buffer(CHUNK)
        n = self.lib.zo_chunk_payload(chunk, len(chunk), out)
        return out.raw[:n]

    def inflate(self, payload: bytes, cap: int = 1 << 20):
        out = ctypes.create_string_buffer(cap)
        st = ctypes.c_int()
        n = self.lib.zo_inflate(payload, len(payload), out, cap, ctypes.byref(st))
        return out.raw[:min(n, cap)], n, st.value

    def symbols(self, data: bytes):
        import numpy as np
        d = np.zeros(max(len(data), 1), dtype=np.uint16)
        l = np.zeros(max(len(data), 1), dtype=np.uint8)
        k = self.lib.zo_lz77_symbols(data, len(data), d.ctypes.data, l.ctypes.data)
        return d[:k].copy(), l[:k].copy()

    def adler32(self, data: bytes) -> int:
        return self.lib.zo_adler32(data, len(data))

    def md5_hex(self, data: bytes) -> str:
        out = ctypes.create_string_buffer(33)
        self.lib.zo_md5_hex(data, len(data), out)
        return out.value.decode()

    def compress_shard(self, in_dir, out_dir, record_file, rank, nranks) -> int:
        return self.lib.zo_compress_shard(in_dir.encode(), out_dir.encode(), record_file.encode(), rank, nranks)

    def decompress_shard(self, shard, out_dir) -> int:
        return self.lib.zo_decompress_shard(shard.encode(), out_dir.encode())


def load() -> Oracle:
    so = os.path.join(ORACLE_DIR, "libzwz_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libzwz_oracle.so"], stdout=subprocess.DEVNULL)
    return Oracle(ctypes.CDLL(so))
