"""ctypes declarations for libxhe.so (include/xhe.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class XheError(RuntimeError):
    pass


def lib_path():
    """The in-tree library, or the build XHE_LIBRARY names (A/B comparisons of two builds on one box)."""
    return os.environ.get("XHE_LIBRARY") or os.path.join(_HERE, "libxhe.so")


_lib = None


class EcdlpStats(ctypes.Structure):
    _fields_ = [("instances", ctypes.c_uint64), ("steps", ctypes.c_uint64), ("walk_ms", ctypes.c_double), ("lanes", ctypes.c_uint64)]


def load_library():
    """Loads libxhe.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise XheError("libxhe.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
    lib = ctypes.CDLL(path)
    u8p, u64p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint64)
    lib.xhe_create.restype = ctypes.c_void_p
    lib.xhe_create.argtypes = [ctypes.c_int]
    lib.xhe_destroy.argtypes = [ctypes.c_void_p]
    lib.xhe_last_error.restype = ctypes.c_char_p
    lib.xhe_last_error.argtypes = [ctypes.c_void_p]
    lib.xhe_msm_vartime.restype = ctypes.c_int32
    lib.xhe_msm_vartime.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p, u8p, u8p]
    lib.xhe_msm_vartime_static.restype = ctypes.c_int32
    lib.xhe_msm_vartime_static.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p, u8p, u8p]
    lib.xhe_ristretto_decompress_batch.restype = ctypes.c_int32
    lib.xhe_ristretto_decompress_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, u8p, u8p]
    lib.xhe_ristretto_compress_batch.restype = ctypes.c_int32
    lib.xhe_ristretto_compress_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, u8p]
    lib.xhe_last_stage_times.restype = ctypes.c_int32
    lib.xhe_last_stage_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float), ctypes.c_int32]
    lib.xhe_set_stage_timing.restype = ctypes.c_int32
    lib.xhe_set_stage_timing.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    i64p = ctypes.POINTER(ctypes.c_int64)
    lib.xhe_ecdlp_table_create.restype = ctypes.c_int32
    lib.xhe_ecdlp_table_create.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
    lib.xhe_ecdlp_table_free.restype = None
    lib.xhe_ecdlp_table_free.argtypes = [ctypes.c_void_p]
    lib.xhe_ecdlp_decode.restype = ctypes.c_int32
    lib.xhe_ecdlp_decode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, i64p, u8p]
    lib.xhe_elgamal_decrypt_decode.restype = ctypes.c_int32
    lib.xhe_elgamal_decrypt_decode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, i64p, u8p]
    lib.xhe_ecdlp_last_stats.restype = ctypes.c_int32
    lib.xhe_ecdlp_last_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(EcdlpStats)]
    _lib = lib
    return lib


class Engine:
    """One GPU, one HIP stream (xhe_ctx)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.ctx = self.lib.xhe_create(int(device))
        if not self.ctx:
            raise XheError("xhe_create failed: %s" % self.lib.xhe_last_error(None).decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.xhe_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise XheError("libxhe error %d: %s" % (rc, self.lib.xhe_last_error(self.ctx).decode()))

    # level 0 -------------------------------------------------------------------------------------
    def msm_vartime(self, scalars32: bytes, points32: bytes):
        """RistrettoPoint::vartime_multiscalar_mul.  Returns (encoding, is_identity)."""
        assert len(scalars32) == len(points32) and len(scalars32) % 32 == 0
        n = len(scalars32) // 32
        out, ident = (ctypes.c_uint8 * 32)(), (ctypes.c_uint8 * 1)()
        self._check(self.lib.xhe_msm_vartime(self.ctx, n, scalars32, points32, out, ident))
        return bytes(out), bool(ident[0])

    def msm_vartime_static(self, scalars32: bytes, points32: bytes):
        """xhe_msm_vartime_static: the same operator from the library's static device array (no allocation, bounded size)."""
        assert len(scalars32) == len(points32) and len(scalars32) % 32 == 0
        n = len(scalars32) // 32
        out, ident = (ctypes.c_uint8 * 32)(), (ctypes.c_uint8 * 1)()
        self._check(self.lib.xhe_msm_vartime_static(self.ctx, n, scalars32, points32, out, ident))
        return bytes(out), bool(ident[0])

    def decompress_batch(self, in32: bytes):
        n = len(in32) // 32
        out, ok = (ctypes.c_uint8 * (128 * max(n, 1)))(), (ctypes.c_uint8 * max(n, 1))()
        self._check(self.lib.xhe_ristretto_decompress_batch(self.ctx, n, in32, out, ok))
        return bytes(out)[:128 * n], bytes(ok)[:n]

    def compress_batch(self, xyzt128: bytes):
        n = len(xyzt128) // 128
        out = (ctypes.c_uint8 * (32 * max(n, 1)))()
        self._check(self.lib.xhe_ristretto_compress_batch(self.ctx, n, xyzt128, out))
        return bytes(out)[:32 * n]

    def set_stage_timing(self, level):
        """2: every stage, 1: the bucket accumulation and the total only, 0: nothing."""
        self._check(self.lib.xhe_set_stage_timing(self.ctx, int(level)))

    def stage_times(self):
        names = (ctypes.c_char_p * 64)()
        ms = (ctypes.c_float * 64)()
        n = self.lib.xhe_last_stage_times(self.ctx, names, ms, 64)
        return [(names[i].decode(), float(ms[i])) for i in range(max(n, 0))]


class EcdlpTable:
    """xhe_ecdlp_table: the baby-step table of the balance decoder on one GPU (the place of ecdlp::ECDLPTablesFileView,
    src/elgamal.rs:79).  2^(table_bits + 4) bytes of device memory."""

    def __init__(self, engine: Engine, table_bits=24):
        self.engine, self.lib, self.table_bits = engine, engine.lib, int(table_bits)
        h = ctypes.c_void_p()
        engine._check(self.lib.xhe_ecdlp_table_create(engine.ctx, self.table_bits, ctypes.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.xhe_ecdlp_table_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _out(self, n):
        return (ctypes.c_int64 * max(n, 1))(), (ctypes.c_uint8 * max(n, 1))()

    def decode(self, points32: bytes, max_bits: int):
        """ECDLPInstance::decode over a batch of encodings: a list of amounts, None where the point is no m*G with m < 2^max_bits."""
        assert len(points32) % 32 == 0
        n = len(points32) // 32
        amounts, status = self._out(n)
        self.engine._check(self.lib.xhe_ecdlp_decode(self.handle, points32, n, int(max_bits), amounts, status))
        return [None if amounts[i] < 0 else int(amounts[i]) for i in range(n)]

    def decrypt_decode(self, secret32: bytes, cts64: bytes, max_bits: int):
        """ElGamalSecretKey::decrypt followed by decode for ciphertexts under one key."""
        assert len(cts64) % 64 == 0 and len(secret32) == 32
        n = len(cts64) // 64
        amounts, status = self._out(n)
        self.engine._check(self.lib.xhe_elgamal_decrypt_decode(self.handle, secret32, cts64, n, int(max_bits), amounts, status))
        return [None if amounts[i] < 0 else int(amounts[i]) for i in range(n)]

    def last_stats(self):
        st = EcdlpStats()
        self.lib.xhe_ecdlp_last_stats(self.handle, ctypes.byref(st))
        return {"instances": int(st.instances), "steps": int(st.steps), "walk_ms": float(st.walk_ms), "lanes": int(st.lanes)}
