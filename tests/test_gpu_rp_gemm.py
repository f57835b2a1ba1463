"""The range proofs' generator scalars (sums over all proofs of a batch) computed on the matrix cores (xhe_k_rpgemm.hip) against
the element-by-element kernel: same 32-byte scalars for every generator, same verdicts.  The sums are exact integers either
way, so the comparison is bit for bit.  XHE_RP_GEMM_MIN_ROWS is read when a batch is uploaded."""
import ctypes
import os

import pytest

from xelis_he_amd import codes
from xelis_he_amd.api import SyntheticBatch, Verdict, _lib

pytestmark = pytest.mark.gpu
XHE_DEBUG_STATIC_SCALARS = 3


@pytest.fixture(scope="module")
def engine():
    import xelis_he_amd
    e = xelis_he_amd.Engine(0)
    yield e
    e.close()


def _verify_and_read(engine, records, offsets, n_txs, view, min_rows, n_gen):
    lib = _lib()
    lib.xhe_batch_upload.restype = lib.xhe_batch_verify.restype = ctypes.c_int32
    lib.xhe_batch_debug_read.restype = ctypes.c_int64
    old = os.environ.get("XHE_RP_GEMM_MIN_ROWS")
    os.environ["XHE_RP_GEMM_MIN_ROWS"] = str(min_rows)
    lib.xhe_settings_reload()          # the library reads its switches once; this one is used when a batch is uploaded
    ctx, h, v = ctypes.c_void_p(engine.ctx), ctypes.c_void_p(), Verdict()
    try:
        engine._check(lib.xhe_batch_upload(ctx, records, offsets, ctypes.c_uint64(n_txs), view, ctypes.byref(h), ctypes.byref(v)))
    finally:
        if old is None:
            del os.environ["XHE_RP_GEMM_MIN_ROWS"]
        else:
            os.environ["XHE_RP_GEMM_MIN_ROWS"] = old
        lib.xhe_settings_reload()
    try:
        engine._check(lib.xhe_batch_verify(ctx, h, ctypes.byref(v), bytes(range(32))))
        out = (ctypes.c_uint8 * (32 * (2 + 2 * n_gen)))()
        got = lib.xhe_batch_debug_read(ctx, h, XHE_DEBUG_STATIC_SCALARS, out, ctypes.c_uint64(len(out)))
        assert got == 2 + 2 * n_gen
    finally:
        lib.xhe_batch_free(ctx, h)
    return (v.accept, v.err_code, v.first_bad_tx), bytes(out)


# (transactions, transfers per transaction, generators per side = 64 * values rounded up to a power of two)
CASES = [(700, 1, 128), (300, 3, 256), (1100, 8, 1024), (40000, 1, 128), (96, 20, 2048)]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_txs,transfers,n_gen", CASES)
def test_matrix_path_equals_element_path(engine, n_txs, transfers, n_gen):
    sb = SyntheticBatch(n_txs, n_transfers=transfers, n_senders=min(64, n_txs), seed=b"rp-gemm-%d-%d" % (n_txs, transfers), threads=16, engine=engine)
    try:
        verdict_m, scalars_m = _verify_and_read(engine, sb.records, sb.offsets, sb.n_txs, sb.view, 1, n_gen)
        verdict_e, scalars_e = _verify_and_read(engine, sb.records, sb.offsets, sb.n_txs, sb.view, 1 << 40, n_gen)
    finally:
        sb.close()
    assert verdict_e == (1, codes.OK, -1)
    assert verdict_m == verdict_e
    assert any(scalars_e[64:])
    for i in range(2 + 2 * n_gen):
        assert scalars_m[32 * i:32 * i + 32] == scalars_e[32 * i:32 * i + 32], ("generator scalar", i)


@pytest.mark.timeout(600)
def test_matrix_path_rejects_a_wrong_proof(engine):
    """One byte of one range proof changed (transaction 333): both paths give the same rejection, and the sums still agree
    (they do not depend on the proof being right)."""
    sb = SyntheticBatch(600, n_transfers=2, n_senders=32, seed=b"rp-gemm-bad", threads=16, engine=engine)
    try:
        rec = bytearray(sb.records)
        # the range proof ends 64 bytes (signature) before the record's end; its last 32 bytes are the scalar b
        end = sb.offsets[334] - 64
        rec[end - 32] ^= 1
        rec = bytes(rec)
        verdict_m, scalars_m = _verify_and_read(engine, rec, sb.offsets, sb.n_txs, sb.view, 1, 256)
        verdict_e, scalars_e = _verify_and_read(engine, rec, sb.offsets, sb.n_txs, sb.view, 1 << 40, 256)
    finally:
        sb.close()
    assert verdict_m == verdict_e
    assert verdict_e[0] == 0
    assert scalars_m == scalars_e
