"""xhe_verify_batches (upload of batch k + 1 beside the verification of batch k, inside the library) against xhe_verify_batch
batch by batch: verdicts, failing indices and ledger bytes are those of the single calls, for accepted, rejected and malformed
batches in one run."""
import ctypes

import pytest

from xelis_he_amd import codes
from xelis_he_amd.api import LedgerOut, SyntheticBatch, Verdict

pytestmark = pytest.mark.gpu


class BatchArgs(ctypes.Structure):
    _fields_ = [("records", ctypes.c_char_p), ("offsets", ctypes.POINTER(ctypes.c_uint64)), ("n_txs", ctypes.c_uint64), ("view", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("seed", ctypes.c_char_p)]


def test_pipelined_batches_equal_single_calls():
    import xelis_he_amd
    engine = xelis_he_amd.Engine(0)
    lib = engine.lib
    lib.xhe_verify_batches.restype = ctypes.c_int32
    try:
        sbs = [SyntheticBatch(3000, n_transfers=1, n_senders=64, seed=b"pipe-a", engine=engine), SyntheticBatch(256, seed=b"pipe-b", mixed=True),
               SyntheticBatch(700, n_transfers=3, n_senders=9, seed=b"pipe-c", engine=engine), SyntheticBatch(40, n_transfers=8, n_senders=5, seed=b"pipe-d", engine=engine)]
        runs = []          # (batch, records)
        for sb in sbs:
            runs.append((sb, sb.records))
        bad_sig = bytearray(sbs[0].records); bad_sig[sbs[0].offsets[1501] - 30] ^= 4
        runs.append((sbs[0], bytes(bad_sig)))
        runs.append((sbs[2], sbs[2].records[:-1] + b"\x00"))                  # last signature byte changed
        trunc = SyntheticBatch(10, n_transfers=1, n_senders=2, seed=b"pipe-e")
        offs = (ctypes.c_uint64 * 11)(*[trunc.offsets[i] for i in range(10)] + [trunc.offsets[10] - 1])
        runs.append((sbs[1], sbs[1].records))
        n = len(runs) + 1
        arr, outs, keep = (BatchArgs * n)(), [], []
        for i, (sb, recs) in enumerate(runs):
            n_bal, n_acc = sb.view.contents.n_balances, sb.view.contents.n_accounts
            cts = (ctypes.c_uint8 * (64 * n_bal))()
            out = LedgerOut(cts, (ctypes.c_uint64 * n_acc)(), (ctypes.c_int64 * n_acc)())
            outs.append((cts, out))
            arr[i].records, arr[i].offsets, arr[i].n_txs = recs, sb.offsets, sb.n_txs
            arr[i].view, arr[i].out, arr[i].seed = ctypes.cast(sb.view, ctypes.c_void_p), ctypes.cast(ctypes.pointer(out), ctypes.c_void_p), bytes([i + 1]) * 32
        # a malformed batch (its last record is cut short) in the middle of the run's tail
        arr[n - 1].records, arr[n - 1].offsets, arr[n - 1].n_txs = trunc.records, offs, trunc.n_txs
        arr[n - 1].view, arr[n - 1].out, arr[n - 1].seed = ctypes.cast(trunc.view, ctypes.c_void_p), None, b"\x09" * 32
        vs = (Verdict * n)()
        engine._check(lib.xhe_verify_batches(ctypes.c_void_p(engine.ctx), ctypes.c_uint64(n), arr, vs))
        for i, (sb, recs) in enumerate(runs):
            n_bal, n_acc = sb.view.contents.n_balances, sb.view.contents.n_accounts
            cts = (ctypes.c_uint8 * (64 * n_bal))()
            out = LedgerOut(cts, (ctypes.c_uint64 * n_acc)(), (ctypes.c_int64 * n_acc)())
            v = Verdict()
            engine._check(lib.xhe_verify_batch(engine.ctx, recs, sb.offsets, sb.n_txs, sb.view, ctypes.byref(out), ctypes.byref(v), bytes([i + 1]) * 32))
            assert (vs[i].accept, vs[i].err_code, vs[i].first_bad_tx) == (v.accept, v.err_code, v.first_bad_tx), i
            if v.accept:
                assert bytes(outs[i][0]) == bytes(cts), i
        assert [vs[i].accept for i in range(4)] == [1, 1, 1, 1]
        assert (vs[4].accept, vs[4].err_code, vs[4].first_bad_tx) == (0, codes.ERR_SIGNATURE, 1500)
        assert (vs[5].accept, vs[5].err_code, vs[5].first_bad_tx) == (0, codes.ERR_SIGNATURE, 699)
        assert (vs[n - 1].accept, vs[n - 1].err_code, vs[n - 1].first_bad_tx) == (0, codes.ERR_MALFORMED, 9)
    finally:
        engine.close()
