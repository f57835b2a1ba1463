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


def test_call_level_error_leaves_no_unverified_accept():
    """A call-level error in the middle of a run (a ledger view whose balance names an account that does not exist: XHE_E_ARG)
    ends the call; the batches in front of it keep their verdicts, every batch from the failing one on reads reject -- an
    unverified batch never reads accept -- and the error names the batch."""
    import xelis_he_amd
    from xelis_he_amd.api import LedgerView
    engine = xelis_he_amd.Engine(0)
    lib = engine.lib
    lib.xhe_verify_batches.restype = ctypes.c_int32
    try:
        sbs = [SyntheticBatch(600, n_transfers=1, n_senders=16, seed=b"err-%d" % i, engine=engine) for i in range(5)]
        bad = 2
        n = len(sbs)
        arr, keep = (BatchArgs * n)(), []
        for i, sb in enumerate(sbs):
            view = sb.view
            if i == bad:
                src = sb.view.contents
                v = LedgerView()
                for name, _ in LedgerView._fields_:
                    setattr(v, name, getattr(src, name))
                acc = (ctypes.c_uint32 * src.n_balances)(*[src.balance_account[j] for j in range(src.n_balances)])
                acc[src.n_balances - 1] = src.n_accounts + 7            # out of range
                v.balance_account = acc
                keep += [v, acc]
                view = ctypes.pointer(v)
            arr[i].records, arr[i].offsets, arr[i].n_txs = sb.records, sb.offsets, sb.n_txs
            arr[i].view, arr[i].out, arr[i].seed = ctypes.cast(view, ctypes.c_void_p), None, bytes([i + 1]) * 32
        vs = (Verdict * n)()
        for i in range(n):
            vs[i].accept = 1                                             # whatever the caller's array held before
        rc = lib.xhe_verify_batches(ctypes.c_void_p(engine.ctx), ctypes.c_uint64(n), arr, vs)
        assert rc == -1                                                  # XHE_E_ARG
        assert b"batch %d" % bad in lib.xhe_last_error(engine.ctx)
        for i in range(bad):
            assert vs[i].accept in (0, 1)                                # verified to the end, or rejected as unverified
        assert vs[0].accept == 1                                         # long finished when batch 2's upload fails
        for i in range(bad, n):
            assert (vs[i].accept, vs[i].err_code, vs[i].first_bad_tx) == (0, codes.ERR_STATE, -1), i
        # the context is still usable
        v = Verdict()
        engine._check(lib.xhe_verify_batch(engine.ctx, sbs[0].records, sbs[0].offsets, sbs[0].n_txs, sbs[0].view, None, ctypes.byref(v), b"\x05" * 32))
        assert v.accept == 1
    finally:
        engine.close()
