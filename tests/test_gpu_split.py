"""xhe_verify_batch cuts a large batch into consecutive parts (upload of part k + 1 beside the verification of part k, every
part verified against the ledger the part before it left).  The cut must not be observable: verdict, error variant, index
of the failing transaction and every ledger byte equal the oracle's and the uncut call's.  XHE_SPLIT_PARTS forces the cut
on small batches (it is read on every call)."""
import ctypes
import os
import random

import pytest

from xelis_he_amd import codes
from xelis_he_amd.api import LedgerOut, SyntheticBatch, Verdict, _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import xelis_he_amd
    e = xelis_he_amd.Engine(0)
    yield e
    e.close()


class _parts:
    def __init__(self, k):
        self.k = k

    def __enter__(self):
        self.before = os.environ.get("XHE_SPLIT_PARTS")
        os.environ["XHE_SPLIT_PARTS"] = str(self.k)
        _lib().xhe_settings_reload()            # the library reads its switches once

    def __exit__(self, *a):
        if self.before is None:
            del os.environ["XHE_SPLIT_PARTS"]
        else:
            os.environ["XHE_SPLIT_PARTS"] = self.before
        _lib().xhe_settings_reload()


def _verify(engine, oracle, view, records, offsets, n_txs, gpu):
    n_bal, n_acc = view.contents.n_balances, view.contents.n_accounts
    cts = (ctypes.c_uint8 * (64 * n_bal))()
    ms = (ctypes.c_int64 * n_acc)()
    nonces = (ctypes.c_uint64 * n_acc)()
    out = LedgerOut(cts, nonces, ms)
    v = Verdict()
    if gpu:
        engine._check(engine.lib.xhe_verify_batch(engine.ctx, records, offsets, n_txs, view, ctypes.byref(out), ctypes.byref(v), b"\x07" * 32))
    else:
        oracle.xo_verify_batch(records, offsets, ctypes.c_uint64(n_txs), view, ctypes.byref(out), ctypes.byref(v), b"\x02" * 32)
    return (v.accept, v.err_code, v.first_bad_tx), bytes(cts), list(ms), list(nonces)


def test_cut_batch_chains_the_ledger(engine, oracle):
    """Four senders whose transactions follow each other through the whole batch: every part starts from balances the part
    before it wrote.  2, 3, 5 and 96 parts (one transaction each) give the uncut call's and the oracle's ledger."""
    sb = SyntheticBatch(96, n_transfers=2, n_senders=4, seed=b"split-chain")
    ref = _verify(engine, oracle, sb.view, sb.records, sb.offsets, sb.n_txs, gpu=False)
    assert ref[0] == (1, codes.OK, -1)
    with _parts(1):
        whole = _verify(engine, oracle, sb.view, sb.records, sb.offsets, sb.n_txs, gpu=True)
    assert whole == ref
    for k in (2, 3, 5, 96):
        with _parts(k):
            got = _verify(engine, oracle, sb.view, sb.records, sb.offsets, sb.n_txs, gpu=True)
        assert got == ref, k


def test_cut_batch_random_corruptions(engine, oracle):
    """Seeded random corruptions of a chained batch verified in three parts: the verdict, the error variant and the index equal
    the oracle's whichever part the damage lies in (a malformed record in a later part comes before an error in an earlier
    one, a wrong proof in an earlier part behind a per-transaction error in a later one)."""
    sb = SyntheticBatch(48, n_transfers=2, n_senders=4, seed=b"split-fuzz")
    offs = list(sb.offsets)
    r = random.Random(1114)
    seen = {}
    with _parts(3):
        for case in range(300):
            bad = bytearray(sb.records)
            for _ in range(r.choice((1, 1, 2, 3))):
                tx = r.randrange(sb.n_txs)
                lo, hi = offs[tx], offs[tx + 1]
                for _ in range(r.choice((1, 1, 2))):
                    pos = r.randrange(lo, hi) if r.random() < 0.8 else r.randrange(lo, min(hi, lo + 80))
                    bad[pos] ^= 1 << r.randrange(8)
            g = _verify(engine, oracle, sb.view, bytes(bad), sb.offsets, sb.n_txs, gpu=True)
            ref = _verify(engine, oracle, sb.view, bytes(bad), sb.offsets, sb.n_txs, gpu=False)
            assert g[0] == ref[0], (case, g[0], ref[0])
            if g[0][0] == 1:
                assert g == ref, case
            seen[g[0][1]] = seen.get(g[0][1], 0) + 1
    print("verdict codes reached:", dict(sorted(seen.items())))
    assert len(seen) >= 4, seen


def test_cut_batch_with_multisig_updates(engine, oracle):
    """A multisig set-up decides which keys later co-signatures are checked against.  In a part that is not the last it makes
    the call start over uncut (the mixed batch: set-ups at the head of every group of 16); in the last part it is part of the
    result, reported at its index in the whole batch."""
    sb = SyntheticBatch(64, seed=b"split-mixed", mixed=True)
    ref = _verify(engine, oracle, sb.view, sb.records, sb.offsets, sb.n_txs, gpu=False)
    assert ref[0] == (1, codes.OK, -1) and max(ref[2]) >= 0
    with _parts(4):
        got = _verify(engine, oracle, sb.view, sb.records, sb.offsets, sb.n_txs, gpu=True)
    assert got == ref
    # one group reordered: the sender's twelve transactions first, the two set-ups (other accounts) last, no co-signed transfers
    offs = list(sb.offsets)
    order = list(range(2, 14)) + [0, 1]
    blob = b"".join(sb.records[offs[i]:offs[i + 1]] for i in order)
    new_offs = [0]
    for i in order:
        new_offs.append(new_offs[-1] + offs[i + 1] - offs[i])
    c_offs = (ctypes.c_uint64 * len(new_offs))(*new_offs)
    ref = _verify(engine, oracle, sb.view, blob, c_offs, len(order), gpu=False)
    assert ref[0] == (1, codes.OK, -1) and sorted(x for x in ref[2] if x >= 0) == [12, 13]
    for k in (2, 3):
        with _parts(k):
            got = _verify(engine, oracle, sb.view, blob, c_offs, len(order), gpu=True)
        assert got == ref, k


def test_cut_batch_whose_later_part_needs_larger_tables(oracle):
    """The first part holds range proofs over 2 values, the second over 16: its upload has to rebuild the context's generator
    tables, which waits for the verification in flight -- the call must finish that part rather than wait for the upload.
    A context of its own, so that the tables start small."""
    import xelis_he_amd
    from helpers import oracle_verify_batch
    from xelis_he_amd.api import NATIVE, Account, Ledger, Transaction, TransactionBuilder, TransferBuilder
    small = Account([(NATIVE, 1000)], "grow-small")
    big = Account([(NATIVE, 1000)], "grow-big")
    sink = Account([(NATIVE, 0)], "grow-sink")
    ledger = Ledger([small, big, sink])
    txs, state = [], ledger.clone()
    for i in range(3):
        tx = TransactionBuilder(1, small.pubkey, ("transfers", [TransferBuilder(sink.pubkey, 5)]), 1, 0).build(
            {NATIVE: 1000 - 6 * i}, small.keypair, "grow-s%d" % i, state.get_account(small.pubkey))
        tx.apply_without_verify(state)
        txs.append(tx)
    for i in range(2):
        tx = TransactionBuilder(1, big.pubkey, ("transfers", [TransferBuilder(sink.pubkey, 1 + k) for k in range(8)]), 1, 0).build(
            {NATIVE: 1000 - 37 * i}, big.keypair, "grow-b%d" % i, state.get_account(big.pubkey))
        tx.apply_without_verify(state)
        txs.append(tx)
    ref_ledger = ledger.clone()
    assert oracle_verify_batch(oracle, txs, ref_ledger) == (True, codes.OK, -1)
    eng = xelis_he_amd.Engine(0)
    try:
        for k in (2, 5):
            got = ledger.clone()
            with _parts(k):
                Transaction.verify_batch(eng, txs, got, seed="grow")
            for key, a in ref_ledger.accounts.items():
                assert got.accounts[key].balances == a.balances, k
            eng.close()
            eng = xelis_he_amd.Engine(0)          # cold tables again
    finally:
        eng.close()


def test_cut_batch_reports_usage_errors(engine, oracle):
    """A ledger view whose balance names an account that does not exist is a usage error of the call, cut or not; the context
    stays usable afterwards."""
    from xelis_he_amd.api import LedgerView, XheError
    sb = SyntheticBatch(32, n_transfers=1, n_senders=2, seed=b"split-usage")
    src = sb.view.contents
    bad = LedgerView()
    ctypes.memmove(ctypes.byref(bad), ctypes.byref(src), ctypes.sizeof(LedgerView))
    accounts = (ctypes.c_uint32 * src.n_balances)(*[src.balance_account[i] for i in range(src.n_balances)])
    accounts[src.n_balances - 1] = src.n_accounts + 7
    bad.balance_account = accounts
    for k in (1, 4):
        with _parts(k):
            with pytest.raises(XheError):
                _verify(engine, oracle, ctypes.pointer(bad), sb.records, sb.offsets, sb.n_txs, gpu=True)
            got = _verify(engine, oracle, sb.view, sb.records, sb.offsets, sb.n_txs, gpu=True)
        assert got[0] == (1, codes.OK, -1), k
