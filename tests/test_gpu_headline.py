"""The HIP path on the benchmark's largest single-GPU configuration (100 000 transactions, 8 transfers each, aggregated range
proofs over 16 value commitments; 10.4 M multiscalar terms, window width 15, two-pass bucket sort), checked through
properties that need no CPU verifier at that size, plus oracle parity on the same shape at sizes the oracle finishes in
seconds: 14 000 transactions (1.46 M terms: the two-pass bucket sort with 14-bit windows) and a 1.6 M-term multiscalar
multiplication (15-bit windows, the plan of the full batch).  The batches come from the GPU-assisted generator, itself pinned
to the CPU prover byte for byte in test_gpu_prover.py."""
import ctypes
import hashlib

import pytest

from helpers import buf
from test_gpu_fullsize import _RawView, _verify
from xelis_he_amd import codes
from xelis_he_amd.api import LedgerOut, SyntheticBatch, Verdict, _lib

pytestmark = pytest.mark.gpu
N_TXS, TRANSFERS = 100000, 8


@pytest.fixture(scope="module")
def engine():
    import xelis_he_amd
    e = xelis_he_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def batch(engine):
    sb = SyntheticBatch(N_TXS, n_transfers=TRANSFERS, n_senders=1024, seed=b"headline", threads=16, engine=engine)
    yield sb
    sb.close()


@pytest.mark.timeout(900)
def test_headline_accept_and_ledger(engine, batch, switch):
    """Accepted, and the resulting ledger equals the CPU apply path's (Transaction::apply_without_verify per transaction).
    A batch of this size is verified in consecutive parts inside the call: the batch in one piece gives the same bytes."""
    v, cts = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    assert v.accept == 1, codes.NAMES.get(v.err_code)
    switch("XHE_SPLIT_PARTS", "1")
    v1, cts1 = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    switch("XHE_SPLIT_PARTS", None)
    assert v1.accept == 1 and cts1 == cts
    lib = _lib()
    view = ctypes.cast(batch.view, ctypes.POINTER(_RawView)).contents
    cur = ctypes.create_string_buffer(ctypes.string_at(view.balance_cts, 64 * view.n_balances), 64 * view.n_balances)
    keys = ctypes.string_at(view.account_keys, 32 * view.n_accounts)
    assets = ctypes.string_at(view.balance_assets, 32 * view.n_balances)
    bal_acc = ctypes.cast(view.balance_account, ctypes.POINTER(ctypes.c_uint32))
    # the apply path on every 8th sender's chain (12 500 transactions): the senders' chains are independent
    offs = batch.offsets
    for i in range(batch.n_txs):
        if (i % 1024) % 8:
            continue
        rec = batch.records[offs[i]:offs[i + 1]]
        assert lib.xhe_tx_apply_host(rec, len(rec), view.n_accounts, keys, view.n_balances, bal_acc, assets, cur) == 0, i
    for s in range(0, 1024, 8):
        assert cts[128 * s:128 * s + 128] == cur.raw[128 * s:128 * s + 128], s       # sender and receiver slots of pair s
    # and the GPU's own ledger-only path over the whole batch gives the same bytes everywhere
    lib.xhe_batch_upload.restype = lib.xhe_batch_apply.restype = lib.xhe_batch_read_ledger.restype = ctypes.c_int32
    ctx, h, av = ctypes.c_void_p(engine.ctx), ctypes.c_void_p(), Verdict()
    engine._check(lib.xhe_batch_upload(ctx, batch.records, batch.offsets, ctypes.c_uint64(batch.n_txs), batch.view, ctypes.byref(h), ctypes.byref(av)))
    try:
        engine._check(lib.xhe_batch_apply(ctx, h, ctypes.byref(av)))
        assert av.accept == 1
        acts = (ctypes.c_uint8 * (64 * view.n_balances))()
        out = LedgerOut(acts, (ctypes.c_uint64 * view.n_accounts)(), (ctypes.c_int64 * view.n_accounts)())
        engine._check(lib.xhe_batch_read_ledger(ctx, h, ctypes.byref(out)))
    finally:
        lib.xhe_batch_free(ctx, h)
    assert bytes(acts) == cts


@pytest.mark.timeout(900)
def test_headline_against_the_oracle_on_every_host_thread(engine, batch, oracle):
    """The whole 100 000-transaction batch through the CPU restatement: cut along its account groups into 32 pieces (the
    senders' chains are independent), each piece verified by the oracle against the whole ledger on a host thread of its own
    (about 1.2 ms per transaction and thread).  Every piece is accepted, and the ledger the pieces leave - every slot from the
    piece that touched it - equals the GPU verification's ledger byte for byte."""
    from helpers import oracle_verify_in_pieces
    v, cts = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    assert v.accept == 1, codes.NAMES.get(v.err_code)
    verdicts, merged = oracle_verify_in_pieces(oracle, batch)
    assert all(pv == (1, codes.OK, -1) for pv in verdicts)
    assert merged == cts


@pytest.mark.timeout(900)
def test_headline_one_bad_signature(engine, batch):
    idx = 77777
    rec = bytearray(batch.records)
    rec[batch.offsets[idx + 1] - 40] ^= 0x10
    v, _ = _verify(engine, bytes(rec), batch.offsets, batch.n_txs, batch.view)
    assert (v.accept, v.err_code, v.first_bad_tx) == (0, codes.ERR_SIGNATURE, idx)


@pytest.mark.timeout(900)
def test_headline_chain_gap_and_wrong_range_proof(engine, batch):
    drop = 5
    offs = [batch.offsets[i] for i in range(batch.n_txs + 1)]
    cut = offs[drop + 1] - offs[drop]
    records = batch.records[:offs[drop]] + batch.records[offs[drop + 1]:]
    new_offs = offs[:drop + 1] + [o - cut for o in offs[drop + 2:]]
    offsets = (ctypes.c_uint64 * len(new_offs))(*new_offs)
    v, _ = _verify(engine, records, offsets, batch.n_txs - 1, batch.view)
    assert (v.accept, v.err_code, v.first_bad_tx) == (0, codes.ERR_GENERIC_PROOF, -1)


@pytest.mark.timeout(900)
def test_eight_transfers_against_the_oracle(engine, oracle):
    """14 000 transactions of the headline shape (1.46 M dynamic terms): verdict and every ledger byte equal the oracle's, and a
    corrupted transfer ciphertext deep in the batch is reported with the oracle's code and index."""
    sb = SyntheticBatch(14000, n_transfers=8, n_senders=256, seed=b"headline-oracle", threads=16, engine=engine)
    n_bal, n_acc = sb.view.contents.n_balances, sb.view.contents.n_accounts

    def run(ctx_engine, records):
        cts = (ctypes.c_uint8 * (64 * n_bal))()
        out = LedgerOut(cts, (ctypes.c_uint64 * n_acc)(), (ctypes.c_int64 * n_acc)())
        v = Verdict()
        if ctx_engine is None:
            oracle.xo_verify_batch(records, sb.offsets, ctypes.c_uint64(sb.n_txs), sb.view, ctypes.byref(out), ctypes.byref(v), b"\x02" * 32)
        else:
            ctx_engine._check(ctx_engine.lib.xhe_verify_batch(ctx_engine.ctx, records, sb.offsets, sb.n_txs, sb.view, ctypes.byref(out), ctypes.byref(v), b"\x01" * 32))
        return (v.accept, v.err_code, v.first_bad_tx), bytes(cts)

    got, cts = run(engine, sb.records)
    ref, ref_cts = run(None, sb.records)
    assert got == ref == (1, codes.OK, -1)
    assert cts == ref_cts
    bad = bytearray(sb.records)
    bad[sb.offsets[13013] + 16 + 32 + 49 + 320 * 5 + 100] ^= 1       # D_s of the sixth transfer of transaction 13 013
    got, _ = run(engine, bytes(bad))
    ref, _ = run(None, bytes(bad))
    assert got == ref and got[0] == 0 and got[2] == 13013
    sb.close()


@pytest.mark.timeout(900)
def test_msm_parity_at_window_width_15(engine, oracle):
    """1.6 M terms: the widest window (15 bits, 16 384 buckets per window) and the two-pass bucket sort, against the oracle."""
    n = 1600000
    base = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
    stream = hashlib.shake_256(b"msm-1.6M").digest(32 * 64 + 32 * n)
    pool = []
    for i in range(64):
        o, idf = buf(32), buf(1)
        oracle.xo_msm_vartime(1, stream[32 * i:32 * i + 32], base, o, idf)
        pool.append(bytes(o))
    pts = b"".join(pool[(i * 7 + i // 64) % 64] for i in range(n))
    scal = stream[32 * 64:]
    got, ident = engine.msm_vartime(scal, pts)
    o, idf = buf(32), buf(1)
    assert oracle.xo_msm_vartime(n, scal, pts, o, idf) == 0
    assert got == bytes(o) and ident == bool(idf[0])
