"""The HIP path at the benchmark's full batch size (10 000 transactions, 1 transfer each), checked through
properties that need no CPU verifier at that size: a valid batch is accepted and its ledger equals the one the
CPU apply path (Transaction::apply_without_verify, no proofs checked) produces; one corrupted signature is
reported at its index; a broken balance chain fails the aggregate sigma check; the multiscalar multiplication
is linear at the term count the batch produces."""
import ctypes
import random

import pytest

from helpers import le32
from xelis_he_amd import codes
from xelis_he_amd.api import LedgerOut, SyntheticBatch, Verdict, _lib

pytestmark = pytest.mark.gpu
L = 2**252 + 27742317777372353535851937790883648493
N_TXS = 10000


class _RawView(ctypes.Structure):
    _fields_ = [("n_accounts", ctypes.c_uint64), ("account_keys", ctypes.c_void_p), ("account_nonces", ctypes.c_void_p),
                ("n_balances", ctypes.c_uint64), ("balance_account", ctypes.c_void_p), ("balance_assets", ctypes.c_void_p), ("balance_cts", ctypes.c_void_p),
                ("n_multisig", ctypes.c_uint64), ("multisig_account", ctypes.c_void_p), ("multisig_threshold", ctypes.c_void_p),
                ("multisig_offsets", ctypes.c_void_p), ("multisig_signers", ctypes.c_void_p)]


@pytest.fixture(scope="module")
def engine():
    import xelis_he_amd
    e = xelis_he_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def batch():
    sb = SyntheticBatch(N_TXS, n_transfers=1, n_senders=1024, seed=b"full-size", threads=16)
    yield sb
    sb.close()


def _verify(engine, records, offsets, n, view, want_ledger=False):
    lib = _lib()
    n_bal, n_acc = view.contents.n_balances, view.contents.n_accounts
    out_cts = (ctypes.c_uint8 * (64 * n_bal))()
    out = LedgerOut(out_cts, (ctypes.c_uint64 * n_acc)(), (ctypes.c_int64 * n_acc)())
    v = Verdict()
    engine._check(lib.xhe_verify_batch(engine.ctx, records, offsets, n, view, ctypes.byref(out), ctypes.byref(v), bytes(range(32))))
    return v, bytes(out_cts)


@pytest.mark.timeout(600)
def test_full_batch_accept_and_ledger(engine, batch):
    v, cts = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    assert v.accept == 1, codes.NAMES.get(v.err_code)
    # CPU apply path over the same records, one transaction after the other
    lib = _lib()
    view = ctypes.cast(batch.view, ctypes.POINTER(_RawView)).contents      # the same struct with untyped pointers (c_char_p would stop at a NUL)
    cur = ctypes.create_string_buffer(ctypes.string_at(view.balance_cts, 64 * view.n_balances), 64 * view.n_balances)
    keys = ctypes.string_at(view.account_keys, 32 * view.n_accounts)
    assets = ctypes.string_at(view.balance_assets, 32 * view.n_balances)
    bal_acc = ctypes.cast(view.balance_account, ctypes.POINTER(ctypes.c_uint32))
    for i in range(batch.n_txs):
        rec = batch.records[batch.offsets[i]:batch.offsets[i + 1]]
        assert lib.xhe_tx_apply_host(rec, len(rec), view.n_accounts, keys, view.n_balances, bal_acc, assets, cur) == 0, i
    assert cts == cur.raw[:64 * view.n_balances]


@pytest.mark.timeout(600)
def test_full_batch_one_bad_signature(engine, batch):
    idx = 7777
    rec = bytearray(batch.records)
    rec[batch.offsets[idx + 1] - 40] ^= 0x10          # inside s || e of transaction idx
    v, _ = _verify(engine, bytes(rec), batch.offsets, batch.n_txs, batch.view)
    assert (v.accept, v.err_code, v.first_bad_tx) == (0, codes.ERR_SIGNATURE, idx)


@pytest.mark.timeout(600)
def test_full_batch_chain_gap(engine, batch):
    # drop one early transaction: its sender's later transactions prove against a balance the ledger never reaches
    drop = 5
    offs = [batch.offsets[i] for i in range(batch.n_txs + 1)]
    cut = offs[drop + 1] - offs[drop]
    records = batch.records[:offs[drop]] + batch.records[offs[drop + 1]:]
    new_offs = offs[:drop + 1] + [o - cut for o in offs[drop + 2:]]
    offsets = (ctypes.c_uint64 * len(new_offs))(*new_offs)
    v, _ = _verify(engine, records, offsets, batch.n_txs - 1, batch.view)
    assert (v.accept, v.err_code, v.first_bad_tx) == (0, codes.ERR_GENERIC_PROOF, -1)


@pytest.mark.timeout(600)
def test_full_batch_with_term_index_entries(engine, batch, switch):
    """The bucket sort normally names point records in its entries; batches of 2^24 points and more keep term indices and the
    accumulation looks the point up.  That form, forced here, gives the same verdicts and the same ledger."""
    v0, cts0 = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    switch("XHE_MSM_TERM_ENTRIES", "1")
    v1, cts1 = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    assert (v0.accept, v1.accept) == (1, 1)
    assert cts0 == cts1
    rec = bytearray(batch.records)
    rec[batch.offsets[4000] + 200] ^= 1               # inside transaction 4000's proofs
    v2, _ = _verify(engine, bytes(rec), batch.offsets, batch.n_txs, batch.view)
    switch("XHE_MSM_TERM_ENTRIES", None)
    v3, _ = _verify(engine, bytes(rec), batch.offsets, batch.n_txs, batch.view)
    assert v2.accept == 0 and (v2.accept, v2.err_code, v2.first_bad_tx) == (v3.accept, v3.err_code, v3.first_bad_tx)


@pytest.mark.timeout(600)
def test_full_batch_with_one_pass_sort(engine, batch, switch):
    """The one-pass bucket sort at 14-bit windows (the form inputs beyond 2^24 terms take, all 2^13 counters of a window in
    LDS), forced on the 350 000-term batch: same verdicts, same ledger."""
    v0, cts0 = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    switch("XHE_MSM_ONEPASS", "1")
    v1, cts1 = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    assert (v0.accept, v1.accept) == (1, 1)
    assert cts0 == cts1
    rec = bytearray(batch.records)
    rec[batch.offsets[123] + 300] ^= 2
    v2, _ = _verify(engine, bytes(rec), batch.offsets, batch.n_txs, batch.view)
    switch("XHE_MSM_ONEPASS", None)
    v3, _ = _verify(engine, bytes(rec), batch.offsets, batch.n_txs, batch.view)
    assert v2.accept == 0 and (v2.accept, v2.err_code, v2.first_bad_tx) == (v3.accept, v3.err_code, v3.first_bad_tx)


@pytest.mark.timeout(600)
def test_full_batch_with_register_accumulation(engine, batch, switch):
    """The bucket accumulation keeps its prefetched record in LDS; the form with the record in registers (the one entries
    that name terms take), forced here: same verdicts, same ledger."""
    v0, cts0 = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    switch("XHE_MSM_ACC_STAGED", "0")
    v1, cts1 = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    assert (v0.accept, v1.accept) == (1, 1) and cts0 == cts1
    rec = bytearray(batch.records)
    rec[batch.offsets[2500] + 400] ^= 4
    v2, _ = _verify(engine, bytes(rec), batch.offsets, batch.n_txs, batch.view)
    switch("XHE_MSM_ACC_STAGED", None)
    v3, _ = _verify(engine, bytes(rec), batch.offsets, batch.n_txs, batch.view)
    assert v2.accept == 0 and (v2.accept, v2.err_code, v2.first_bad_tx) == (v3.accept, v3.err_code, v3.first_bad_tx)


@pytest.mark.timeout(600)
def test_full_batch_cut_into_parts(engine, batch, switch):
    """Batches from 128 MiB of records on are verified in consecutive parts, each against the ledger the part before it left
    (1 024 senders here, every one with transactions in every part).  Forced on this batch with 4 and 7 parts: the ledger of
    the batch in one piece, a bad signature at its index in the whole batch, a broken chain as GenericProof."""
    switch("XHE_SPLIT_PARTS", "1")
    v0, cts0 = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    assert v0.accept == 1
    rec = bytearray(batch.records)
    rec[batch.offsets[7777 + 1] - 40] ^= 0x10
    offs = [batch.offsets[i] for i in range(batch.n_txs + 1)]
    drop = 5
    cut = offs[drop + 1] - offs[drop]
    gap_records = batch.records[:offs[drop]] + batch.records[offs[drop + 1]:]
    gap_offsets = (ctypes.c_uint64 * batch.n_txs)(*(offs[:drop + 1] + [o - cut for o in offs[drop + 2:]]))
    for parts in ("4", "7"):
        switch("XHE_SPLIT_PARTS", parts)
        v1, cts1 = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
        assert v1.accept == 1 and cts1 == cts0, parts
        v2, _ = _verify(engine, bytes(rec), batch.offsets, batch.n_txs, batch.view)
        assert (v2.accept, v2.err_code, v2.first_bad_tx) == (0, codes.ERR_SIGNATURE, 7777), parts
        v3, _ = _verify(engine, gap_records, gap_offsets, batch.n_txs - 1, batch.view)
        assert (v3.accept, v3.err_code, v3.first_bad_tx) == (0, codes.ERR_GENERIC_PROOF, -1), parts


@pytest.mark.timeout(600)
def test_full_batch_without_term_merging(engine, batch, switch):
    """Terms that share a point are merged before the bucket sort; unmerged (XHE_MERGE_TERMS=0) the check is the same sum."""
    switch("XHE_MERGE_TERMS", "0")
    v1, cts1 = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    rec = bytearray(batch.records)
    rec[batch.offsets[9000] + 500] ^= 8
    v2, _ = _verify(engine, bytes(rec), batch.offsets, batch.n_txs, batch.view)
    switch("XHE_MERGE_TERMS", None)
    v0, cts0 = _verify(engine, batch.records, batch.offsets, batch.n_txs, batch.view)
    v3, _ = _verify(engine, bytes(rec), batch.offsets, batch.n_txs, batch.view)
    assert (v0.accept, v1.accept) == (1, 1) and cts0 == cts1
    assert v2.accept == 0 and (v2.accept, v2.err_code, v2.first_bad_tx) == (v3.accept, v3.err_code, v3.first_bad_tx)


@pytest.mark.timeout(600)
def test_msm_linear_at_batch_term_count(engine):
    n = 350258                                        # terms of the 10 000-transaction check
    r = random.Random(350258)
    base = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
    pool = [engine.msm_vartime(le32(r.randrange(L)), base)[0] for _ in range(64)]
    pts = b"".join(pool[r.randrange(64)] for _ in range(n))
    scal = [r.randrange(L) for _ in range(n)]
    k = r.randrange(L)
    r1, _ = engine.msm_vartime(b"".join(le32(s) for s in scal), pts)
    r2, _ = engine.msm_vartime(b"".join(le32(s * k % L) for s in scal), pts)
    r3, _ = engine.msm_vartime(le32(k), r1)
    assert r2 == r3
    # and the sum is what the pool's discrete logs say: sum s_i * P_i = (sum s_i * k_i) * B is not available here
    # (the k_i were not kept), so close with additivity instead: msm(s, P) + msm(t, P) = msm(s + t, P)
    t = [r.randrange(L) for _ in range(n)]
    r4, _ = engine.msm_vartime(b"".join(le32(x) for x in t), pts)
    r5, _ = engine.msm_vartime(b"".join(le32((a + b) % L) for a, b in zip(scal, t)), pts)
    r6, _ = engine.msm_vartime(le32(1) + le32(1), r1 + r4)
    assert r5 == r6
