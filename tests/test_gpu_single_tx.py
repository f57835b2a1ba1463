"""The single-transaction row: Transaction::verify (/root/reference/src/tx/verify.rs:520-542, a batch of one behind the C ABI)
against the oracle's verify_single -- accept and every class of rejection -- and Transaction::apply_without_verify
(:545-619) on the GPU (xhe_batch_apply) against the oracle's own apply path, not against the product's host apply."""
import ctypes

import pytest

from helpers import oracle_verify_batch
from test_gpu_verify import ASSET55, _mutate, _resigned
from xelis_he_amd import codes
from xelis_he_amd.api import (NATIVE, Account, Ledger, LedgerOut, SyntheticBatch, Transaction, TransactionBuilder, TransferBuilder, VerificationError)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import xelis_he_amd
    e = xelis_he_amd.Engine(0)
    yield e
    e.close()


def _single_same(engine, oracle, tx, ledger, expect):
    lg, lo = ledger.clone(), ledger.clone()
    try:
        tx.verify(engine, lg, seed="single")
        got = (True, codes.OK, -1)
    except VerificationError as e:
        got = (False, e.code, e.first_bad_tx)
    ref = oracle_verify_batch(oracle, [tx], lo, single=True)
    # the reference's single verify has no batch index; an error inside the one transaction is index 0 on both sides or -1 for
    # the aggregate checks
    assert got[:2] == ref[:2] == expect, (got, ref, expect)
    if got[0]:
        for k, a in lo.accounts.items():
            assert lg.accounts[k].balances == a.balances
        assert lg.multisig_accounts == lo.multisig_accounts
    return lg


def test_transaction_verify_against_verify_single(engine, oracle):
    alice = Account([(NATIVE, 100), (ASSET55, 30)], "one-a")
    bob = Account([(NATIVE, 0), (ASSET55, 0)], "one-b")
    ledger = Ledger([alice, bob])
    b = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 10), TransferBuilder(bob.pubkey, 7, ASSET55)]), 2, 0)
    u = b.build_unsigned({NATIVE: 100, ASSET55: 30}, alice.keypair, "one-u", alice)
    tx = u.sign(alice.keypair, "one-s")
    lg = _single_same(engine, oracle, tx, ledger, (True, codes.OK))
    assert lg.balance_decrypts_to(alice.pubkey, NATIVE, 88) and lg.balance_decrypts_to(bob.pubkey, ASSET55, 7)
    burn = TransactionBuilder(1, alice.pubkey, ("burn", ASSET55, 30), 1, 0).build({NATIVE: 100, ASSET55: 30}, alice.keypair, "one-burn", alice)
    _single_same(engine, oracle, burn, ledger, (True, codes.OK))
    n = len(tx.record)
    pre, rp = 16 + 8, 16 + 8 + 49 + 640
    cm0 = n - 64 - 512
    rejects = [
        (_mutate(tx, n - 64, bytes(64)), codes.ERR_SIGNATURE),
        (_mutate(tx, pre + 41, (5).to_bytes(8, "big")), codes.ERR_INVALID_NONCE),
        (_mutate(tx, pre + 1, bytes([0xee]) * 32), codes.ERR_STATE),
        (_mutate(tx, cm0, bytes([3]) * 32), codes.ERR_FORMAT),
        (_mutate(tx, pre + 49 + 64, bytes([0xff]) * 32), codes.ERR_DECOMPRESSION),
        (_resigned(u, alice.keypair, [(cm0 + 64, bytes(32))], "one-id"), codes.ERR_TRANSCRIPT_IDENTITY),
        (_resigned(u, alice.keypair, [(cm0 + 256 + 64, bytes([0xff]) * 32)], "one-eq"), codes.ERR_COMMITMENT_EQ),
        (_resigned(u, alice.keypair, [(pre + 49 + 320 + 160, bytes([0xff]) * 32)], "one-val"), codes.ERR_CT_VALIDITY),
        (_resigned(u, alice.keypair, [(cm0 + 64 + 96, (12345).to_bytes(32, "little"))], "one-gen"), codes.ERR_GENERIC_PROOF),
        (_resigned(u, alice.keypair, [(rp + 128, bytes([u.record[rp + 128] ^ 1]))], "one-rp"), codes.ERR_RANGE_PROOF),
        (Transaction(tx.record[:-3]), codes.ERR_MALFORMED),
    ]
    for bad, code in rejects:
        _single_same(engine, oracle, bad, ledger, (False, code))
    # a multisig account: co-signed accepted, plain rejected
    cos = Account([(NATIVE, 0)], "one-c")
    ml = Ledger([alice, bob, cos])
    ml.set_multisig_for_account(alice.pubkey, [cos.pubkey], 1)
    _single_same(engine, oracle, tx, ml, (False, codes.ERR_FORMAT))
    u2 = b.build_unsigned({NATIVE: 100, ASSET55: 30}, alice.keypair, "one-u2", alice)
    u2.set_multisig([(0, cos.keypair.sign(u2.hash(), "one-cs"))])
    _single_same(engine, oracle, u2.sign(alice.keypair, "one-s2"), ml, (True, codes.OK))


def test_apply_batch_against_the_oracle_apply(engine, oracle):
    """xhe_batch_apply (the ledger scan alone) against the oracle's apply_without_verify run in order: a mixed batch (multisig
    set-ups, two-asset transfers, burns) and a chain batch; every ledger byte equal."""
    lib = engine.lib
    lib.xhe_batch_upload.restype = lib.xhe_batch_apply.restype = lib.xhe_batch_read_ledger.restype = ctypes.c_int32
    from xelis_he_amd.api import Verdict
    for sb in (SyntheticBatch(256, seed=b"apply-mixed", mixed=True), SyntheticBatch(300, n_transfers=3, n_senders=7, seed=b"apply-chain")):
        n_bal, n_acc = sb.view.contents.n_balances, sb.view.contents.n_accounts
        ref_cts = (ctypes.c_uint8 * (64 * n_bal))()
        ref_out = LedgerOut(ref_cts, (ctypes.c_uint64 * n_acc)(), (ctypes.c_int64 * n_acc)())
        assert oracle.xo_apply_batch(sb.records, sb.offsets, ctypes.c_uint64(sb.n_txs), sb.view, ctypes.byref(ref_out)) == 0
        ctx, h, v = ctypes.c_void_p(engine.ctx), ctypes.c_void_p(), Verdict()
        engine._check(lib.xhe_batch_upload(ctx, sb.records, sb.offsets, ctypes.c_uint64(sb.n_txs), sb.view, ctypes.byref(h), ctypes.byref(v)))
        try:
            engine._check(lib.xhe_batch_apply(ctx, h, ctypes.byref(v)))
            assert v.accept == 1
            cts = (ctypes.c_uint8 * (64 * n_bal))()
            out = LedgerOut(cts, (ctypes.c_uint64 * n_acc)(), (ctypes.c_int64 * n_acc)())
            engine._check(lib.xhe_batch_read_ledger(ctx, h, ctypes.byref(out)))
        finally:
            lib.xhe_batch_free(ctx, h)
        assert bytes(cts) == bytes(ref_cts)
        sb.close()
