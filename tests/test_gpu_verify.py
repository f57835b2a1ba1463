"""Transaction::verify_batch on the GPU (through the C ABI) against the CPU oracle on identical inputs:
verdict, error variant, index of the failing transaction and the resulting ledger bytes must be equal.
Scenarios mirror the reference's own tests (/root/reference/src/lib.rs:254-1093)."""
import ctypes

import pytest

from helpers import oracle_verify_batch
from xelis_he_amd import codes
from xelis_he_amd.api import (NATIVE, Account, Ledger, SyntheticBatch, Transaction, TransactionBuilder, TransactionUnsigned, TransferBuilder,
                              VerificationError, LedgerOut, Verdict)

pytestmark = pytest.mark.gpu
ASSET55 = bytes([55]) * 32


@pytest.fixture(scope="module")
def engine():
    import xelis_he_amd
    e = xelis_he_amd.Engine(0)
    yield e
    e.close()


def gpu_verify(engine, txs, ledger):
    """Returns (accept, code, first_bad); the ledger mirror is updated on accept."""
    try:
        Transaction.verify_batch(engine, txs, ledger, seed="gpu")
        return True, codes.OK, -1
    except VerificationError as e:
        return False, e.code, e.first_bad_tx


def check_same(engine, oracle, txs, ledger, expect_accept=None, expect_code=None):
    lg, lo = ledger.clone(), ledger.clone()
    got = gpu_verify(engine, txs, lg)
    ref = oracle_verify_batch(oracle, txs, lo)
    assert got == ref, (got, ref)
    if expect_accept is not None:
        assert got[0] == expect_accept, codes.NAMES.get(got[1])
    if expect_code is not None:
        assert got[1] == expect_code, codes.NAMES.get(got[1])
    if got[0]:
        for k, a in lo.accounts.items():
            assert lg.accounts[k].balances == a.balances
        assert lg.multisig_accounts == lo.multisig_accounts
    return lg, got


def test_single_transfer(engine, oracle):
    alice, bob = Account([(NATIVE, 100)], "a"), Account([(NATIVE, 0)], "b")
    tx = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 10)]), 1, 0).build({NATIVE: 100}, alice.keypair, "t", alice)
    lg, _ = check_same(engine, oracle, [tx], Ledger([alice, bob]), expect_accept=True)
    assert lg.balance_decrypts_to(alice.pubkey, NATIVE, 89) and lg.balance_decrypts_to(bob.pubkey, NATIVE, 10)


def test_empty_batch(engine, oracle):
    alice = Account([(NATIVE, 100)], "a")
    check_same(engine, oracle, [], Ledger([alice]), expect_accept=True)


def test_burns(engine, oracle):
    alice = Account([(NATIVE, 100)], 1)
    tx = TransactionBuilder(1, alice.pubkey, ("burn", NATIVE, 10), 1, 0).build({NATIVE: 100}, alice.keypair, "burn", alice)
    lg, _ = check_same(engine, oracle, [tx], Ledger([alice]), expect_accept=True)
    assert lg.balance_decrypts_to(alice.pubkey, NATIVE, 89)
    carol = Account([(NATIVE, 1), (ASSET55, 50)], 2)
    tx = TransactionBuilder(1, carol.pubkey, ("burn", ASSET55, 50), 1, 0).build({NATIVE: 1, ASSET55: 50}, carol.keypair, "burn2", carol)
    lg, _ = check_same(engine, oracle, [tx], Ledger([carol]), expect_accept=True)
    assert lg.balance_decrypts_to(carol.pubkey, NATIVE, 0) and lg.balance_decrypts_to(carol.pubkey, ASSET55, 0)


def test_realistic(engine, oracle):
    bob = Account([(NATIVE, 100), (ASSET55, 2)], "bob")
    alice = Account([(NATIVE, 0), (ASSET55, 0)], "alice")
    eve = Account([(NATIVE, 52), (ASSET55, 0)], "eve")
    ledger = Ledger([bob, alice, eve])
    tx1 = TransactionBuilder(1, bob.pubkey, ("transfers", [TransferBuilder(alice.pubkey, 52), TransferBuilder(eve.pubkey, 4),
                                                            TransferBuilder(eve.pubkey, 2, ASSET55)]), 1, 0).build(
        {NATIVE: 100, ASSET55: 2}, bob.keypair, "tx1", bob)
    after1 = ledger.clone()
    tx1.apply_without_verify(after1)
    tx2 = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(eve.pubkey, 30)]), 1, 0).build(
        {NATIVE: 52}, alice.keypair, "tx2", after1.get_account(alice.pubkey))
    check_same(engine, oracle, [tx2], ledger, expect_accept=False, expect_code=codes.ERR_GENERIC_PROOF)
    lg, _ = check_same(engine, oracle, [tx1, tx2], ledger, expect_accept=True)
    for acct, asset, amount in [(bob, NATIVE, 43), (bob, ASSET55, 0), (alice, NATIVE, 21), (alice, ASSET55, 0), (eve, NATIVE, 86), (eve, ASSET55, 2)]:
        assert lg.balance_decrypts_to(acct.pubkey, asset, amount)


@pytest.mark.parametrize("signers,threshold,sign_with", [([0], 1, [0]), ([0, 1], 2, [0, 1]), ([0, 1], 1, [1])])
def test_multisig(engine, oracle, signers, threshold, sign_with):
    alice, bob = Account([(NATIVE, 100)], 1), Account([(NATIVE, 0)], 2)
    cos = [Account([(NATIVE, 0)], 10 + i) for i in range(2)]
    b = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 10)]), 1, 0)
    u = b.build_unsigned({NATIVE: 100}, alice.keypair, "ms", alice)
    h = u.hash()
    u.set_multisig([(i, cos[i].keypair.sign(h, "cs%d" % i)) for i in sign_with])
    tx = u.sign(alice.keypair, "final")
    ledger = Ledger([alice, bob] + cos)
    ledger.set_multisig_for_account(alice.pubkey, [cos[i].pubkey for i in signers], threshold)
    check_same(engine, oracle, [tx], ledger, expect_accept=True)
    # without the multisig configuration the same transaction is a format error; with a plain tx and a config too
    check_same(engine, oracle, [tx], Ledger([alice, bob] + cos), expect_accept=False, expect_code=codes.ERR_FORMAT)
    plain = b.build({NATIVE: 100}, alice.keypair, "plain", alice)
    check_same(engine, oracle, [plain], ledger, expect_accept=False, expect_code=codes.ERR_FORMAT)
    u2 = b.build_unsigned({NATIVE: 100}, alice.keypair, "ms2", alice)
    u2.set_multisig([(i, cos[i].keypair.sign(b"not the hash", "bad%d" % i)) for i in sign_with])
    check_same(engine, oracle, [u2.sign(alice.keypair, "f2")], ledger, expect_accept=False, expect_code=codes.ERR_SIGNATURE)


def test_multisig_setup_then_use_in_one_batch(engine, oracle):
    alice, bob = Account([(NATIVE, 100)], 1), Account([(NATIVE, 0)], 2)
    charlie, dave = Account([(NATIVE, 0)], 3), Account([(NATIVE, 0)], 4)
    ledger = Ledger([alice, bob, charlie, dave])
    setup = TransactionBuilder(1, alice.pubkey, ("multisig", [charlie.pubkey, dave.pubkey], 2), 1, 0).build({NATIVE: 100}, alice.keypair, "setup", alice)
    after = ledger.clone()
    setup.apply_without_verify(after)
    b = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 5)]), 1, 0)
    u = b.build_unsigned({NATIVE: 99}, alice.keypair, "use", after.get_account(alice.pubkey))
    h = u.hash()
    u.set_multisig([(0, charlie.keypair.sign(h, "c")), (1, dave.keypair.sign(h, "d"))])
    use = u.sign(alice.keypair, "s")
    lg, _ = check_same(engine, oracle, [setup, use], ledger, expect_accept=True)
    assert lg.get_multisig_for_account(alice.pubkey) == ([charlie.pubkey, dave.pubkey], 2)
    assert lg.balance_decrypts_to(alice.pubkey, NATIVE, 93)
    # the transfer without co-signatures after the setup is rejected at index 1
    plain = b.build({NATIVE: 99}, alice.keypair, "plain", after.get_account(alice.pubkey))
    _, got = check_same(engine, oracle, [setup, plain], ledger, expect_accept=False, expect_code=codes.ERR_FORMAT)
    assert got[2] == 1
    # delete the configuration
    d = TransactionBuilder(1, alice.pubkey, ("multisig", [], 0), 1, 0)
    u = d.build_unsigned({NATIVE: 99}, alice.keypair, "del", after.get_account(alice.pubkey))
    h = u.hash()
    u.set_multisig([(0, charlie.keypair.sign(h, "c2")), (1, dave.keypair.sign(h, "d2"))])
    lg, _ = check_same(engine, oracle, [setup, u.sign(alice.keypair, "s2")], ledger, expect_accept=True)
    assert lg.get_multisig_for_account(alice.pubkey) is None


def _mutate(tx, offset, new_bytes):
    r = bytearray(tx.record)
    r[offset:offset + len(new_bytes)] = new_bytes
    return Transaction(bytes(r))


def test_tampering(engine, oracle):
    alice = Account([(NATIVE, 100)], 1)
    ledger = Ledger([alice])
    tx = TransactionBuilder(1, alice.pubkey, ("burn", NATIVE, 10), 1, 0).build({NATIVE: 100}, alice.keypair, "b2", alice)
    check_same(engine, oracle, [tx], ledger, expect_accept=True)
    n, pre = len(tx.record), 16
    cm_off = n - 64 - 256
    cases = [
        (_mutate(tx, pre + 49 + 32, (11).to_bytes(8, "big")), codes.ERR_SIGNATURE),     # burn amount
        (_mutate(tx, pre + 49, bytes([1]) * 32), codes.ERR_FORMAT),                      # burn asset
        (_mutate(tx, n - 64, bytes(64)), codes.ERR_SIGNATURE),                           # zero signature
        (_mutate(tx, cm_off, bytes([1]) * 32), codes.ERR_FORMAT),                        # commitment asset
        (_mutate(tx, pre + 33, (0).to_bytes(8, "big")), codes.ERR_SIGNATURE),            # fee 0
        (_mutate(tx, pre + 33, (12).to_bytes(8, "big")), codes.ERR_SIGNATURE),           # fee 12
        (_mutate(tx, pre + 41, (1).to_bytes(8, "big")), codes.ERR_INVALID_NONCE),        # nonce
        (_mutate(tx, pre + 1, bytes([0xff]) * 32), codes.ERR_STATE),                     # unknown source
        (_mutate(tx, cm_off + 32, bytes([0xff]) * 32), codes.ERR_DECOMPRESSION),         # commitment point invalid
        (_mutate(tx, cm_off + 64, bytes(32)), codes.ERR_SIGNATURE),                      # Y_0 zero: signature fails first
        (_mutate(tx, cm_off + 64 + 96 + 31, bytes([0xff])), codes.ERR_MALFORMED),        # non-canonical z_s
        (Transaction(tx.record[:-1]), codes.ERR_MALFORMED),                              # truncated
    ]
    for bad, code in cases:
        check_same(engine, oracle, [bad], ledger, expect_accept=False, expect_code=code)


def test_invalid_ledger_ciphertext_and_unknown_receiver(engine, oracle):
    alice, bob = Account([(NATIVE, 100)], 1), Account([(NATIVE, 0)], 2)
    tx = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 10)]), 1, 0).build({NATIVE: 100}, alice.keypair, "t", alice)
    bad_sender = Ledger([alice, bob]).clone()
    bad_sender.accounts[alice.pubkey].balances[NATIVE] = bytes([0xff]) * 64
    check_same(engine, oracle, [tx], bad_sender, expect_accept=False, expect_code=codes.ERR_DECOMPRESSION)
    bad_recv = Ledger([alice, bob]).clone()
    bad_recv.accounts[bob.pubkey].balances[NATIVE] = bytes([0xff]) * 64
    check_same(engine, oracle, [tx], bad_recv, expect_accept=False, expect_code=codes.ERR_DECOMPRESSION)
    check_same(engine, oracle, [tx], Ledger([alice]), expect_accept=False, expect_code=codes.ERR_STATE)


def test_chain_batch_with_failures(engine, oracle):
    alice, bob = Account([(NATIVE, 1000)], 1), Account([(NATIVE, 0)], 2)
    ledger = Ledger([alice, bob])
    prover = ledger.clone()
    txs, bal = [], 1000
    for k in range(12):
        tx = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 1)]), 3, 0).build(
            {NATIVE: bal}, alice.keypair, "c%d" % k, prover.get_account(alice.pubkey))
        tx.apply_without_verify(prover)
        bal -= 4
        txs.append(tx)
    lg, _ = check_same(engine, oracle, txs, ledger, expect_accept=True)
    assert lg.balance_decrypts_to(alice.pubkey, NATIVE, 1000 - 48) and lg.balance_decrypts_to(bob.pubkey, NATIVE, 12)
    bad = list(txs)
    r = bytearray(bad[7].record)
    r[16 + 4 + 49 + 320 + 128] ^= 1                 # t_x of the range proof: the signature covers it
    bad[7] = Transaction(bytes(r))
    _, got = check_same(engine, oracle, bad, ledger, expect_accept=False, expect_code=codes.ERR_SIGNATURE)
    assert got[2] == 7
    check_same(engine, oracle, txs[:5] + txs[6:], ledger, expect_accept=False, expect_code=codes.ERR_GENERIC_PROOF)
    # an overspending prover: sigma proofs fail
    liar = Account([(NATIVE, 5)], 9)
    lie = TransactionBuilder(1, liar.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 50)]), 1, 0).build({NATIVE: 100}, liar.keypair, "lie", liar)
    check_same(engine, oracle, [lie], Ledger([liar, bob]), expect_accept=False, expect_code=codes.ERR_GENERIC_PROOF)


def test_eight_transfers(engine, oracle):
    alice = Account([(NATIVE, 1000), (ASSET55, 77)], 1)
    dests = [Account([(NATIVE, 0), (ASSET55, 0)], 20 + i) for i in range(3)]
    ts = [TransferBuilder(dests[i % 3].pubkey, 1 + i, ASSET55 if i % 4 == 3 else NATIVE) for i in range(8)]
    tx = TransactionBuilder(1, alice.pubkey, ("transfers", ts), 3, 0).build({NATIVE: 1000, ASSET55: 77}, alice.keypair, "eight", alice)
    lg, _ = check_same(engine, oracle, [tx], Ledger([alice] + dests), expect_accept=True)
    assert lg.balance_decrypts_to(alice.pubkey, ASSET55, 77 - 4 - 8)


def test_seventeen_transfers_then_small(engine, oracle):
    """18 value commitments -> 2048 generators per side: beyond the fixed-base tables the context keeps, so the
    static generators go through the bucket pipeline; a small transaction afterwards is back on the tables."""
    alice = Account([(NATIVE, 5000)], "big-a")
    dests = [Account([(NATIVE, 0)], "big-d%d" % i) for i in range(4)]
    ts = [TransferBuilder(dests[i % 4].pubkey, 1 + i) for i in range(17)]
    tx = TransactionBuilder(1, alice.pubkey, ("transfers", ts), 2, 0).build({NATIVE: 5000}, alice.keypair, "seventeen", alice)
    ledger = Ledger([alice] + dests)
    lg, _ = check_same(engine, oracle, [tx], ledger, expect_accept=True)
    assert lg.balance_decrypts_to(alice.pubkey, NATIVE, 5000 - 2 - sum(range(1, 18)))
    bad = bytearray(tx.record)
    bad[-100] ^= 1
    check_same(engine, oracle, [Transaction(bytes(bad))], ledger, expect_accept=False)
    bob, carol = Account([(NATIVE, 100)], "big-b"), Account([(NATIVE, 0)], "big-c")
    small = TransactionBuilder(1, bob.pubkey, ("transfers", [TransferBuilder(carol.pubkey, 10)]), 1, 0).build({NATIVE: 100}, bob.keypair, "small", bob)
    check_same(engine, oracle, [small], Ledger([bob, carol]), expect_accept=True)


def _resigned(unsigned, keypair, edits, seed):
    """A lying prover: the unsigned record with some bytes replaced, signed afterwards (so the signature is valid)."""
    r = bytearray(unsigned.record)
    for off, data in edits:
        r[off:off + len(data)] = data
    return TransactionUnsigned(bytes(r)).sign(keypair, seed)


@pytest.mark.parametrize("n_transfers", [1, 8, 17])
def test_range_proof_rejections(engine, oracle, n_transfers):
    """Valid signature and sigma proofs, wrong range proof: the error is RangeProof (verify.rs:504-514), whatever part
    of the proof is wrong.  The range proof is the last thing the transcript absorbs, so the sigma challenges are intact."""
    alice = Account([(NATIVE, 5000)], "rp-a")
    dests = [Account([(NATIVE, 0)], "rp-d%d" % i) for i in range(3)]
    ts = [TransferBuilder(dests[i % 3].pubkey, 1 + i) for i in range(n_transfers)]
    ledger = Ledger([alice] + dests)
    u = TransactionBuilder(1, alice.pubkey, ("transfers", ts), 2, 0).build_unsigned({NATIVE: 5000}, alice.keypair, "rp%d" % n_transfers, alice)
    check_same(engine, oracle, [u.sign(alice.keypair, "ok")], ledger, expect_accept=True)
    m = 1
    while m < n_transfers + 1:
        m *= 2
    lg = 6 + m.bit_length() - 1
    rp = 16 + 4 * n_transfers + 49 + 320 * n_transfers
    rec = u.record
    flip = lambda off: (off, bytes([rec[off] ^ 1]))
    cases = {
        "t_x": [flip(rp + 128)], "t_x_blinding": [flip(rp + 160)], "e_blinding": [flip(rp + 192)],
        "a": [flip(rp + 224 + 64 * lg)], "b": [flip(rp + 224 + 64 * lg + 32)],
        "S := A": [(rp + 32, rec[rp:rp + 32])], "T_1 <-> T_2": [(rp + 64, rec[rp + 96:rp + 128]), (rp + 96, rec[rp + 64:rp + 96])],
        "L_0 <-> R_0": [(rp + 224, rec[rp + 256:rp + 288]), (rp + 256, rec[rp + 224:rp + 256])],
        "last R := last L": [(rp + 224 + 64 * (lg - 1) + 32, rec[rp + 224 + 64 * (lg - 1):rp + 224 + 64 * (lg - 1) + 32])],
    }
    for name, edits in cases.items():
        bad = _resigned(u, alice.keypair, edits, "bad-" + name)
        _, got = check_same(engine, oracle, [bad], ledger, expect_accept=False, expect_code=codes.ERR_RANGE_PROOF)
        assert got[2] == -1, name
    # a wrong proof among right ones is found too
    good = u.sign(alice.keypair, "ok2")
    bob = Account([(NATIVE, 100)], "rp-b")
    other = TransactionBuilder(1, bob.pubkey, ("burn", NATIVE, 10), 1, 0).build({NATIVE: 100}, bob.keypair, "rp-burn", bob)
    check_same(engine, oracle, [other, _resigned(u, alice.keypair, cases["a"], "bad-mixed")], Ledger([alice, bob] + dests), expect_accept=False,
               expect_code=codes.ERR_RANGE_PROOF)
    check_same(engine, oracle, [other, good], Ledger([alice, bob] + dests), expect_accept=True)


def test_synthetic_batch(engine, oracle):
    sb = SyntheticBatch(96, n_transfers=1, n_senders=8, seed=b"synth-96")
    lib = engine.lib
    n_bal = sb.view.contents.n_balances
    out_cts = (ctypes.c_uint8 * (64 * n_bal))()
    out_nonces = (ctypes.c_uint64 * sb.view.contents.n_accounts)()
    ms = (ctypes.c_int64 * sb.view.contents.n_accounts)()
    out = LedgerOut(out_cts, out_nonces, ms)
    v = Verdict()
    engine._check(lib.xhe_verify_batch(engine.ctx, sb.records, sb.offsets, sb.n_txs, sb.view, ctypes.byref(out), ctypes.byref(v), b"\x01" * 32))
    assert v.accept == 1, codes.NAMES.get(v.err_code)
    ref_cts = (ctypes.c_uint8 * (64 * n_bal))()
    ref_out = LedgerOut(ref_cts, out_nonces, ms)
    rv = Verdict()
    oracle.xo_verify_batch(sb.records, sb.offsets, ctypes.c_uint64(sb.n_txs), sb.view, ctypes.byref(ref_out), ctypes.byref(rv), b"\x02" * 32)
    assert rv.accept == 1
    assert bytes(out_cts) == bytes(ref_cts)


def _raw_verify(lib, fn_owner, ctx, sb, records=None):
    n_bal, n_acc = sb.view.contents.n_balances, sb.view.contents.n_accounts
    cts = (ctypes.c_uint8 * (64 * n_bal))()
    ms = (ctypes.c_int64 * n_acc)()
    out = LedgerOut(cts, (ctypes.c_uint64 * n_acc)(), ms)
    v = Verdict()
    recs = sb.records if records is None else records
    if ctx is None:
        fn_owner.xo_verify_batch(recs, sb.offsets, ctypes.c_uint64(sb.n_txs), sb.view, ctypes.byref(out), ctypes.byref(v), b"\x02" * 32)
    else:
        fn_owner._check(lib.xhe_verify_batch(ctx, recs, sb.offsets, sb.n_txs, sb.view, ctypes.byref(out), ctypes.byref(v), b"\x01" * 32))
    return (v.accept, v.err_code, v.first_bad_tx), bytes(cts), list(ms)


def test_mixed_synthetic_batch(engine, oracle):
    """The mixed workload (multisig set-ups and their use in one batch, two-asset transactions, burns): verdict, ledger
    bytes and multisig bookkeeping equal the oracle's; a broken co-signature and a broken burn are found at their index."""
    sb = SyntheticBatch(256, seed=b"mixed-256", mixed=True)
    got, cts, ms = _raw_verify(engine.lib, engine, engine.ctx, sb)
    ref, ref_cts, ref_ms = _raw_verify(None, oracle, None, sb)
    assert got == ref == (1, codes.OK, -1)
    assert cts == ref_cts and ms == ref_ms
    offs = list(sb.offsets)
    for tx, back in [(16 * 3 + 14, 64 + 65 + 10), (16 * 9 + 11, 64 + 256 + 300), (16 * 15 + 0, 64 + 10)]:
        bad = bytearray(sb.records)
        bad[offs[tx + 1] - back] ^= 1              # a co-signature / the range proof of a burn / a commitment proof scalar of a set-up
        g, _, _ = _raw_verify(engine.lib, engine, engine.ctx, sb, bytes(bad))
        r, _, _ = _raw_verify(None, oracle, None, sb, bytes(bad))
        assert g == r and g[0] == 0 and g[2] == tx, (tx, g, r)


def test_random_corruptions_of_a_mixed_batch(engine, oracle):
    """Differential run over seeded random corruptions of the mixed batch (one to three bytes changed anywhere in one to two
    records: headers, keys, amounts, nonces, proof points and scalars, multisig entries, signatures): the verdict, the error
    variant and the index of the failing transaction equal the oracle's for every one of them."""
    import random
    sb = SyntheticBatch(48, seed=b"mixed-fuzz", mixed=True)
    offs = list(sb.offsets)
    r = random.Random(20261114)
    seen = {}
    for case in range(600):
        bad = bytearray(sb.records)
        for _ in range(r.choice((1, 1, 1, 2))):
            tx = r.randrange(sb.n_txs)
            lo, hi = offs[tx], offs[tx + 1]
            for _ in range(r.choice((1, 1, 2, 3))):
                pos = r.randrange(lo, hi) if r.random() < 0.8 else r.randrange(lo, min(hi, lo + 80))     # some near the header and the source key
                bad[pos] ^= 1 << r.randrange(8)
        g, g_cts, g_ms = _raw_verify(engine.lib, engine, engine.ctx, sb, bytes(bad))
        ref, ref_cts, ref_ms = _raw_verify(None, oracle, None, sb, bytes(bad))
        assert g == ref, (case, g, ref)
        if g[0] == 1:
            assert g_cts == ref_cts and g_ms == ref_ms, case
        seen[g[1]] = seen.get(g[1], 0) + 1
    print("verdict codes reached:", dict(sorted(seen.items())))
    assert len(seen) >= 5, seen              # the corruptions reach several error variants, not one


def test_apply_without_verify_batch(engine, oracle):
    """xhe_batch_apply (the ledger scan alone) against the host's apply_without_verify run transaction by transaction,
    and against the ledger an accepting verify_batch leaves; unknown receivers and broken ciphertexts are reported."""
    bob = Account([(NATIVE, 100), (ASSET55, 2)], "ap-bob")
    alice = Account([(NATIVE, 0), (ASSET55, 0)], "ap-alice")
    eve = Account([(NATIVE, 52), (ASSET55, 0)], "ap-eve")
    ledger = Ledger([bob, alice, eve])
    tx1 = TransactionBuilder(1, bob.pubkey, ("transfers", [TransferBuilder(alice.pubkey, 52), TransferBuilder(eve.pubkey, 4),
                                                            TransferBuilder(eve.pubkey, 2, ASSET55)]), 1, 0).build(
        {NATIVE: 100, ASSET55: 2}, bob.keypair, "ap1", bob)
    after1 = ledger.clone()
    tx1.apply_without_verify(after1)
    tx2 = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(eve.pubkey, 30)]), 1, 0).build(
        {NATIVE: 52}, alice.keypair, "ap2", after1.get_account(alice.pubkey))
    tx3 = TransactionBuilder(1, eve.pubkey, ("burn", NATIVE, 7), 2, 0).build({NATIVE: 52}, eve.keypair, "ap3", eve)
    txs = [tx1, tx2, tx3]
    host = ledger.clone()
    for t in txs:
        t.apply_without_verify(host)
    gpu = ledger.clone()
    Transaction.apply_batch(engine, txs, gpu)
    verified = ledger.clone()
    Transaction.verify_batch(engine, [tx1, tx2], verified, seed="ap")
    for k, a in host.accounts.items():
        assert gpu.accounts[k].balances == a.balances
    assert gpu.balance_decrypts_to(eve.pubkey, NATIVE, 52 + 4 + 30 - 9) and gpu.balance_decrypts_to(bob.pubkey, NATIVE, 43)
    assert verified.accounts[alice.pubkey].balances == gpu.accounts[alice.pubkey].balances
    # apply does not check proofs or signatures: a transaction with a zeroed signature still applies
    forged = _mutate(tx3, len(tx3.record) - 64, bytes(64))
    again = ledger.clone()
    Transaction.apply_batch(engine, [tx1, tx2, forged], again)
    assert again.accounts[eve.pubkey].balances == host.accounts[eve.pubkey].balances
    # unknown receiver -> State at its index; broken transfer ciphertext -> Decompression
    with pytest.raises(VerificationError) as e:
        Transaction.apply_batch(engine, txs, Ledger([bob, alice]))
    assert e.value.code == codes.ERR_STATE and e.value.first_bad_tx == 0
    bad = _mutate(tx2, 16 + 4 + 49 + 64, bytes([0xff]) * 32)
    with pytest.raises(VerificationError) as e:
        Transaction.apply_batch(engine, [tx1, bad], ledger.clone())
    assert e.value.code == codes.ERR_DECOMPRESSION and e.value.first_bad_tx == 1
    # a larger chained batch: equal to the ledger verify_batch produces
    sb = SyntheticBatch(256, n_transfers=2, n_senders=16, seed=b"apply-256")
    lib = engine.lib
    lib.xhe_batch_upload.restype = lib.xhe_batch_apply.restype = lib.xhe_batch_read_ledger.restype = ctypes.c_int32
    n_bal, n_acc = sb.view.contents.n_balances, sb.view.contents.n_accounts
    res = []
    for mode in ("apply", "verify"):
        h, v = ctypes.c_void_p(), Verdict()
        ctx = ctypes.c_void_p(engine.ctx)
        engine._check(lib.xhe_batch_upload(ctx, sb.records, sb.offsets, ctypes.c_uint64(sb.n_txs), sb.view, ctypes.byref(h), ctypes.byref(v)))
        if mode == "apply":
            engine._check(lib.xhe_batch_apply(ctx, h, ctypes.byref(v)))
        else:
            engine._check(lib.xhe_batch_verify(ctx, h, ctypes.byref(v), b"\x05" * 32))
        assert v.accept == 1
        cts = (ctypes.c_uint8 * (64 * n_bal))()
        out = LedgerOut(cts, (ctypes.c_uint64 * n_acc)(), (ctypes.c_int64 * n_acc)())
        engine._check(lib.xhe_batch_read_ledger(ctx, h, ctypes.byref(out)))
        lib.xhe_batch_free(ctx, h)
        res.append(bytes(cts))
    assert res[0] == res[1]


def test_transcript_stream_in_several_passes(engine, oracle, switch):
    """The transcripts' byte stream is capped per pass (1 GiB); with the cap lowered to two tiles the 300 transactions
    of this batch take three passes over the same buffer and must give the same challenges (verdict and ledger)."""
    sb = SyntheticBatch(300, n_transfers=1, n_senders=20, seed=b"passes-300")
    ref, ref_cts, _ = _raw_verify(None, oracle, None, sb)
    switch("XHE_TRANSCRIPT_CAP_WORDS", str(2 * 21 * 64 * 30))
    got, cts, _ = _raw_verify(engine.lib, engine, engine.ctx, sb)
    assert got == ref == (1, codes.OK, -1) and cts == ref_cts
    bad = bytearray(sb.records)
    bad[list(sb.offsets)[280] + 16 + 4 + 49 + 100] ^= 1           # inside a transfer ciphertext of transaction 280 (last pass)
    g, _, _ = _raw_verify(engine.lib, engine, engine.ctx, sb, bytes(bad))
    r, _, _ = _raw_verify(None, oracle, None, sb, bytes(bad))
    assert g == r and g[0] == 0 and g[2] == 280


def test_large_batch_schedule_on_a_small_batch(engine, oracle, switch):
    """Batches of 2048 transactions or more take another schedule (public keys decoded and signatures verified on a stream
    of their own from the start).  XHE_SIG_STREAM_MIN_TXS = 1 runs it on the 256-transaction mixed batch (1.25 signatures
    per transaction, co-signatures included): same verdicts, ledger and failing indices."""
    sb = SyntheticBatch(256, seed=b"mixed-256", mixed=True)
    ref, ref_cts, ref_ms = _raw_verify(None, oracle, None, sb)
    switch("XHE_SIG_STREAM_MIN_TXS", "1")
    got, cts, ms = _raw_verify(engine.lib, engine, engine.ctx, sb)
    assert got == ref == (1, codes.OK, -1)
    assert cts == ref_cts and ms == ref_ms
    offs = list(sb.offsets)
    # the main signature's s and e of a transfer, a co-signature, the signature of the last transaction
    for tx, back in [(16 * 2 + 3, 10), (16 * 2 + 3, 40), (16 * 3 + 14, 64 + 65 + 10), (255, 1)]:
        bad = bytearray(sb.records)
        bad[offs[tx + 1] - back] ^= 1
        g, _, _ = _raw_verify(engine.lib, engine, engine.ctx, sb, bytes(bad))
        r, _, _ = _raw_verify(None, oracle, None, sb, bytes(bad))
        assert g == r and g[0] == 0 and g[2] == tx, (tx, back, g, r)
    sb2 = SyntheticBatch(700, n_transfers=1, n_senders=50, seed=b"lane-700")
    assert _raw_verify(engine.lib, engine, engine.ctx, sb2) == _raw_verify(None, oracle, None, sb2)


def test_transcripts_on_one_lane_per_transaction(engine, oracle, switch):
    """Passes of more than 2048 wavefronts hash the transcripts with one lane per transaction instead of a pair of lanes
    (bit-interleaved state).  XHE_TRANSCRIPT_PAIR_MAX_WAVES = 0 runs that kernel on small batches (five transcript shapes of
    the mixed batch, eight-transfer transactions): same verdicts, ledger and failing index as the oracle."""
    switch("XHE_TRANSCRIPT_PAIR_MAX_WAVES", "0")
    sb = SyntheticBatch(256, seed=b"mixed-256", mixed=True)
    ref, ref_cts, ref_ms = _raw_verify(None, oracle, None, sb)
    got, cts, ms = _raw_verify(engine.lib, engine, engine.ctx, sb)
    assert got == ref == (1, codes.OK, -1)
    assert cts == ref_cts and ms == ref_ms
    sb8 = SyntheticBatch(40, n_transfers=8, n_senders=8, seed=b"one-lane-t8")
    assert _raw_verify(engine.lib, engine, engine.ctx, sb8) == _raw_verify(None, oracle, None, sb8)
    bad = bytearray(sb8.records)
    bad[list(sb8.offsets)[33] + 16 + 4 + 49 + 100] ^= 1
    g, _, _ = _raw_verify(engine.lib, engine, engine.ctx, sb8, bytes(bad))
    r, _, _ = _raw_verify(None, oracle, None, sb8, bytes(bad))
    assert g == r and g[0] == 0 and g[2] == 33


@pytest.mark.parametrize("n_senders,forced", [(1, None), (3, None), (150, None), (150, "1"), (1, "0")])
def test_signatures_through_key_tables(engine, oracle, switch, n_senders, forced):
    """Batches whose keys sign six or more transactions each add from a table of multiples built once per distinct signing
    key; the others build the table per signature (XHE_SIG_KEY_TABLES forces either).  One key signing the whole batch (the
    reference bench's single chain), a few keys, and as many keys as transactions give the oracle's verdicts on both paths,
    also with a flipped bit in s or e of the first, a middle and the last signature (the error names that transaction)."""
    if forced is not None:
        switch("XHE_SIG_KEY_TABLES", forced)
    sb = SyntheticBatch(150, n_transfers=1, n_senders=n_senders, seed=b"key-tables-%d" % n_senders)
    ref, ref_cts, _ = _raw_verify(None, oracle, None, sb)
    got, cts, _ = _raw_verify(engine.lib, engine, engine.ctx, sb)
    assert got == ref == (1, codes.OK, -1) and cts == ref_cts
    offs = list(sb.offsets)
    for tx, back in [(0, 5), (77, 40), (149, 64)]:
        bad = bytearray(sb.records)
        bad[offs[tx + 1] - back] ^= 0x10
        g, _, _ = _raw_verify(engine.lib, engine, engine.ctx, sb, bytes(bad))
        r, _, _ = _raw_verify(None, oracle, None, sb, bytes(bad))
        assert g == r and g[0] == 0 and g[2] == tx, (n_senders, forced, tx, back, g, r)
