"""Transactions built by the product's CPU builder, verified by the CPU oracle: the scenarios of the
reference's own tests (/root/reference/src/lib.rs:254-1093) -- every accept/reject case, the decrypted
end balances, batch-versus-single equivalence.  Two independent code bases (prover in
xelis-he_amd/csrc, verifier in oracle/) must agree on transcript, proofs and ledger arithmetic."""
import pytest

from helpers import oracle_verify_batch
from xelis_he_amd import codes
from xelis_he_amd.api import (NATIVE, Account, Keypair, Ledger, Transaction, TransactionBuilder, TransactionUnsigned,
                              TransferBuilder)

ASSET55 = bytes([55]) * 32


def _transfer_tx(alice, bob, amount=10, fee=1, nonce=0, balance=100, seed="t"):
    b = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, amount)]), fee, nonce)
    assert b.get_transaction_cost(NATIVE) == amount + fee and len(b.used_assets()) == 1
    return b, b.build({NATIVE: balance}, alice.keypair, seed, alice)


def test_simple_transfer_single_and_batch(oracle):
    alice, bob = Account([(NATIVE, 100)], "a"), Account([(NATIVE, 0)], "b")
    _, tx = _transfer_tx(alice, bob)
    for single in (True, False):
        ledger = Ledger([alice, bob]).clone()
        ok, code, _ = oracle_verify_batch(oracle, [tx], ledger, single=single)
        assert ok, codes.NAMES[code]
        assert ledger.balance_decrypts_to(alice.pubkey, NATIVE, 89)
        assert ledger.balance_decrypts_to(bob.pubkey, NATIVE, 10)


def test_invalid_multisig(oracle):                      # lib.rs:254-307
    alice, bob, charlie = Account([(NATIVE, 100)], 1), Account([(NATIVE, 0)], 2), Account([(NATIVE, 0)], 3)
    _, tx = _transfer_tx(alice, bob)
    ledger = Ledger([alice, bob])
    assert oracle_verify_batch(oracle, [tx], ledger.clone(), single=True)[0]
    ledger.set_multisig_for_account(alice.pubkey, [charlie.pubkey], 1)
    ok, code, _ = oracle_verify_batch(oracle, [tx], ledger, single=True)
    assert not ok and code == codes.ERR_FORMAT


@pytest.mark.parametrize("signers,threshold,sign_with", [
    ([0], 1, [0]),              # lib.rs:309-366
    ([0, 1], 2, [0, 1]),        # lib.rs:368-428
    ([0, 1], 1, [1]),           # lib.rs:430-490
])
def test_multisig(oracle, signers, threshold, sign_with):
    alice, bob = Account([(NATIVE, 100)], 1), Account([(NATIVE, 0)], 2)
    cos = [Account([(NATIVE, 0)], 10 + i) for i in range(2)]
    b = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 10)]), 1, 0)
    u = b.build_unsigned({NATIVE: 100}, alice.keypair, "ms", alice)
    h = u.hash()
    u.set_multisig([(i, cos[i].keypair.sign(h, "cs%d" % i)) for i in sign_with])
    tx = u.sign(alice.keypair, "final")
    ledger = Ledger([alice, bob] + cos)
    ledger.set_multisig_for_account(alice.pubkey, [cos[i].pubkey for i in signers], threshold)
    ok, code, _ = oracle_verify_batch(oracle, [tx], ledger, single=True)
    assert ok, codes.NAMES[code]
    # a wrong co-signature is rejected
    u2 = b.build_unsigned({NATIVE: 100}, alice.keypair, "ms2", alice)
    u2.set_multisig([(i, cos[i].keypair.sign(b"not the hash", "bad%d" % i)) for i in sign_with])
    ledger2 = Ledger([alice, bob] + cos)
    ledger2.set_multisig_for_account(alice.pubkey, [cos[i].pubkey for i in signers], threshold)
    ok, code, _ = oracle_verify_batch(oracle, [u2.sign(alice.keypair, "f2")], ledger2, single=True)
    assert not ok and code == codes.ERR_SIGNATURE


def test_multisig_setup_and_delete(oracle):              # lib.rs:492-612
    alice = Account([(NATIVE, 100)], 1)
    charlie, dave = Account([(NATIVE, 0)], 3), Account([(NATIVE, 0)], 4)
    b = TransactionBuilder(1, alice.pubkey, ("multisig", [charlie.pubkey, dave.pubkey], 2), 1, 0)
    assert b.get_transaction_cost(NATIVE) == 1
    tx = b.build({NATIVE: 100}, alice.keypair, "setup", alice)
    ledger = Ledger([alice, charlie, dave])
    ok, code, _ = oracle_verify_batch(oracle, [tx], ledger, single=True)
    assert ok, codes.NAMES[code]
    assert ledger.get_multisig_for_account(alice.pubkey) == ([charlie.pubkey, dave.pubkey], 2)
    # delete: signed by the existing signers
    alice2 = ledger.get_account(alice.pubkey)
    d = TransactionBuilder(1, alice.pubkey, ("multisig", [], 0), 1, 0)
    u = d.build_unsigned({NATIVE: 99}, alice.keypair, "del", alice2)
    h = u.hash()
    u.set_multisig([(0, charlie.keypair.sign(h, "c")), (1, dave.keypair.sign(h, "d"))])
    ok, code, _ = oracle_verify_batch(oracle, [u.sign(alice.keypair, "s")], ledger, single=True)
    assert ok, codes.NAMES[code]
    assert ledger.get_multisig_for_account(alice.pubkey) is None


def test_burn_batch(oracle):                             # lib.rs:614-703
    alice = Account([(NATIVE, 100)], 1)
    b = TransactionBuilder(1, alice.pubkey, ("burn", NATIVE, 10), 1, 0)
    assert b.get_transaction_cost(NATIVE) == 11
    ledger = Ledger([alice])
    ok, code, _ = oracle_verify_batch(oracle, [b.build({NATIVE: 100}, alice.keypair, "burn", alice)], ledger)
    assert ok, codes.NAMES[code]
    assert ledger.balance_decrypts_to(alice.pubkey, NATIVE, 89)
    carol = Account([(NATIVE, 1), (ASSET55, 50)], 2)
    b = TransactionBuilder(1, carol.pubkey, ("burn", ASSET55, 50), 1, 0)
    assert b.get_transaction_cost(NATIVE) == 1 and b.get_transaction_cost(ASSET55) == 50
    ledger = Ledger([carol])
    ok, code, _ = oracle_verify_batch(oracle, [b.build({NATIVE: 1, ASSET55: 50}, carol.keypair, "burn2", carol)], ledger)
    assert ok, codes.NAMES[code]
    assert ledger.balance_decrypts_to(carol.pubkey, NATIVE, 0) and ledger.balance_decrypts_to(carol.pubkey, ASSET55, 0)


def _mutate(tx, offset, new_bytes):
    r = bytearray(tx.record)
    r[offset:offset + len(new_bytes)] = new_bytes
    return Transaction(bytes(r))


def test_invalid_burn_and_tampering(oracle):             # lib.rs:705-829
    alice = Account([(NATIVE, 100)], 1)
    ledger = Ledger([alice])
    tx = TransactionBuilder(1, alice.pubkey, ("burn", NATIVE, 100), 0, 0).build({NATIVE: 100}, alice.keypair, "b", alice)
    assert oracle_verify_batch(oracle, [tx], ledger.clone(), single=True)[0]
    pre = 16
    # burn amount 101 (big-endian u64 after the 32-byte asset)
    ok, code, _ = oracle_verify_batch(oracle, [_mutate(tx, pre + 49 + 32, (101).to_bytes(8, "big"))], ledger.clone(), single=True)
    assert not ok and code == codes.ERR_SIGNATURE
    # burn asset [1;32]
    ok, code, _ = oracle_verify_batch(oracle, [_mutate(tx, pre + 49, bytes([1]) * 32)], ledger.clone(), single=True)
    assert not ok and code == codes.ERR_FORMAT

    tx = TransactionBuilder(1, alice.pubkey, ("burn", NATIVE, 10), 1, 0).build({NATIVE: 100}, alice.keypair, "b2", alice)
    assert oracle_verify_batch(oracle, [tx], ledger.clone(), single=True)[0]
    n = len(tx.record)
    # zero signature
    ok, code, _ = oracle_verify_batch(oracle, [_mutate(tx, n - 64, bytes(64))], ledger.clone(), single=True)
    assert not ok and code == codes.ERR_SIGNATURE
    # commitment asset changed
    cm_off = n - 64 - 256
    ok, code, _ = oracle_verify_batch(oracle, [_mutate(tx, cm_off, bytes([1]) * 32)], ledger.clone(), single=True)
    assert not ok and code == codes.ERR_FORMAT
    # commitments cleared: rebuild the record without the commitment
    r = bytearray(tx.record)
    del r[cm_off:cm_off + 256]
    r[4:6] = (0).to_bytes(2, "little")
    ok, code, _ = oracle_verify_batch(oracle, [Transaction(bytes(r))], ledger.clone(), single=True)
    assert not ok and code == codes.ERR_FORMAT
    # fee 0 / 12, nonce 1
    for fee in (0, 12):
        ok, code, _ = oracle_verify_batch(oracle, [_mutate(tx, pre + 33, fee.to_bytes(8, "big"))], ledger.clone(), single=True)
        assert not ok and code == codes.ERR_SIGNATURE
    ok, code, _ = oracle_verify_batch(oracle, [_mutate(tx, pre + 41, (1).to_bytes(8, "big"))], ledger.clone(), single=True)
    assert not ok and code == codes.ERR_INVALID_NONCE
    # a proof scalar flipped: signature covers it
    ok, code, _ = oracle_verify_batch(oracle, [_mutate(tx, cm_off + 64 + 96, bytes([tx.record[cm_off + 64 + 96] ^ 1]))], ledger.clone(), single=True)
    assert not ok and code == codes.ERR_SIGNATURE


def test_realistic(oracle):                              # lib.rs:831-949
    bob = Account([(NATIVE, 100), (ASSET55, 2)], "bob")
    alice = Account([(NATIVE, 0), (ASSET55, 0)], "alice")
    eve = Account([(NATIVE, 52), (ASSET55, 0)], "eve")
    ledger = Ledger([bob, alice, eve])
    b1 = TransactionBuilder(1, bob.pubkey, ("transfers", [TransferBuilder(alice.pubkey, 52), TransferBuilder(eve.pubkey, 4),
                                                           TransferBuilder(eve.pubkey, 2, ASSET55)]), 1, 0)
    assert b1.get_transaction_cost(NATIVE) == 57 and b1.get_transaction_cost(ASSET55) == 2
    tx1 = b1.build({NATIVE: 100, ASSET55: 2}, bob.keypair, "tx1", bob)
    after1 = ledger.clone()
    tx1.apply_without_verify(after1)
    b2 = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(eve.pubkey, 30)]), 1, 0)
    assert b2.get_transaction_cost(NATIVE) == 31 and b2.get_transaction_cost(ASSET55) == 0
    tx2 = b2.build({NATIVE: 52}, alice.keypair, "tx2", after1.get_account(alice.pubkey))
    # tx2 alone against the original ledger fails (alice's balance is still 0)
    ok, code, _ = oracle_verify_batch(oracle, [tx2], ledger.clone())
    assert not ok and code == codes.ERR_GENERIC_PROOF
    ok, code, _ = oracle_verify_batch(oracle, [tx1, tx2], ledger)
    assert ok, codes.NAMES[code]
    for acct, asset, amount in [(bob, NATIVE, 43), (bob, ASSET55, 0), (alice, NATIVE, 21), (alice, ASSET55, 0), (eve, NATIVE, 86), (eve, ASSET55, 2)]:
        assert ledger.balance_decrypts_to(acct.pubkey, asset, amount)
    # the host apply_without_verify agrees with the verifier's ledger
    tx2.apply_without_verify(after1)
    for k, a in ledger.accounts.items():
        assert a.balances == after1.accounts[k].balances


def test_non_native_asset_and_extra_data(oracle):        # lib.rs:951-1093
    bob = Account([(NATIVE, 100), (ASSET55, 2)], 1)
    alice = Account([(NATIVE, 0), (ASSET55, 0)], 2)
    ledger = Ledger([bob, alice])
    b = TransactionBuilder(1, bob.pubkey, ("transfers", [TransferBuilder(alice.pubkey, 2, ASSET55, b"hi, this is the president talking")]), 10, 0)
    assert b.get_transaction_cost(NATIVE) == 10 and b.get_transaction_cost(ASSET55) == 2
    tx = b.build({NATIVE: 100, ASSET55: 2}, bob.keypair, "nn", bob)
    ok, code, _ = oracle_verify_batch(oracle, [tx], ledger, single=True)
    assert ok, codes.NAMES[code]
    for acct, asset, amount in [(bob, NATIVE, 90), (bob, ASSET55, 0), (alice, NATIVE, 0), (alice, ASSET55, 2)]:
        assert ledger.balance_decrypts_to(acct.pubkey, asset, amount)


def test_range_proof_catches_overspend(oracle):
    """A prover that lies about its balance (ciphertext of 5, claims 100) produces sigma proofs that do
    not verify; the batch check reports GenericProof."""
    alice, bob = Account([(NATIVE, 5)], 1), Account([(NATIVE, 0)], 2)
    tx = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 50)]), 1, 0).build({NATIVE: 100}, alice.keypair, "lie", alice)
    ok, code, _ = oracle_verify_batch(oracle, [tx], Ledger([alice, bob]))
    assert not ok and code == codes.ERR_GENERIC_PROOF


def test_batch_with_bad_tx_inside(oracle):
    alice, bob = Account([(NATIVE, 1000)], 1), Account([(NATIVE, 0)], 2)
    ledger = Ledger([alice, bob])
    prover = ledger.clone()
    txs, bal = [], 1000
    for k in range(6):
        tx = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 1)]), 3, 0).build(
            {NATIVE: bal}, alice.keypair, "c%d" % k, prover.get_account(alice.pubkey))
        tx.apply_without_verify(prover)
        bal -= 4
        txs.append(tx)
    ok, code, _ = oracle_verify_batch(oracle, txs, ledger.clone())
    assert ok, codes.NAMES[code]
    # corrupt a range-proof scalar of tx 3 and re-sign is impossible -> signature error at index 3
    bad = list(txs)
    r = bytearray(bad[3].record)
    r[16 + 4 + 49 + 320 + 128] ^= 1
    bad[3] = Transaction(bytes(r))
    ok, code, idx = oracle_verify_batch(oracle, bad, ledger.clone())
    assert not ok and code == codes.ERR_SIGNATURE and idx == 3
    # dropping tx 2 breaks the ledger chain for tx 3..5: sigma batch check fails
    ok, code, idx = oracle_verify_batch(oracle, txs[:2] + txs[3:], ledger.clone())
    assert not ok and code == codes.ERR_GENERIC_PROOF and idx == -1
