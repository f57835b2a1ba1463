"""GPU-vs-oracle parity on the reference's error variants and edge cases that need a lying prover or an unusual record:
Transcript(IdentityPoint), CommitmentEqProof and CiphertextValidityProof (/root/reference/src/proofs.rs:151-179,293-317),
a multisig signer index beyond the configuration (/root/reference/src/tx/verify.rs:276), transfers with extra data
(:638-640), CallContract / DeployContract records, a range proof whose length disagrees with the number of value
commitments, the reference bench's 255-transfer transaction (/root/reference/benches/tx.rs:108-110), and the challenges and
generators the kernels work with against the committed fixtures.  Integer work: the bar is bit-exact."""
import ctypes

import pytest

from golden_io import FlatCase, primitives, tx_cases
from helpers import buf
from test_gpu_verify import ASSET55, _resigned, check_same
from xelis_he_amd import codes
from xelis_he_amd._capi import XheError
from xelis_he_amd.api import NATIVE, Account, Ledger, Transaction, TransactionBuilder, TransactionUnsigned, TransferBuilder, Verdict

pytestmark = pytest.mark.gpu

BAD_POINT = bytes([0xff]) * 32          # not a canonical field element: never decodes
PRE1 = 16 + 4                            # header of a one-transfer record; the pre-image follows
XFER0 = PRE1 + 49                        # first transfer: asset, dest, C, D_s, D_r, proof(Y_0, Y_1, Y_2, z_r, z_x)


@pytest.fixture(scope="module")
def engine():
    import xelis_he_amd
    e = xelis_he_amd.Engine(0)
    yield e
    e.close()


def _one_transfer():
    alice, bob = Account([(NATIVE, 100)], "gap-a"), Account([(NATIVE, 0)], "gap-b")
    u = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 10)]), 1, 0).build_unsigned({NATIVE: 100}, alice.keypair, "gap-u", alice)
    cm = len(u.record) - 64 - 256          # the single source commitment: asset, commitment, Y_0, Y_1, Y_2, z_s, z_x, z_r
    return alice, bob, u, cm


@pytest.mark.parametrize("which", [0, 1, 2])
def test_identity_and_undecodable_points_in_sigma_proofs(engine, oracle, which):
    """A prover that signs a record whose sigma-proof commitments Y_i are the all-zero encoding (rejected by
    validate_and_append_point) or do not decode (rejected after the challenge): the signature is valid, so the verdict is the
    proof's own error variant, in the reference's order (commitments before transfers)."""
    alice, bob, u, cm = _one_transfer()
    ledger = Ledger([alice, bob])
    check_same(engine, oracle, [u.sign(alice.keypair, "ok")], ledger, expect_accept=True)
    y_cm, y_xf = cm + 64 + 32 * which, XFER0 + 160 + 32 * which
    cases = [
        ([(y_cm, bytes(32))], codes.ERR_TRANSCRIPT_IDENTITY),
        ([(y_xf, bytes(32))], codes.ERR_TRANSCRIPT_IDENTITY),
        ([(y_cm, BAD_POINT)], codes.ERR_COMMITMENT_EQ),
        ([(y_xf, BAD_POINT)], codes.ERR_CT_VALIDITY),
        # both proofs broken: the commitment's proof is checked first
        ([(y_cm, BAD_POINT), (y_xf, bytes(32))], codes.ERR_COMMITMENT_EQ),
        ([(y_cm, bytes(32)), (y_xf, BAD_POINT)], codes.ERR_TRANSCRIPT_IDENTITY),
        # an identity point in a later Y of the same proof is met before the undecodable earlier one is decoded
        ([(cm + 64, BAD_POINT), (cm + 64 + 64, bytes(32))], codes.ERR_TRANSCRIPT_IDENTITY),
    ]
    for k, (edits, code) in enumerate(cases):
        bad = _resigned(u, alice.keypair, edits, "gap-bad-%d-%d" % (which, k))
        _, got = check_same(engine, oracle, [bad], ledger, expect_accept=False, expect_code=code)
        assert got[2] == 0


def test_error_variants_inside_a_batch(engine, oracle):
    """The same lying provers at a position inside a chain: the index of the failing transaction is reported and the
    transactions before it do not mask it."""
    alice, bob = Account([(NATIVE, 1000)], "gap-chain-a"), Account([(NATIVE, 0)], "gap-chain-b")
    ledger = Ledger([alice, bob])
    prover, unsigned, bal = ledger.clone(), [], 1000
    for k in range(6):
        u = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 1)]), 3, 0).build_unsigned(
            {NATIVE: bal}, alice.keypair, "gap-c%d" % k, prover.get_account(alice.pubkey))
        u.sign(alice.keypair, "gap-cs%d" % k).apply_without_verify(prover)
        bal -= 4
        unsigned.append(u)
    good = [u.sign(alice.keypair, "gap-g%d" % k) for k, u in enumerate(unsigned)]
    check_same(engine, oracle, good, ledger, expect_accept=True)
    cm = len(unsigned[0].record) - 64 - 256
    for pos, edits, code in [(4, [(cm + 64, bytes(32))], codes.ERR_TRANSCRIPT_IDENTITY), (2, [(XFER0 + 160 + 32, BAD_POINT)], codes.ERR_CT_VALIDITY),
                             (5, [(cm + 64 + 32, BAD_POINT)], codes.ERR_COMMITMENT_EQ)]:
        txs = list(good)
        txs[pos] = _resigned(unsigned[pos], alice.keypair, edits, "gap-in-%d" % pos)
        _, got = check_same(engine, oracle, txs, ledger, expect_accept=False, expect_code=code)
        assert got[2] == pos


def test_random_lying_provers(engine, oracle):
    """Differential run over seeded random lying provers: a chain of two-transfer transactions in which one transaction has one
    or two bytes of its transfers, proofs, range proof or source commitment changed BEFORE it is signed, so the signature
    holds and the error comes from the proof stages (identity / undecodable points, sigma proofs, the merged check and its
    GenericProof / RangeProof split, non-canonical scalars).  Verdict, variant and index equal the oracle's every time."""
    import random
    alice = Account([(NATIVE, 1000)], "liar-a")
    bob, carol = Account([(NATIVE, 0)], "liar-b"), Account([(NATIVE, 0)], "liar-c")
    ledger = Ledger([alice, bob, carol])
    prover, unsigned, bal = ledger.clone(), [], 1000
    for k in range(5):
        u = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 1), TransferBuilder(carol.pubkey, 2)]), 3, 0).build_unsigned(
            {NATIVE: bal}, alice.keypair, "liar-c%d" % k, prover.get_account(alice.pubkey))
        u.sign(alice.keypair, "liar-cs%d" % k).apply_without_verify(prover)
        bal -= 6
        unsigned.append(u)
    good = [u.sign(alice.keypair, "liar-g%d" % k) for k, u in enumerate(unsigned)]
    check_same(engine, oracle, good, ledger, expect_accept=True)
    body0 = 16 + 8 + 49                      # header, two extra-data lengths, version/source/fee/nonce: the transfers start here
    r = random.Random(777)
    seen = {}
    for case in range(250):
        pos = r.randrange(len(unsigned))
        n = len(unsigned[pos].record)
        edits = []
        for _ in range(r.choice((1, 1, 2))):
            off = r.randrange(body0, n)
            edits.append((off, bytes([unsigned[pos].record[off] ^ (1 << r.randrange(8))])))
        txs = list(good)
        try:
            txs[pos] = _resigned(unsigned[pos], alice.keypair, edits, "liar-%d" % case)
        except XheError:                     # the builder refuses to sign a record that does not parse (a scalar made non-canonical)
            seen["unsignable"] = seen.get("unsignable", 0) + 1
            continue
        _, got = check_same(engine, oracle, txs, ledger)
        seen[got[1]] = seen.get(got[1], 0) + 1
    print("verdict codes reached:", seen)
    assert len([k for k in seen if k != "unsignable"]) >= 4, seen


def test_multisig_index_out_of_range_is_skipped(engine, oracle):
    """verify.rs:276: a co-signature whose signer index is beyond the configured signers is not checked at all."""
    alice, bob, cos = Account([(NATIVE, 100)], "gap-ms-a"), Account([(NATIVE, 0)], "gap-ms-b"), Account([(NATIVE, 0)], "gap-ms-c")
    b = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 10)]), 1, 0)
    ledger = Ledger([alice, bob, cos])
    ledger.set_multisig_for_account(alice.pubkey, [cos.pubkey], 2)       # one signer, threshold 2: only reachable with a skipped entry
    u = b.build_unsigned({NATIVE: 100}, alice.keypair, "gap-ms", alice)
    h = u.hash()
    u.set_multisig([(0, cos.keypair.sign(h, "gap-c0")), (7, bob.keypair.sign(b"anything", "gap-junk"))])
    check_same(engine, oracle, [u.sign(alice.keypair, "gap-ms-f")], ledger, expect_accept=True)
    # the in-range entry is still checked
    u2 = b.build_unsigned({NATIVE: 100}, alice.keypair, "gap-ms2", alice)
    u2.set_multisig([(0, cos.keypair.sign(b"not the hash", "gap-c1")), (200, bob.keypair.sign(b"anything", "gap-junk2"))])
    check_same(engine, oracle, [u2.sign(alice.keypair, "gap-ms-f2")], ledger, expect_accept=False, expect_code=codes.ERR_SIGNATURE)
    # two out-of-range entries: nothing is checked, the count alone matches the threshold
    u3 = b.build_unsigned({NATIVE: 100}, alice.keypair, "gap-ms3", alice)
    u3.set_multisig([(9, bob.keypair.sign(b"x", "gap-j3")), (10, bob.keypair.sign(b"y", "gap-j4"))])
    check_same(engine, oracle, [u3.sign(alice.keypair, "gap-ms-f3")], ledger, expect_accept=True)


def test_transfer_with_extra_data(engine, oracle):
    """ExtraData bytes only enter the signed pre-image (verify.rs:638-640) and move everything behind them."""
    alice, bob = Account([(NATIVE, 100), (ASSET55, 9)], "gap-x-a"), Account([(NATIVE, 0), (ASSET55, 0)], "gap-x-b")
    ledger = Ledger([alice, bob])
    ts = [TransferBuilder(bob.pubkey, 3, NATIVE, b"memo: rent for march"), TransferBuilder(bob.pubkey, 4, ASSET55), TransferBuilder(bob.pubkey, 5, NATIVE, bytes(range(200)))]
    tx = TransactionBuilder(1, alice.pubkey, ("transfers", ts), 2, 0).build({NATIVE: 100, ASSET55: 9}, alice.keypair, "gap-extra", alice)
    lg, _ = check_same(engine, oracle, [tx], ledger, expect_accept=True)
    assert lg.balance_decrypts_to(alice.pubkey, NATIVE, 90) and lg.balance_decrypts_to(bob.pubkey, ASSET55, 4)
    r = bytearray(tx.record)
    r[16 + 12 + 49 + 160 + 5] ^= 1          # inside the first transfer's extra data
    check_same(engine, oracle, [Transaction(bytes(r))], ledger, expect_accept=False, expect_code=codes.ERR_SIGNATURE)
    plain = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 1)]), 1, 0).build({NATIVE: 100, ASSET55: 9}, alice.keypair, "gap-plain", alice)
    check_same(engine, oracle, [plain], ledger, expect_accept=True)


def test_contract_records(engine, oracle):
    """CallContract (deposits leave the sender's balances in the clear) and DeployContract (fee only)."""
    alice = Account([(NATIVE, 500), (ASSET55, 40)], "gap-k-a")
    ledger = Ledger([alice])
    call = TransactionBuilder(1, alice.pubkey, ("call_contract", bytes([7]) * 32, [(NATIVE, 20), (ASSET55, 15)], b"\x01\x02params"), 5, 0).build(
        {NATIVE: 500, ASSET55: 40}, alice.keypair, "gap-call", alice)
    lg, _ = check_same(engine, oracle, [call], ledger, expect_accept=True)
    assert lg.balance_decrypts_to(alice.pubkey, NATIVE, 475) and lg.balance_decrypts_to(alice.pubkey, ASSET55, 25)
    after = ledger.clone()
    call.apply_without_verify(after)
    deploy = TransactionBuilder(1, alice.pubkey, ("deploy_contract", b"module bytes" * 40), 7, 0).build({NATIVE: 475}, alice.keypair, "gap-deploy", after.get_account(alice.pubkey))
    lg, _ = check_same(engine, oracle, [call, deploy], ledger, expect_accept=True)
    assert lg.balance_decrypts_to(alice.pubkey, NATIVE, 468)
    check_same(engine, oracle, [deploy], ledger, expect_accept=False, expect_code=codes.ERR_GENERIC_PROOF)
    nodep = TransactionBuilder(1, alice.pubkey, ("call_contract", bytes([9]) * 32, [], b""), 1, 0).build({NATIVE: 500}, alice.keypair, "gap-call0", alice)
    check_same(engine, oracle, [nodep], ledger, expect_accept=True)
    # a changed deposit is covered by the signature; re-signed it breaks the commitment proof
    r = bytearray(call.record)
    r[16 + 49 + 32 + 32 + 7] ^= 1           # low byte of the native deposit
    check_same(engine, oracle, [Transaction(bytes(r))], ledger, expect_accept=False, expect_code=codes.ERR_SIGNATURE)


def _with_range_proof_rounds(u, delta_rounds):
    """The unsigned record with |delta_rounds| (L, R) pairs appended to / removed from its range proof."""
    rec = bytearray(u.record)
    n_items = int.from_bytes(rec[2:4], "little")
    rp_len = int.from_bytes(rec[12:14], "little")
    rp_off = 16 + 4 * n_items + 49 + 320 * n_items
    tail = rp_off + rp_len - 64                  # a, b
    if delta_rounds > 0:
        pair = bytes(rec[tail - 64:tail])
        rec[tail:tail] = pair * delta_rounds
    else:
        del rec[tail + 64 * delta_rounds:tail]
    rec[12:14] = (rp_len + 64 * delta_rounds).to_bytes(2, "little")
    return TransactionUnsigned(bytes(rec))


@pytest.mark.parametrize("delta", [1, -1, 3])
def test_range_proof_length_disagrees_with_value_count(engine, oracle, delta):
    """A well-formed range proof with more or fewer inner-product rounds than log2(64 m): the transcript has no template
    for that shape (the serial transcript kernel runs) and the verdict is RangeProof, alone and beside a valid transaction."""
    alice, bob, u, _ = _one_transfer()
    ledger = Ledger([alice, bob])
    bad = _with_range_proof_rounds(u, delta).sign(alice.keypair, "gap-rp-%d" % delta)
    check_same(engine, oracle, [bad], ledger, expect_accept=False, expect_code=codes.ERR_RANGE_PROOF)
    carol = Account([(NATIVE, 50)], "gap-rp-c")
    burn = TransactionBuilder(1, carol.pubkey, ("burn", NATIVE, 5), 1, 0).build({NATIVE: 50}, carol.keypair, "gap-rp-burn", carol)
    check_same(engine, oracle, [burn, bad], Ledger([alice, bob, carol]), expect_accept=False, expect_code=codes.ERR_RANGE_PROOF)
    check_same(engine, oracle, [burn, u.sign(alice.keypair, "gap-rp-ok")], Ledger([alice, bob, carol]), expect_accept=True)


def test_255_transfers(engine, oracle):
    """benches/tx.rs:108-110: one transaction with 255 transfers, 256 value commitments, 16 384 generators per side."""
    alice = Account([(NATIVE, 100000)], "gap-255-a")
    dests = [Account([(NATIVE, 0)], "gap-255-d%d" % i) for i in range(5)]
    ts = [TransferBuilder(dests[i % 5].pubkey, 1 + (i % 7)) for i in range(255)]
    u = TransactionBuilder(1, alice.pubkey, ("transfers", ts), 3, 0).build_unsigned({NATIVE: 100000}, alice.keypair, "gap-255", alice)
    ledger = Ledger([alice] + dests)
    lg, _ = check_same(engine, oracle, [u.sign(alice.keypair, "gap-255-s")], ledger, expect_accept=True)
    assert lg.balance_decrypts_to(alice.pubkey, NATIVE, 100000 - 3 - sum(1 + (i % 7) for i in range(255)))
    rp = 16 + 4 * 255 + 49 + 320 * 255
    bad = _resigned(u, alice.keypair, [(rp + 128, bytes([u.record[rp + 128] ^ 1]))], "gap-255-bad")
    check_same(engine, oracle, [bad], ledger, expect_accept=False, expect_code=codes.ERR_RANGE_PROOF)


def _debug_read(engine, case, what, cap):
    lib = engine.lib
    lib.xhe_batch_upload.restype = lib.xhe_batch_verify.restype = ctypes.c_int32
    lib.xhe_batch_debug_read.restype = ctypes.c_int64
    ctx, h, v = ctypes.c_void_p(engine.ctx), ctypes.c_void_p(), Verdict()
    engine._check(lib.xhe_batch_upload(ctx, case.blob, case.offsets, ctypes.c_uint64(len(case.records)), ctypes.byref(case.view), ctypes.byref(h), ctypes.byref(v)))
    try:
        assert v.accept == 1
        engine._check(lib.xhe_batch_verify(ctx, h, ctypes.byref(v), case.seed))
        out = buf(cap)
        n = lib.xhe_batch_debug_read(ctx, h, ctypes.c_int32(what), out, ctypes.c_uint64(cap))
        assert n >= 0
        return bytes(out)[:32 * n], (v.accept, v.err_code)
    finally:
        lib.xhe_batch_free(ctx, h)


def test_challenges_equal_the_fixtures(engine):
    """Every Fiat-Shamir challenge the kernels derive (sigma c and w, range-proof y, z, x, w, u_j) against the challenges
    stored with the committed transactions: a wrong transcript shows here, not only as a rejected batch."""
    checked = 0
    for case in tx_cases():
        chal = case["expect"]["challenges"]
        if not case["records"] or any(c is None for c in chal) or case["expect"]["err_code"] in (codes.ERR_MALFORMED,):
            continue
        fc = FlatCase(case)
        want = b"".join(bytes.fromhex(c) for c in chal)
        got, _ = _debug_read(engine, fc, 1, len(want) + 64)
        assert got == want, case["name"]
        checked += 1
    assert checked >= 10


def test_generators_equal_the_fixtures(engine):
    p = primitives()
    case = next(c for c in tx_cases() if c["name"] == "transfer_T1")
    got, _ = _debug_read(engine, FlatCase(case), 2, 32 * (2 + 2 * 4))
    pts = [got[32 * i:32 * i + 32].hex() for i in range(10)]
    assert pts[0] == p["rfc9496_generator_multiples"][1]
    assert pts[1] == p["pedersen_blinding_generator"]
    assert pts[2:6] == p["bulletproof_generators_party0"]["G"]
    assert pts[6:10] == p["bulletproof_generators_party0"]["H"]
