#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.

The reference (a Rust crate whose arithmetic lives in un-vendored git dependencies) cannot be built
or run in this environment and holds no golden vectors of its own, so these fixtures are NOT outputs
of the reference.  They are:
  primitives.json   public known answers (RFC 9496 generator multiples, the published Pedersen
                    blinding generator, Merlin's "test protocol" vectors, BLAKE3 official digests)
                    plus values of the Python big-integer specification (oracle/spec_ristretto.py)
                    and of the CPU oracle (first Bulletproofs generators, seeded multiscalar sums);
  tx_cases.json     deterministic transactions of every kind the verification path accepts, built by
                    the CPU builder of libxhe.so from fixed seeds, with the CPU oracle's verdict,
                    Fiat-Shamir challenges and resulting ledger bytes, and tampered variants with
                    the expected error code.
They freeze today's oracle behaviour (any later change to oracle or kernels must reproduce them
byte for byte) and give the GPU tests fixed vectors that do not depend on a CPU library at run time.

usage: python tests/golden/make_golden.py     (CPU only; needs libxhe.so and oracle/liboracle.so)
"""
import ctypes
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers  # noqa: E402
from oracle import spec_ristretto as S  # noqa: E402
from xelis_he_amd import codes  # noqa: E402
from xelis_he_amd.api import (NATIVE, Account, Ledger, LedgerOut, Transaction, TransactionBuilder, TransactionUnsigned,  # noqa: E402
                              TransferBuilder, Verdict)

ASSET55 = bytes([55]) * 32
BASEPOINT = "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"
SEED = hashlib.sha3_256(b"golden batch factors").digest()


def msm_inputs(oracle, seed, n):
    """The seeded multiscalar inputs the tests regenerate: a pool of min(n, 32) points k_i * B and n scalars."""
    r = random.Random(seed)
    pool = []
    for _ in range(min(max(n, 1), 32)):
        o, idf = helpers.buf(32), helpers.buf(1)
        oracle.xo_msm_vartime(1, helpers.le32(r.randrange(S.L)), bytes.fromhex(BASEPOINT), o, idf)
        pool.append(bytes(o))
    pts = b"".join(pool[i % len(pool)] for i in range(n))
    scal = b"".join(helpers.le32(r.randrange(S.L)) for _ in range(n))
    return scal, pts


def primitives(oracle):
    out = {}
    out["rfc9496_generator_multiples"] = [
        "0000000000000000000000000000000000000000000000000000000000000000",
        BASEPOINT,
        "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919",
        "94741f5d5d52755ece4f23f044ee27d5d1ea1e2bd196b462166b16152a9d0259",
    ]
    h = helpers.buf(32)
    oracle.xo_pedersen_h(h)
    out["pedersen_blinding_generator"] = bytes(h).hex()
    g, hh = helpers.buf(32 * 4), helpers.buf(32 * 4)
    oracle.xo_bp_generators(4, g, hh)
    out["bulletproof_generators_party0"] = {"G": [bytes(g)[32 * i:32 * i + 32].hex() for i in range(4)],
                                            "H": [bytes(hh)[32 * i:32 * i + 32].hex() for i in range(4)]}
    out["blake3"] = {"": "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262",
                     "616263": "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"}
    out["merlin_test_protocol"] = [
        {"ops": [["A", "some label", "some data".encode().hex()], ["C", "challenge", 32]],
         "challenge": "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"},
    ]
    # encodings: spec multiples and rejected byte strings
    r = random.Random(9496)
    enc = []
    for _ in range(24):
        k = r.randrange(S.L)
        o, idf = helpers.buf(32), helpers.buf(1)
        oracle.xo_msm_vartime(1, helpers.le32(k), bytes.fromhex(BASEPOINT), o, idf)
        assert bytes(o) == S.encode(S.mul(k, S.BASE))
        enc.append({"k": hex(k), "encoding": bytes(o).hex()})
    out["spec_generator_multiples"] = enc
    bad = [helpers.le32(S.P), helpers.le32(1), helpers.le32(S.P - 1), bytes([255] * 32)] + [r.randbytes(32) for _ in range(40)]
    ok = helpers.buf(len(bad))
    xyzt = helpers.buf(128 * len(bad))
    oracle.xo_ristretto_decompress_batch(len(bad), b"".join(bad), xyzt, ok)
    out["decode_accepts"] = [{"bytes": b.hex(), "ok": int(ok[i]), "spec_ok": int(S.decode(b) is not None)} for i, b in enumerate(bad)]
    msm = []
    for n in (1, 2, 189, 190, 499, 800, 4096):
        scal, pts = msm_inputs(oracle, "golden-msm-%d" % n, n)
        o, idf = helpers.buf(32), helpers.buf(1)
        assert oracle.xo_msm_vartime(n, scal, pts, o, idf) == 0
        msm.append({"n": n, "seed": "golden-msm-%d" % n, "result": bytes(o).hex(), "input_sha3_256": hashlib.sha3_256(scal + pts).hexdigest()})
    out["msm"] = msm
    return out


def flat_ledger(f):
    v = f["view"]
    return {
        "account_keys": f["account_keys"].hex(), "nonces": [int(x) for x in f["nonces"]][:len(f["keys"])],
        "balance_account": [int(x) for x in f["balance_account"]][:len(f["slots"])], "balance_assets": f["balance_assets"].hex(),
        "balance_cts": f["balance_cts"].hex(),
        "ms_account": [int(x) for x in f["ms_account"]][:int(v.n_multisig)], "ms_threshold": f["ms_threshold"].hex(),
        "ms_offsets": [int(x) for x in f["ms_offsets"]], "ms_signers": f["ms_signers"].hex(),
    }


def run_case(oracle, name, txs, ledger, single=False):
    f = ledger.flatten()
    blob = b"".join(t.record for t in txs)
    offs = [0]
    for t in txs:
        offs.append(offs[-1] + len(t.record))
    offsets = (ctypes.c_uint64 * len(offs))(*offs)
    out_cts = (ctypes.c_uint8 * max(64 * len(f["slots"]), 1))()
    out_nonces = (ctypes.c_uint64 * max(len(f["keys"]), 1))()
    ms_set = (ctypes.c_int64 * max(len(f["keys"]), 1))()
    out = LedgerOut(out_cts, out_nonces, ms_set)
    v = Verdict()
    oracle.xo_verify_batch(blob, offsets, ctypes.c_uint64(len(txs)), ctypes.byref(f["view"]), ctypes.byref(out), ctypes.byref(v), SEED)
    chal = []
    for t in txs:
        cb = helpers.buf(32 * 64)
        n = oracle.xo_tx_challenges(t.record, ctypes.c_uint64(len(t.record)), cb, ctypes.c_uint64(64))
        chal.append(bytes(cb)[:32 * max(n, 0)].hex() if n >= 0 else None)
    case = {"name": name, "ledger": flat_ledger(f), "records": [t.record.hex() for t in txs], "seed": SEED.hex(),
            "expect": {"accept": int(v.accept), "err_code": int(v.err_code), "err_name": codes.NAMES.get(int(v.err_code)), "first_bad_tx": int(v.first_bad_tx),
                       "challenges": chal}}
    if v.accept:
        case["expect"]["out_cts"] = bytes(out_cts)[:64 * len(f["slots"])].hex()
        case["expect"]["multisig_set_by"] = [int(x) for x in ms_set][:len(f["keys"])]
    return case


def mutate(tx, offset, new_bytes):
    r = bytearray(tx.record)
    r[offset:offset + len(new_bytes)] = new_bytes
    return Transaction(bytes(r))


def tx_cases(oracle):
    cases = []
    alice, bob = Account([(NATIVE, 100)], "g-alice"), Account([(NATIVE, 0)], "g-bob")
    t1 = TransactionBuilder(1, alice.pubkey, ("transfers", [TransferBuilder(bob.pubkey, 10)]), 1, 0).build({NATIVE: 100}, alice.keypair, "g-t1", alice)
    cases.append(run_case(oracle, "transfer_T1", [t1], Ledger([alice, bob])))

    # eight transfers of one asset: m = 16, three identity dud commitments... (A=1, T=8 -> 9 values, 7 duds)
    dests = [Account([(NATIVE, 0)], "g-d%d" % i) for i in range(8)]
    carol = Account([(NATIVE, 1000)], "g-carol")
    t8 = TransactionBuilder(1, carol.pubkey, ("transfers", [TransferBuilder(d.pubkey, 3 + i) for i, d in enumerate(dests)]), 2, 0).build(
        {NATIVE: 1000}, carol.keypair, "g-t8", carol)
    cases.append(run_case(oracle, "transfer_T8", [t8], Ledger([carol] + dests)))

    dave = Account([(NATIVE, 100)], "g-dave")
    burn = TransactionBuilder(1, dave.pubkey, ("burn", NATIVE, 10), 1, 0).build({NATIVE: 100}, dave.keypair, "g-burn", dave)
    cases.append(run_case(oracle, "burn_native", [burn], Ledger([dave])))
    erin = Account([(NATIVE, 1), (ASSET55, 50)], "g-erin")
    burn2 = TransactionBuilder(1, erin.pubkey, ("burn", ASSET55, 50), 1, 0).build({NATIVE: 1, ASSET55: 50}, erin.keypair, "g-burn2", erin)
    cases.append(run_case(oracle, "burn_non_native", [burn2], Ledger([erin])))

    frank, s1, s2 = Account([(NATIVE, 100)], "g-frank"), Account([(NATIVE, 0)], "g-s1"), Account([(NATIVE, 0)], "g-s2")
    setup = TransactionBuilder(1, frank.pubkey, ("multisig", [s1.pubkey, s2.pubkey], 2), 1, 0).build({NATIVE: 100}, frank.keypair, "g-setup", frank)
    cases.append(run_case(oracle, "multisig_setup", [setup], Ledger([frank, s1, s2])))

    gina, hal = Account([(NATIVE, 100)], "g-gina"), Account([(NATIVE, 0)], "g-hal")
    u = TransactionBuilder(1, gina.pubkey, ("transfers", [TransferBuilder(hal.pubkey, 10)]), 1, 0).build_unsigned({NATIVE: 100}, gina.keypair, "g-ms", gina)
    hh = u.hash()
    u.set_multisig([(0, s1.keypair.sign(hh, "g-c0")), (1, s2.keypair.sign(hh, "g-c1"))])
    signed = u.sign(gina.keypair, "g-final")
    lg = Ledger([gina, hal, s1, s2])
    lg.set_multisig_for_account(gina.pubkey, [s1.pubkey, s2.pubkey], 2)
    cases.append(run_case(oracle, "multisig_signed_transfer", [signed], lg))

    # a chain of dependent transactions plus a two-asset transfer in one batch
    ivy = Account([(NATIVE, 1000), (ASSET55, 20)], "g-ivy")
    jon = Account([(NATIVE, 0), (ASSET55, 0)], "g-jon")
    ledger = Ledger([ivy, jon])
    prover, txs, bal = ledger.clone(), [], 1000
    for k in range(3):
        tx = TransactionBuilder(1, ivy.pubkey, ("transfers", [TransferBuilder(jon.pubkey, 1)]), 3, 0).build(
            {NATIVE: bal, ASSET55: 20}, ivy.keypair, "g-chain%d" % k, prover.get_account(ivy.pubkey))
        tx.apply_without_verify(prover)
        bal -= 4
        txs.append(tx)
    two = TransactionBuilder(1, ivy.pubkey, ("transfers", [TransferBuilder(jon.pubkey, 5), TransferBuilder(jon.pubkey, 7, ASSET55)]), 1, 0).build(
        {NATIVE: bal, ASSET55: 20}, ivy.keypair, "g-two", prover.get_account(ivy.pubkey))
    two.apply_without_verify(prover)
    txs.append(two)
    cases.append(run_case(oracle, "chain_and_two_assets", txs, ledger))

    # rejected variants
    n = len(t1.record)
    pre = 16 + 4
    cases.append(run_case(oracle, "reject_zero_signature", [mutate(t1, n - 64, bytes(64))], Ledger([alice, bob])))
    cases.append(run_case(oracle, "reject_fee_changed", [mutate(t1, pre + 33, (2).to_bytes(8, "big"))], Ledger([alice, bob])))
    cases.append(run_case(oracle, "reject_nonce", [mutate(t1, pre + 41, (1).to_bytes(8, "big"))], Ledger([alice, bob])))
    cases.append(run_case(oracle, "reject_commitment_asset", [mutate(t1, n - 64 - 256, bytes([1]) * 32)], Ledger([alice, bob])))
    cases.append(run_case(oracle, "reject_bad_point", [mutate(t1, pre + 49 + 64, bytes([255]) * 32)], Ledger([alice, bob])))
    cases.append(run_case(oracle, "reject_chain_gap", txs[:1] + txs[2:], ledger))
    bad = list(txs)
    bad[2] = mutate(bad[2], pre + 49 + 320 + 128, bytes([bad[2].record[pre + 49 + 320 + 128] ^ 1]))
    cases.append(run_case(oracle, "reject_signature_inside_batch", bad, ledger))
    liar_a, liar_b = Account([(NATIVE, 5)], "g-liar"), Account([(NATIVE, 0)], "g-liar-b")
    lie = TransactionBuilder(1, liar_a.pubkey, ("transfers", [TransferBuilder(liar_b.pubkey, 50)]), 1, 0).build({NATIVE: 100}, liar_a.keypair, "g-lie", liar_a)
    cases.append(run_case(oracle, "reject_overspend", [lie], Ledger([liar_a, liar_b])))
    ms_missing = Ledger([alice, bob, s1])
    ms_missing.set_multisig_for_account(alice.pubkey, [s1.pubkey], 1)
    cases.append(run_case(oracle, "reject_multisig_required", [t1], ms_missing))
    cases.append(run_case(oracle, "empty_batch", [], Ledger([alice, bob])))

    # lying provers: the record is edited before it is signed, so the signature is valid and the proof's own error variant shows
    # (proofs.rs:151-179,293-317): Transcript(IdentityPoint), CommitmentEqProof, CiphertextValidityProof
    def resigned(unsigned, keypair, edits, seed):
        r = bytearray(unsigned.record)
        for off, data in edits:
            r[off:off + len(data)] = data
        return TransactionUnsigned(bytes(r)).sign(keypair, seed)
    kim, lee = Account([(NATIVE, 100)], "g-kim"), Account([(NATIVE, 0)], "g-lee")
    ku = TransactionBuilder(1, kim.pubkey, ("transfers", [TransferBuilder(lee.pubkey, 10)]), 1, 0).build_unsigned({NATIVE: 100}, kim.keypair, "g-ku", kim)
    cm, xf_proof, bad_point = len(ku.record) - 64 - 256, 16 + 4 + 49 + 160, bytes([255]) * 32
    kl = Ledger([kim, lee])
    cases.append(run_case(oracle, "reject_identity_in_eq_proof", [resigned(ku, kim.keypair, [(cm + 64, bytes(32))], "g-k1")], kl))
    cases.append(run_case(oracle, "reject_identity_in_validity_proof", [resigned(ku, kim.keypair, [(xf_proof + 32, bytes(32))], "g-k2")], kl))
    cases.append(run_case(oracle, "reject_undecodable_eq_proof_point", [resigned(ku, kim.keypair, [(cm + 64 + 32, bad_point)], "g-k3")], kl))
    cases.append(run_case(oracle, "reject_undecodable_validity_proof_point", [resigned(ku, kim.keypair, [(xf_proof + 64, bad_point)], "g-k4")], kl))
    # a range proof with one inner-product round too many (verify.rs:504-514 -> RangeProof)
    rec = bytearray(ku.record)
    rp_len = int.from_bytes(rec[12:14], "little")
    tail = 16 + 4 + 49 + 320 + rp_len - 64
    rec[tail:tail] = bytes(rec[tail - 64:tail])
    rec[12:14] = (rp_len + 64).to_bytes(2, "little")
    cases.append(run_case(oracle, "reject_range_proof_rounds", [TransactionUnsigned(bytes(rec)).sign(kim.keypair, "g-k5")], kl))
    # multisig: an entry whose signer index is beyond the configuration is skipped (verify.rs:276)
    mu = TransactionBuilder(1, kim.pubkey, ("transfers", [TransferBuilder(lee.pubkey, 10)]), 1, 0).build_unsigned({NATIVE: 100}, kim.keypair, "g-mu", kim)
    mu.set_multisig([(0, s1.keypair.sign(mu.hash(), "g-m0")), (7, s2.keypair.sign(b"unchecked", "g-m7"))])
    ml = Ledger([kim, lee, s1])
    ml.set_multisig_for_account(kim.pubkey, [s1.pubkey], 2)
    cases.append(run_case(oracle, "multisig_index_out_of_range_skipped", [mu.sign(kim.keypair, "g-mf")], ml))
    # extra data, contract call with deposits, contract deployment
    ex = TransactionBuilder(1, kim.pubkey, ("transfers", [TransferBuilder(lee.pubkey, 3, NATIVE, b"golden memo"), TransferBuilder(lee.pubkey, 4)]), 2, 0).build(
        {NATIVE: 100}, kim.keypair, "g-extra", kim)
    cases.append(run_case(oracle, "transfer_with_extra_data", [ex], kl))
    moe = Account([(NATIVE, 500), (ASSET55, 40)], "g-moe")
    call = TransactionBuilder(1, moe.pubkey, ("call_contract", bytes([7]) * 32, [(NATIVE, 20), (ASSET55, 15)], b"params"), 5, 0).build(
        {NATIVE: 500, ASSET55: 40}, moe.keypair, "g-call", moe)
    after = Ledger([moe]).clone()
    call.apply_without_verify(after)
    deploy = TransactionBuilder(1, moe.pubkey, ("deploy_contract", b"golden module" * 9), 7, 0).build({NATIVE: 475}, moe.keypair, "g-deploy", after.get_account(moe.pubkey))
    cases.append(run_case(oracle, "contract_call_then_deploy", [call, deploy], Ledger([moe])))
    return cases


def main():
    oracle = helpers.load_oracle()
    with open(os.path.join(HERE, "primitives.json"), "w") as f:
        json.dump(primitives(oracle), f, indent=1)
    cases = tx_cases(oracle)
    with open(os.path.join(HERE, "tx_cases.json"), "w") as f:
        json.dump({"cases": cases}, f, indent=1)
    for c in cases:
        print("%-32s accept=%d err=%s first_bad=%d txs=%d" % (c["name"], c["expect"]["accept"], c["expect"]["err_name"], c["expect"]["first_bad_tx"], len(c["records"])))


if __name__ == "__main__":
    main()
