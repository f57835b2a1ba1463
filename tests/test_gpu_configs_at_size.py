"""The two benchmark configurations that are quoted on eight GPUs, at the size ONE GPU gets of them:
  * the mixed batch (groups of 16: multisig set-ups, one- and two-asset transfers, burns of both assets, co-signed
    transfers; /root/reference/src/lib.rs:492-703, shapes of /root/reference/benches/tx.rs:252-343) at 100 000
    transactions, and at 8 192 transactions against the oracle (verdict, every ledger byte, multisig_set_by, and a seeded
    differential run over corrupted copies and over lying provers whose signatures hold);
  * the one-transfer batch at 125 000 transactions (a million transactions over eight GPUs).
At full size the checks are properties that need no CPU verifier: accept, the ledger of the GPU's own apply path
(Transaction::apply_without_verify, /root/reference/src/tx/verify.rs:545-619) and of the CPU apply path on a sample of
account groups, one corrupted signature reported at its index, a dropped transaction as GenericProof.
Integer work: the bar is bit-exact.  The batches come from the GPU-assisted generator, pinned to the CPU prover byte for
byte below (mixed shapes) and in test_gpu_prover.py (transfers)."""
import ctypes
import random
from concurrent.futures import ThreadPoolExecutor

import pytest

from test_gpu_fullsize import _RawView, _verify
from xelis_he_amd import codes
from xelis_he_amd.api import LedgerOut, SyntheticBatch, Verdict, _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import xelis_he_amd
    e = xelis_he_amd.Engine(0)
    yield e
    e.close()


def test_mixed_gpu_prover_matches_cpu_prover(engine):
    """Range proofs over 1, 2 and 4 value commitments proved on the GPU, co-signatures and signatures over them: the same
    records as the CPU generator's for the same seed."""
    cpu = SyntheticBatch(96, seed=b"mixed-prover", mixed=True, threads=8)
    gpu = SyntheticBatch(96, seed=b"mixed-prover", mixed=True, threads=8, engine=engine)
    assert list(gpu.offsets) == list(cpu.offsets)
    if gpu.records != cpu.records:
        first = next(i for i in range(len(cpu.records)) if cpu.records[i] != gpu.records[i])
        tx = max(k for k in range(96) if cpu.offsets[k] <= first)
        raise AssertionError("records differ first at byte %d (tx %d, offset %d inside the record)" % (first, tx, first - cpu.offsets[tx]))
    v, _ = _verify(engine, gpu.records, gpu.offsets, gpu.n_txs, gpu.view)
    assert v.accept == 1


def _run(which, engine, oracle, sb, records, want_ledger=True):
    n_bal, n_acc = sb.view.contents.n_balances, sb.view.contents.n_accounts
    cts = (ctypes.c_uint8 * (64 * n_bal))()
    ms = (ctypes.c_int64 * n_acc)(*([-7] * n_acc))
    out = LedgerOut(cts, (ctypes.c_uint64 * n_acc)(), ms)
    v = Verdict()
    if which == "oracle":
        oracle.xo_verify_batch(records, sb.offsets, ctypes.c_uint64(sb.n_txs), sb.view, ctypes.byref(out), ctypes.byref(v), b"\x02" * 32)
    else:
        engine._check(engine.lib.xhe_verify_batch(engine.ctx, records, sb.offsets, sb.n_txs, sb.view, ctypes.byref(out), ctypes.byref(v), b"\x01" * 32))
    return (v.accept, v.err_code, v.first_bad_tx), (bytes(cts), list(ms)) if want_ledger else None


def _resigned_mixed(lib, sb, secrets, tx, r, seed):
    """Transaction tx of a mixed batch with one or two bits of its pre-image flipped behind the header fields (version, source,
    fee, nonce stay), co-signed and signed again with the accounts' secrets: same length, valid signatures."""
    import hashlib
    rec = sb.records[sb.offsets[tx]:sb.offsets[tx + 1]]
    g, pos = divmod(tx, 16)
    n_items, n_ms = int.from_bytes(rec[2:4], "little"), int.from_bytes(rec[6:8], "little")
    pre_off = 16 + (4 * n_items if rec[0] == codes.TX_TRANSFERS else 0)
    pre_end = len(rec) - 65 * n_ms - 64
    pre = bytearray(rec[pre_off:pre_end])
    for _ in range(r.choice((1, 1, 2))):
        pre[r.randrange(49, len(pre))] ^= 1 << r.randrange(8)
    hdr = bytearray(rec[:pre_off]); hdr[1] = 0; hdr[6] = hdr[7] = 0
    unsigned = bytes(hdr) + bytes(pre) + bytes(64)
    secret = lambda a: secrets[32 * (6 * g + a):32 * (6 * g + a) + 32]
    source = 2 + pos if pos < 2 else (2 + pos - 14 if pos >= 14 else 0)
    ids, sigs = b"", b""
    if n_ms:
        h = (ctypes.c_uint8 * 32)()
        if lib.xhe_tx_hash(unsigned, ctypes.c_uint64(len(unsigned)), h) != 0:
            return None
        for c in range(2):
            sig = (ctypes.c_uint8 * 64)()
            lib.xhe_sign(secret(4 + c), bytes(h), ctypes.c_uint64(32), hashlib.sha3_256(seed + b"/co%d" % c).digest(), sig)
            ids += bytes([c]); sigs += bytes(sig)
    out, n = (ctypes.c_uint8 * (len(rec) + 64))(), ctypes.c_uint64()
    rc = lib.xhe_tx_finalize(unsigned, ctypes.c_uint64(len(unsigned)), 1 if n_ms else 0, n_ms, ids, sigs, secret(source), hashlib.sha3_256(seed).digest(), out,
                             ctypes.c_uint64(len(out)), ctypes.byref(n))
    return bytes(out)[:n.value] if rc == 0 else None


@pytest.mark.timeout(900)
def test_mixed_8192_against_the_oracle(engine, oracle):
    """8 192 mixed transactions (512 account groups): verdict, every ledger byte and multisig_set_by equal the oracle's; then
    120 seeded corruptions (one to three flipped bits anywhere in one record: headers, keys, proofs, co-signatures, signatures)
    -- verdict, variant and index equal the oracle's every time."""
    sb = SyntheticBatch(8192, seed=b"mixed-oracle", mixed=True, threads=16, engine=engine)
    got, got_ledger = _run("gpu", engine, oracle, sb, sb.records)
    ref, ref_ledger = _run("oracle", engine, oracle, sb, sb.records)
    assert got == ref == (1, codes.OK, -1)
    assert got_ledger[0] == ref_ledger[0]
    assert got_ledger[1] == ref_ledger[1]
    assert sum(1 for x in ref_ledger[1] if x >= 0) == 2 * 512          # two set-ups per group
    r = random.Random(8192)
    cases = []
    for _ in range(120):
        tx = r.randrange(sb.n_txs)
        bad = bytearray(sb.records)
        for _ in range(r.choice((1, 1, 2, 3))):
            bad[r.randrange(sb.offsets[tx], sb.offsets[tx + 1])] ^= 1 << r.randrange(8)
        cases.append((tx, bytes(bad)))
    with ThreadPoolExecutor(16) as pool:                                  # the oracle is sequential per call; its calls are independent
        refs = list(pool.map(lambda c: _run("oracle", engine, oracle, sb, c[1], False)[0], cases))
    seen = {}
    for (tx, bad), ref in zip(cases, refs):
        got = _run("gpu", engine, oracle, sb, bad, False)[0]
        assert got == ref, (tx, got, ref)
        seen[got[1]] = seen.get(got[1], 0) + 1
    print("verdict codes reached:", {codes.NAMES.get(k, k): v for k, v in seen.items()})
    assert len(seen) >= 4, seen
    # 80 lying provers: one to two bits of a record's pre-image (keys, ciphertexts, sigma proofs, range proof, commitments)
    # changed BEFORE its co-signers and its source sign it, so the signatures hold and the error comes from the proof stages
    lib = _lib()
    lib.xhe_synth_secrets.restype = ctypes.c_void_p
    secrets = ctypes.string_at(lib.xhe_synth_secrets(sb.handle), 32 * 6 * (sb.n_txs // 16))
    cases, unsignable = [], 0
    for case in range(80):
        tx = r.randrange(sb.n_txs)
        rec = _resigned_mixed(lib, sb, secrets, tx, r, b"liar-%d" % case)
        if rec is None:                               # a scalar made non-canonical: the builder does not sign a record that does not parse
            unsignable += 1
            continue
        assert len(rec) == sb.offsets[tx + 1] - sb.offsets[tx]
        cases.append((tx, sb.records[:sb.offsets[tx]] + rec + sb.records[sb.offsets[tx + 1]:]))
    with ThreadPoolExecutor(16) as pool:
        refs = list(pool.map(lambda c: _run("oracle", engine, oracle, sb, c[1], False)[0], cases))
    seen = {}
    for (tx, bad), ref in zip(cases, refs):
        got = _run("gpu", engine, oracle, sb, bad, False)[0]
        assert got == ref, (tx, got, ref)
        seen[got[1]] = seen.get(got[1], 0) + 1
    print("lying provers, verdict codes reached:", {codes.NAMES.get(k, k): v for k, v in seen.items()}, "unsignable:", unsignable)
    assert len(seen) >= 3, seen
    sb.close()


def _apply_on_gpu(engine, sb):
    lib = _lib()
    lib.xhe_batch_upload.restype = lib.xhe_batch_apply.restype = lib.xhe_batch_read_ledger.restype = ctypes.c_int32
    view = sb.view.contents
    ctx, h, av = ctypes.c_void_p(engine.ctx), ctypes.c_void_p(), Verdict()
    engine._check(lib.xhe_batch_upload(ctx, sb.records, sb.offsets, ctypes.c_uint64(sb.n_txs), sb.view, ctypes.byref(h), ctypes.byref(av)))
    try:
        engine._check(lib.xhe_batch_apply(ctx, h, ctypes.byref(av)))
        assert av.accept == 1
        acts = (ctypes.c_uint8 * (64 * view.n_balances))()
        out = LedgerOut(acts, (ctypes.c_uint64 * view.n_accounts)(), (ctypes.c_int64 * view.n_accounts)())
        engine._check(lib.xhe_batch_read_ledger(ctx, h, ctypes.byref(out)))
    finally:
        lib.xhe_batch_free(ctx, h)
    return bytes(acts)


def _without(sb, drop):
    offs = [sb.offsets[i] for i in range(sb.n_txs + 1)]
    cut = offs[drop + 1] - offs[drop]
    records = sb.records[:offs[drop]] + sb.records[offs[drop + 1]:]
    new_offs = offs[:drop + 1] + [o - cut for o in offs[drop + 2:]]
    return records, (ctypes.c_uint64 * len(new_offs))(*new_offs)


@pytest.fixture(scope="module")
def mixed(engine):
    sb = SyntheticBatch(100000, seed=b"mixed-full", mixed=True, threads=16, engine=engine)
    yield sb
    sb.close()


@pytest.mark.timeout(900)
def test_mixed_100000_accept_and_ledger(engine, mixed):
    v, cts = _verify(engine, mixed.records, mixed.offsets, mixed.n_txs, mixed.view)
    assert v.accept == 1, codes.NAMES.get(v.err_code)
    assert cts == _apply_on_gpu(engine, mixed)
    # the CPU apply path on every 50th account group (125 groups, 2 000 transactions): the groups are independent, six accounts
    # and eight ledger slots each
    lib = _lib()
    view = ctypes.cast(mixed.view, ctypes.POINTER(_RawView)).contents
    keys = ctypes.string_at(view.account_keys, 32 * view.n_accounts)
    assets = ctypes.string_at(view.balance_assets, 32 * view.n_balances)
    all_cts = ctypes.string_at(view.balance_cts, 64 * view.n_balances)
    for g in range(0, mixed.n_txs // 16, 50):
        g_keys, g_assets = keys[32 * 6 * g:32 * 6 * (g + 1)], assets[32 * 8 * g:32 * 8 * (g + 1)]
        cur = ctypes.create_string_buffer(all_cts[64 * 8 * g:64 * 8 * (g + 1)], 64 * 8)
        bal_acc = (ctypes.c_uint32 * 8)(0, 0, 1, 1, 2, 3, 4, 5)
        for i in range(16 * g, 16 * g + 16):
            rec = mixed.records[mixed.offsets[i]:mixed.offsets[i + 1]]
            assert lib.xhe_tx_apply_host(rec, len(rec), 6, g_keys, 8, bal_acc, g_assets, cur) == 0, i
        assert cts[64 * 8 * g:64 * 8 * (g + 1)] == cur.raw[:64 * 8], g


@pytest.mark.timeout(900)
def test_mixed_100000_against_the_oracle(engine, oracle, mixed):
    """All 100 000 mixed transactions through the CPU restatement (32 pieces along the account groups, 16 host threads): every
    piece accepted, and the ledger the pieces leave equals the GPU verification's byte for byte."""
    from helpers import oracle_verify_in_pieces
    v, cts = _verify(engine, mixed.records, mixed.offsets, mixed.n_txs, mixed.view)
    assert v.accept == 1, codes.NAMES.get(v.err_code)
    verdicts, merged = oracle_verify_in_pieces(oracle, mixed)
    assert all(pv == (1, codes.OK, -1) for pv in verdicts)
    assert merged == cts


@pytest.mark.timeout(900)
def test_mixed_100000_bad_signatures_and_chain_gap(engine, mixed):
    # a co-signature (transaction 14 of group 4 321: the first multisig entry's signature, behind its index byte)
    idx = 16 * 4321 + 14
    rec = bytearray(mixed.records)
    rec[mixed.offsets[idx + 1] - 64 - 2 * 65 + 1 + 20] ^= 0x10
    v, _ = _verify(engine, bytes(rec), mixed.offsets, mixed.n_txs, mixed.view)
    assert (v.accept, v.err_code, v.first_bad_tx) == (0, codes.ERR_SIGNATURE, idx)
    # the source's signature of a burn
    idx = 16 * 6000 + 12
    rec = bytearray(mixed.records)
    rec[mixed.offsets[idx + 1] - 40] ^= 0x10
    v, _ = _verify(engine, bytes(rec), mixed.offsets, mixed.n_txs, mixed.view)
    assert (v.accept, v.err_code, v.first_bad_tx) == (0, codes.ERR_SIGNATURE, idx)
    # without the first transfer of group 5's sender, the sender's later transactions prove against a balance the ledger never reaches
    records, offsets = _without(mixed, 16 * 5 + 2)
    v, _ = _verify(engine, records, offsets, mixed.n_txs - 1, mixed.view)
    assert (v.accept, v.err_code, v.first_bad_tx) == (0, codes.ERR_GENERIC_PROOF, -1)
    # without a multisig set-up, the account's co-signed transfer carries co-signatures its account is not configured for
    records, offsets = _without(mixed, 16 * 7 + 1)
    v, _ = _verify(engine, records, offsets, mixed.n_txs - 1, mixed.view)
    assert (v.accept, v.err_code, v.first_bad_tx) == (0, codes.ERR_FORMAT, 16 * 7 + 15 - 1)


@pytest.fixture(scope="module")
def share(engine):
    sb = SyntheticBatch(125000, n_transfers=1, n_senders=1024, seed=b"share-of-a-million", threads=16, engine=engine)
    yield sb
    sb.close()


@pytest.mark.timeout(900)
def test_share_125000_accept_and_ledger(engine, share):
    v, cts = _verify(engine, share.records, share.offsets, share.n_txs, share.view)
    assert v.accept == 1, codes.NAMES.get(v.err_code)
    assert cts == _apply_on_gpu(engine, share)
    lib = _lib()
    view = ctypes.cast(share.view, ctypes.POINTER(_RawView)).contents
    cur = ctypes.create_string_buffer(ctypes.string_at(view.balance_cts, 64 * view.n_balances), 64 * view.n_balances)
    keys = ctypes.string_at(view.account_keys, 32 * view.n_accounts)
    assets = ctypes.string_at(view.balance_assets, 32 * view.n_balances)
    bal_acc = ctypes.cast(view.balance_account, ctypes.POINTER(ctypes.c_uint32))
    for i in range(share.n_txs):                      # every 16th sender's chain with the CPU apply path
        if (i % 1024) % 16:
            continue
        rec = share.records[share.offsets[i]:share.offsets[i + 1]]
        assert lib.xhe_tx_apply_host(rec, len(rec), view.n_accounts, keys, view.n_balances, bal_acc, assets, cur) == 0, i
    for s in range(0, 1024, 16):
        assert cts[128 * s:128 * s + 128] == cur.raw[128 * s:128 * s + 128], s


@pytest.mark.timeout(900)
def test_share_125000_against_the_oracle(engine, oracle, share):
    """All 125 000 transactions through the CPU restatement, as above."""
    from helpers import oracle_verify_in_pieces
    v, cts = _verify(engine, share.records, share.offsets, share.n_txs, share.view)
    assert v.accept == 1, codes.NAMES.get(v.err_code)
    verdicts, merged = oracle_verify_in_pieces(oracle, share)
    assert all(pv == (1, codes.OK, -1) for pv in verdicts)
    assert merged == cts


@pytest.mark.timeout(900)
def test_share_125000_bad_signature_and_chain_gap(engine, share):
    idx = 101010
    rec = bytearray(share.records)
    rec[share.offsets[idx + 1] - 40] ^= 0x10
    v, _ = _verify(engine, bytes(rec), share.offsets, share.n_txs, share.view)
    assert (v.accept, v.err_code, v.first_bad_tx) == (0, codes.ERR_SIGNATURE, idx)
    records, offsets = _without(share, 5)
    v, _ = _verify(engine, records, offsets, share.n_txs - 1, share.view)
    assert (v.accept, v.err_code, v.first_bad_tx) == (0, codes.ERR_GENERIC_PROOF, -1)
