"""Sharded verification on the GPU.  (1) xhe_verify_batch_sharded with two contexts on device 0 against the single-context call:
verdict, failing index and every ledger byte, for accepted and rejected batches.  (2) The process-per-GPU driver
(sharding.verify_batch_sharded over torch.distributed) with the HIP verifier: two ranks sharing device 0, reducing through gloo
HOST tensors, on the mixed 256-transaction batch, against the single-context result."""
import ctypes
import multiprocessing as mp
import os
import socket

import pytest

from xelis_he_amd import codes
from xelis_he_amd.api import LedgerOut, SyntheticBatch, Verdict

pytestmark = pytest.mark.gpu


def _single(engine, sb, records, offsets=None, n=None):
    n_bal, n_acc = sb.view.contents.n_balances, sb.view.contents.n_accounts
    cts = (ctypes.c_uint8 * (64 * n_bal))()
    ms = (ctypes.c_int64 * n_acc)()
    out = LedgerOut(cts, (ctypes.c_uint64 * n_acc)(), ms)
    v = Verdict()
    engine._check(engine.lib.xhe_verify_batch(engine.ctx, records, offsets or sb.offsets, sb.n_txs if n is None else n, sb.view, ctypes.byref(out), ctypes.byref(v), b"\x03" * 32))
    return (v.accept, v.err_code, v.first_bad_tx), bytes(cts), list(ms)


def _sharded(engines, sb, records, offsets=None, n=None):
    lib = engines[0].lib
    lib.xhe_verify_batch_sharded.restype = ctypes.c_int32
    n_bal, n_acc = sb.view.contents.n_balances, sb.view.contents.n_accounts
    cts = (ctypes.c_uint8 * (64 * n_bal))()
    ms = (ctypes.c_int64 * n_acc)()
    out = LedgerOut(cts, (ctypes.c_uint64 * n_acc)(), ms)
    v = Verdict()
    ctxs = (ctypes.c_void_p * len(engines))(*[e.ctx for e in engines])
    rc = lib.xhe_verify_batch_sharded(ctxs, len(engines), records, offsets or sb.offsets, ctypes.c_uint64(sb.n_txs if n is None else n), sb.view, ctypes.byref(out),
                                      ctypes.byref(v), b"\x04" * 32)
    assert rc == 0, rc
    return (v.accept, v.err_code, v.first_bad_tx), bytes(cts), list(ms)


def test_two_contexts_equal_one():
    import xelis_he_amd
    e1, e2, e3 = xelis_he_amd.Engine(0), xelis_he_amd.Engine(0), xelis_he_amd.Engine(0)
    try:
        for sb in (SyntheticBatch(256, seed=b"shard-mixed", mixed=True), SyntheticBatch(600, n_transfers=2, n_senders=11, seed=b"shard-chains", engine=e1)):
            want = _single(e1, sb, sb.records)
            assert want[0] == (1, codes.OK, -1)
            assert _sharded([e1, e2], sb, sb.records) == want
            assert _sharded([e1, e2, e3], sb, sb.records) == want
            offs = list(sb.offsets)
            # a broken signature, a broken sigma-proof scalar (GenericProof), a broken range proof that is re-signed cannot be made
            # here, so a flipped inner-product scalar inside the signed bytes: Signature; and a cut record: Malformed
            for tx, back in [(5, 10), (sb.n_txs - 1, 40), (sb.n_txs // 2, 64 + 100)]:
                bad = bytearray(sb.records)
                bad[offs[tx + 1] - back] ^= 1
                want = _single(e1, sb, bytes(bad))
                got = _sharded([e1, e2], sb, bytes(bad))
                assert got[0] == want[0] and want[0][0] == 0 and want[0][2] == tx, (tx, back, got[0], want[0])
            # a chain gap: the aggregate sigma check fails on one shard only
            drop = 3
            cut = offs[drop + 1] - offs[drop]
            records = sb.records[:offs[drop]] + sb.records[offs[drop + 1]:]
            new_offs = (ctypes.c_uint64 * sb.n_txs)(*(offs[:drop + 1] + [o - cut for o in offs[drop + 2:]]))
            want = _single(e1, sb, records, new_offs, sb.n_txs - 1)
            got = _sharded([e1, e2], sb, records, new_offs, sb.n_txs - 1)
            assert got[0] == want[0] and want[0][0] == 0
            # malformed: the last record cut short
            short = (ctypes.c_uint64 * (sb.n_txs + 1))(*(offs[:-1] + [offs[-1] - 1]))
            want = _single(e1, sb, sb.records, short)
            got = _sharded([e1, e2], sb, sb.records, short)
            assert got[0] == want[0] == (0, codes.ERR_MALFORMED, sb.n_txs - 1)
            sb.close()
    finally:
        for e in (e1, e2, e3):
            e.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, records, ledger_blob, queue):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import pickle
    import torch.distributed as dist
    import xelis_he_amd
    from xelis_he_amd import sharding
    from xelis_he_amd.api import Transaction
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    engine = xelis_he_amd.Engine(0)                # both ranks share the one GPU of the box; the reduction goes through host memory
    try:
        ledger = pickle.loads(ledger_blob)
        txs = [Transaction(r) for r in records]
        res = sharding.verify_batch_sharded(sharding.gpu_verify_fn(engine, seed="rank-%d" % rank), txs, ledger, dist, apply_fn=sharding.gpu_apply_fn(engine))
        cts = {k.hex(): {a.hex(): c.hex() for a, c in acc.balances.items()} for k, acc in ledger.accounts.items()}
        ms = {k.hex(): (len(v[0]), v[1]) for k, v in ledger.multisig_accounts.items()}
        queue.put((rank, res, cts, ms, [len(s) for s in sharding.partition(records, world)]))
    finally:
        engine.close()
        dist.destroy_process_group()


class _PublicKeyOnly:
    """Stands in for a Keypair in a Ledger mirror built from a flat ledger (no secrets are known there)."""

    def __init__(self, pubkey):
        self.pubkey = pubkey


def _mixed_ledger_and_txs(n_groups, seed):
    """The mixed synthetic batch as Python objects: a Ledger mirror of its flat ledger and its records."""
    return _ledger_and_txs(SyntheticBatch(16 * n_groups, seed=seed, mixed=True))


def _ledger_and_txs(sb):
    from xelis_he_amd.api import Account, Ledger, Transaction
    from test_gpu_fullsize import _RawView
    view = ctypes.cast(sb.view, ctypes.POINTER(_RawView)).contents      # the same struct with untyped pointers (c_char_p would stop at a NUL)
    keys = ctypes.string_at(view.account_keys, 32 * view.n_accounts)
    assets = ctypes.string_at(view.balance_assets, 32 * view.n_balances)
    cts = ctypes.string_at(view.balance_cts, 64 * view.n_balances)
    bal_acc = ctypes.cast(view.balance_account, ctypes.POINTER(ctypes.c_uint32))
    accounts = []
    for a in range(view.n_accounts):
        acc = Account.__new__(Account)
        acc.keypair = _PublicKeyOnly(keys[32 * a:32 * a + 32])
        acc.balances, acc.nonce = {}, 0
        accounts.append(acc)
    for i in range(view.n_balances):
        accounts[bal_acc[i]].balances[assets[32 * i:32 * i + 32]] = cts[64 * i:64 * i + 64]
    ledger = Ledger(accounts)
    recs = [sb.records[sb.offsets[i]:sb.offsets[i + 1]] for i in range(sb.n_txs)]
    return sb, ledger, [Transaction(r) for r in recs]


@pytest.mark.timeout(600)
def test_two_ranks_with_the_gpu_verifier():
    import pickle
    import xelis_he_amd
    from xelis_he_amd.api import Transaction
    sb, ledger, txs = _mixed_ledger_and_txs(16, b"ranks-mixed")
    engine = xelis_he_amd.Engine(0)
    try:
        single = ledger.clone()
        Transaction.verify_batch(engine, txs, single, seed="single")
        want_cts = {k.hex(): {a.hex(): c.hex() for a, c in acc.balances.items()} for k, acc in single.accounts.items()}
        want_ms = {k.hex(): (len(v[0]), v[1]) for k, v in single.multisig_accounts.items()}
        bad = [t.record for t in txs]
        bad[77] = bad[77][:-20] + bytes([bad[77][-20] ^ 1]) + bad[77][-19:]
        try:
            Transaction.verify_batch(engine, [Transaction(r) for r in bad], ledger.clone(), seed="single-bad")
            single_bad = (True, codes.OK, -1)
        except Exception as e:          # VerificationError
            single_bad = (False, e.code, e.first_bad_tx)
    finally:
        engine.close()
    assert single_bad == (False, codes.ERR_SIGNATURE, 77)
    _check_two_ranks(ledger, [([t.record for t in txs], (True, codes.OK, -1)), (bad, single_bad)], want_cts, want_ms, groups=True)
    sb.close()


def _check_two_ranks(ledger, cases, want_cts, want_ms, groups):
    import pickle
    ctx = mp.get_context("spawn")
    for records, expect in cases:
        q, port = ctx.Queue(), _free_port()
        procs = [ctx.Process(target=_rank, args=(r, 2, port, records, pickle.dumps(ledger), q)) for r in range(2)]
        for p in procs:
            p.start()
        out = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        for rank, res, cts, ms, sizes in out:
            assert res == expect, (rank, res)
            if groups:
                assert all(s > 0 for s in sizes)
            if expect[0]:
                assert cts == want_cts and ms == want_ms, rank


def _plan(lib, sb, n_shards):
    lib.xhe_plan_shards.restype = ctypes.c_int32
    shard_of, ranges, bad = (ctypes.c_uint32 * sb.n_txs)(), ctypes.c_int32(), ctypes.c_int64()
    assert lib.xhe_plan_shards(sb.records, sb.offsets, ctypes.c_uint64(sb.n_txs), n_shards, shard_of, ctypes.byref(ranges), ctypes.byref(bad)) == 0
    return list(shard_of), ranges.value


@pytest.mark.timeout(900)
def test_single_sender_chain_over_contexts(oracle):
    """One sender for all 4 096 transactions (the reference bench's shape, /root/reference/benches/tx.rs:129-189): one account
    group, so xhe_verify_batch_sharded cuts consecutive ranges, runs the ledger through them and verifies every range from the
    state the ranges before it leave.  Two and three contexts against the single-context call and the oracle: verdict, failing
    index, every ledger byte; for the valid chain, a bad signature in a later range, a missing transaction in the first range
    and a transfer ciphertext that does not decode (the ledger chain breaks there)."""
    import xelis_he_amd
    e1, e2, e3 = xelis_he_amd.Engine(0), xelis_he_amd.Engine(0), xelis_he_amd.Engine(0)
    try:
        sb = SyntheticBatch(4096, n_transfers=1, n_senders=1, seed=b"one-sender", engine=e1)
        shard_of, ranges = _plan(e1.lib, sb, 3)
        assert ranges == 1 and shard_of == sorted(shard_of) and [shard_of.count(s) for s in range(3)] == [1366, 1365, 1365]
        assert _plan(e1.lib, SyntheticBatch(512, n_transfers=1, n_senders=16, seed=b"groups"), 2)[1] == 0
        want = _single(e1, sb, sb.records)
        assert want[0] == (1, codes.OK, -1)
        n_bal, n_acc = sb.view.contents.n_balances, sb.view.contents.n_accounts
        ref_cts = (ctypes.c_uint8 * (64 * n_bal))()
        ref_out, rv = LedgerOut(ref_cts, (ctypes.c_uint64 * n_acc)(), (ctypes.c_int64 * n_acc)()), Verdict()
        oracle.xo_verify_batch(sb.records, sb.offsets, ctypes.c_uint64(sb.n_txs), sb.view, ctypes.byref(ref_out), ctypes.byref(rv), b"\x02" * 32)
        assert (rv.accept, rv.err_code, rv.first_bad_tx) == want[0] and bytes(ref_cts) == want[1]
        assert _sharded([e1, e2], sb, sb.records) == want
        assert _sharded([e1, e2, e3], sb, sb.records) == want
        offs = list(sb.offsets)
        bad = bytearray(sb.records)
        bad[offs[3001] - 40] ^= 1                         # transaction 3 000's signature: the last range of three, the second of two
        undecodable = bytearray(sb.records)
        c_at = offs[2000] + 16 + 4 + 49 + 64              # the amount commitment of transaction 2 000's transfer
        undecodable[c_at:c_at + 32] = bytes([0xff]) * 32
        for records, expect in ((bytes(bad), (0, codes.ERR_SIGNATURE, 3000)), (bytes(undecodable), (0, codes.ERR_DECOMPRESSION, 2000))):
            assert _single(e1, sb, records)[0] == expect
            assert _sharded([e1, e2], sb, records)[0] == expect
            assert _sharded([e1, e2, e3], sb, records)[0] == expect
        drop = 100
        cut = offs[drop + 1] - offs[drop]
        records = sb.records[:offs[drop]] + sb.records[offs[drop + 1]:]
        new_offs = (ctypes.c_uint64 * sb.n_txs)(*(offs[:drop + 1] + [o - cut for o in offs[drop + 2:]]))
        want_gap = _single(e1, sb, records, new_offs, sb.n_txs - 1)
        assert want_gap[0] == (0, codes.ERR_GENERIC_PROOF, -1)
        assert _sharded([e1, e2], sb, records, new_offs, sb.n_txs - 1)[0] == want_gap[0]
        assert _sharded([e1, e2, e3], sb, records, new_offs, sb.n_txs - 1)[0] == want_gap[0]
        sb.close()
    finally:
        for e in (e1, e2, e3):
            e.close()


@pytest.mark.timeout(600)
def test_two_ranks_single_sender_chain_with_the_gpu_verifier():
    """The process-per-GPU driver on the single-sender chain: the ledger passes from rank 0 to rank 1 (a broadcast of the
    ciphertexts, host tensors over gloo here), each rank verifies its range on the GPU."""
    import xelis_he_amd
    from xelis_he_amd.api import Transaction
    engine = xelis_he_amd.Engine(0)
    try:
        sb = SyntheticBatch(512, n_transfers=2, n_senders=1, seed=b"ranks-one-sender", engine=engine)
        _, ledger, txs = _ledger_and_txs(sb)
        single = ledger.clone()
        Transaction.verify_batch(engine, txs, single, seed="single")
        want_cts = {k.hex(): {a.hex(): c.hex() for a, c in acc.balances.items()} for k, acc in single.accounts.items()}
    finally:
        engine.close()
    bad = [t.record for t in txs]
    bad[400] = bad[400][:-20] + bytes([bad[400][-20] ^ 1]) + bad[400][-19:]
    _check_two_ranks(ledger, [([t.record for t in txs], (True, codes.OK, -1)), (bad, (False, codes.ERR_SIGNATURE, 400))], want_cts, {}, groups=False)
    sb.close()
