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
        res = sharding.verify_batch_sharded(sharding.gpu_verify_fn(engine, seed="rank-%d" % rank), txs, ledger, dist)
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
    from xelis_he_amd.api import Account, Ledger, Transaction
    sb = SyntheticBatch(16 * n_groups, seed=seed, mixed=True)
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
    ctx = mp.get_context("spawn")
    for records, expect in (([t.record for t in txs], (True, codes.OK, -1)), (bad, single_bad)):
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
            assert all(s > 0 for s in sizes)
            if expect[0]:
                assert cts == want_cts and ms == want_ms, rank
    sb.close()
