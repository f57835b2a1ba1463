"""The multi-rank path on CPU: two processes, torch.distributed over gloo (host tensors), sharding by
account group and the MIN all-reduce of verdict keys.  The per-shard verifier here is the CPU oracle
(test infrastructure); on GPUs the same driver runs xhe_verify_batch per rank over RCCL."""
import ctypes
import multiprocessing as mp
import os
import socket

import pytest

from helpers import oracle_verify_batch
from xelis_he_amd import codes, sharding
from xelis_he_amd.api import NATIVE, Account, Ledger, Transaction, TransactionBuilder, TransferBuilder


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_verify_fn():
    import helpers
    from xelis_he_amd.api import Verdict
    oracle = helpers.load_oracle()

    def run(shard_txs, f, out):
        blob = b"".join(t.record for t in shard_txs)
        offs = [0]
        for t in shard_txs:
            offs.append(offs[-1] + len(t.record))
        offsets = (ctypes.c_uint64 * len(offs))(*offs)
        v = Verdict()
        oracle.xo_verify_batch(blob, offsets, ctypes.c_uint64(len(shard_txs)), ctypes.byref(f["view"]), ctypes.byref(out), ctypes.byref(v), bytes([7]) * 32)
        return bool(v.accept), int(v.err_code), int(v.first_bad_tx)
    return run


def _oracle_apply_fn():
    """Transaction::apply_without_verify over a range with the oracle: the ledger's ciphertexts after the range, or None."""
    import helpers
    from xelis_he_amd.api import LedgerOut
    oracle = helpers.load_oracle()
    oracle.xo_apply_batch.restype = ctypes.c_int32

    def run(range_txs, ledger, start_cts):
        start = ledger.clone()
        f = start.flatten()
        for i, (acc_i, asset) in enumerate(f["slots"]):
            start.accounts[f["keys"][acc_i]].balances[asset] = start_cts[64 * i:64 * i + 64]
        f = start.flatten()
        blob = b"".join(t.record for t in range_txs)
        offs = [0]
        for t in range_txs:
            offs.append(offs[-1] + len(t.record))
        offsets = (ctypes.c_uint64 * len(offs))(*offs)
        cts = (ctypes.c_uint8 * max(64 * len(f["slots"]), 1))()
        out = LedgerOut(cts, (ctypes.c_uint64 * max(len(f["keys"]), 1))(), (ctypes.c_int64 * max(len(f["keys"]), 1))())
        rc = oracle.xo_apply_batch(blob, offsets, ctypes.c_uint64(len(range_txs)), ctypes.byref(f["view"]), ctypes.byref(out))
        return bytes(cts)[:64 * len(f["slots"])] if rc == 0 else None
    return run


def _worker(rank, world, port, records, ledger, queue):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        txs = [Transaction(r) for r in records]
        res = sharding.verify_batch_sharded(_oracle_verify_fn(), txs, ledger, dist, apply_fn=_oracle_apply_fn())
        cts = {k.hex(): {a.hex(): c.hex() for a, c in acc.balances.items()} for k, acc in ledger.accounts.items()}
        queue.put((rank, res, cts, sharding.partition(records, world)))
    finally:
        dist.destroy_process_group()


def _run_two_ranks(records, ledger):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, records, ledger, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out)


def _chains(n_chains, length, seed):
    """n_chains independent sender->receiver chains, interleaved in the batch."""
    senders = [Account([(NATIVE, 1000)], "%s-s%d" % (seed, i)) for i in range(n_chains)]
    receivers = [Account([(NATIVE, 0)], "%s-r%d" % (seed, i)) for i in range(n_chains)]
    ledger = Ledger(senders + receivers)
    prover, txs, bal = ledger.clone(), [], [1000] * n_chains
    for k in range(length):
        for i in range(n_chains):
            tx = TransactionBuilder(1, senders[i].pubkey, ("transfers", [TransferBuilder(receivers[i].pubkey, 1 + i)]), 2, 0).build(
                {NATIVE: bal[i]}, senders[i].keypair, "%s-%d-%d" % (seed, i, k), prover.get_account(senders[i].pubkey))
            tx.apply_without_verify(prover)
            bal[i] -= 3 + i
            txs.append(tx)
    return ledger, txs


def test_partition_keeps_account_groups_together():
    ledger, txs = _chains(3, 2, "part")
    recs = [t.record for t in txs]
    shards = sharding.partition(recs, 2)
    assert sorted(shards[0] + shards[1]) == list(range(len(txs)))
    owner = {}
    for r, s in enumerate(shards):
        assert s == sorted(s)
        for i in s:
            src, dests = sharding.tx_accounts(recs[i])
            for k in [src] + dests:
                assert owner.setdefault(k, r) == r
    assert sharding.partition(recs, 1) == [list(range(len(txs)))]
    # keys: accept is the maximum, per-transaction errors order by index, aggregates come after them
    k_sig5 = sharding.verdict_key(False, codes.ERR_SIGNATURE, 1, [3, 5])
    k_dec2 = sharding.verdict_key(False, codes.ERR_DECOMPRESSION, 0, [2])
    k_gen = sharding.verdict_key(False, codes.ERR_GENERIC_PROOF, -1, [0, 1])
    k_rp = sharding.verdict_key(False, codes.ERR_RANGE_PROOF, -1, [4])
    assert k_dec2 < k_sig5 < k_gen < k_rp < sharding.ACCEPT_KEY
    assert sharding.decode_key(k_sig5) == (False, codes.ERR_SIGNATURE, 5)
    assert sharding.decode_key(k_gen) == (False, codes.ERR_GENERIC_PROOF, -1)
    assert sharding.decode_key(sharding.ACCEPT_KEY) == (True, codes.OK, -1)


@pytest.mark.timeout(300)
def test_two_ranks_accept_and_merge(oracle):
    ledger, txs = _chains(2, 2, "acc")
    single = ledger.clone()
    ok, code, _ = oracle_verify_batch(oracle, txs, single)
    assert ok, codes.NAMES[code]
    out = _run_two_ranks([t.record for t in txs], ledger)
    want = {k.hex(): {a.hex(): c.hex() for a, c in acc.balances.items()} for k, acc in single.accounts.items()}
    for rank, res, cts, shards in out:
        assert res == (True, codes.OK, -1)
        assert cts == want                      # every rank ends with the single-process ledger, byte for byte
        assert all(len(s) > 0 for s in shards)


@pytest.mark.timeout(300)
def test_two_ranks_first_error_wins(oracle):
    ledger, txs = _chains(2, 2, "rej")
    recs = [t.record for t in txs]
    # transaction 3 gets a zero signature
    bad = list(recs)
    bad[3] = bad[3][:-64] + bytes(64)
    single_ok, single_code, single_idx = oracle_verify_batch(oracle, [Transaction(r) for r in bad], ledger.clone())
    assert (single_ok, single_code, single_idx) == (False, codes.ERR_SIGNATURE, 3)
    for rank, res, _, _ in _run_two_ranks(bad, ledger):
        assert res == (False, codes.ERR_SIGNATURE, 3)
    # an aggregate failure on one rank (chain gap) and a per-transaction failure on the other: the latter is reported
    # batch: tx0 = chain1 link0 (valid), tx1 = chain0 link1 with a zero signature, tx2 = chain1 link1 (valid)
    gap = [recs[1], recs[2][:-64] + bytes(64), recs[3]]
    single = oracle_verify_batch(oracle, [Transaction(r) for r in gap], ledger.clone())
    for rank, res, _, _ in _run_two_ranks(gap, ledger):
        assert res == single == (False, codes.ERR_SIGNATURE, 1)
    # chain gap only: GenericProof with no index
    gap2 = [recs[1], recs[2], recs[3]]          # chain0's first link is missing
    single = oracle_verify_batch(oracle, [Transaction(r) for r in gap2], ledger.clone())
    assert single == (False, codes.ERR_GENERIC_PROOF, -1)
    for rank, res, _, _ in _run_two_ranks(gap2, ledger):
        assert res == single


@pytest.mark.timeout(300)
def test_two_ranks_single_sender_chain(oracle):
    """One sender for every transaction (the reference bench's shape, benches/tx.rs:129-189): a single account group, so the
    batch is cut into consecutive ranges and the ledger passes from rank 0 to rank 1.  Verdict and ledger equal the
    single-process run for a valid chain, a bad signature in the second range, and a gap in the first range."""
    ledger, txs = _chains(1, 6, "one")
    recs = [t.record for t in txs]
    shards, ranges = sharding.plan(recs, 2)
    assert ranges and shards == [[0, 1, 2], [3, 4, 5]]
    assert sharding.plan([t.record for t in _chains(2, 3, "two")[1]], 2)[1] is False          # two equal groups: no ranges
    single = ledger.clone()
    assert oracle_verify_batch(oracle, txs, single) == (True, codes.OK, -1)
    want = {k.hex(): {a.hex(): c.hex() for a, c in acc.balances.items()} for k, acc in single.accounts.items()}
    for rank, res, cts, _ in _run_two_ranks(recs, ledger):
        assert res == (True, codes.OK, -1)
        assert cts == want
    bad = list(recs)
    bad[4] = bad[4][:-64] + bytes(64)
    assert oracle_verify_batch(oracle, [Transaction(r) for r in bad], ledger.clone()) == (False, codes.ERR_SIGNATURE, 4)
    for rank, res, _, _ in _run_two_ranks(bad, ledger):
        assert res == (False, codes.ERR_SIGNATURE, 4)
    gap = recs[:1] + recs[2:]                      # the second link is missing: every later proof is against a balance never reached
    single = oracle_verify_batch(oracle, [Transaction(r) for r in gap], ledger.clone())
    assert single == (False, codes.ERR_GENERIC_PROOF, -1)
    for rank, res, _, _ in _run_two_ranks(gap, ledger):
        assert res == single
