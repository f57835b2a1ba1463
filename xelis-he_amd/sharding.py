"""Multi-GPU Transaction::verify_batch: one process per GPU, transactions sharded by account group.

A transaction's checks read the sender's balance as every earlier transaction touching that
(account, asset) left it (/root/reference/src/tx/verify.rs:301-341), so the unit of sharding is a
connected component of the graph {source, destination} accounts, not a transaction.  Components are
bin-packed onto ranks by multiscalar term count; every rank verifies its transactions, in batch order,
against the same ledger snapshot.  The only exchange is one all-reduce (MIN) of an 8-byte key that is
both the AND of the ranks' accept bits and the reference's error precedence: per-transaction errors in
batch order first (verify.rs:492-498), then GenericProof, then RangeProof (verify.rs:500-514).
With torch.distributed backend "nccl" the all-reduce is RCCL over xGMI; with "gloo" the tensors are
host tensors.

When the account groups do not balance (one sender for the whole batch, /root/reference/benches/tx.rs:129-189) the library
plans consecutive ranges instead (xhe_plan_shards).  Range s must start from the ledger the ranges before it leave: the ranks
pass the ledger's ciphertexts along -- rank s applies its range (Transaction::apply_without_verify, verify.rs:545-619, the
ledger arithmetic alone) to the state it received and broadcasts the result -- and every rank verifies its range against the
state it started from.  That broadcast chain is the one exchange of ledger data in the path.
"""
import ctypes

from . import codes
from ._capi import XheError, load_library

ACCEPT_KEY = (1 << 63) - 1


def _lib():
    lib = load_library()
    if not getattr(lib, "_shard_ready", False):
        lib.xhe_partition_batch.restype = ctypes.c_int32
        lib.xhe_partition_batch.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32),
                                            ctypes.POINTER(ctypes.c_int64)]
        lib.xhe_plan_shards.restype = ctypes.c_int32
        lib.xhe_plan_shards.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32),
                                        ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)]
        lib.xhe_slice_shard.restype = ctypes.c_int32
        lib.xhe_slice_shard.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        lib.xhe_verdict_key.restype = ctypes.c_uint64
        lib.xhe_verdict_key.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64]
        lib.xhe_verdict_from_key.argtypes = [ctypes.c_uint64, ctypes.c_void_p]
        lib._shard_ready = True
    return lib


class MalformedRecord(Exception):
    """A record of the batch does not parse: the verdict is XHE_ERR_MALFORMED at .index, whatever else the batch holds."""

    def __init__(self, index):
        super().__init__("record %d is malformed" % index)
        self.index = index


def tx_accounts(record: bytes):
    """(source, [destinations]) of a flat transaction record (include/xhe.h); the record must have parsed (partition checks)."""
    ttype, n_items = record[0], int.from_bytes(record[2:4], "little")
    pre = 16 + (4 * n_items if ttype == codes.TX_TRANSFERS else 0)
    source = record[pre + 1:pre + 33]
    dests = []
    if ttype == codes.TX_TRANSFERS:
        off = pre + 49
        for i in range(n_items):
            extra = int.from_bytes(record[16 + 4 * i:20 + 4 * i], "little")
            dests.append(record[off + 32:off + 64])
            off += 320 + extra
    return source, dests


def partition(records, world_size):
    """Transaction indices per rank (xhe_partition_batch): connected components of the account graph, heaviest first onto the
    lightest rank; indices ascending inside a rank (batch order is preserved).  Every record is parsed by the library first:
    a malformed one raises MalformedRecord with the smallest such index."""
    n = len(records)
    blob = b"".join(records)
    offs = [0]
    for r in records:
        offs.append(offs[-1] + len(r))
    offsets = (ctypes.c_uint64 * (n + 1))(*offs)
    shard_of = (ctypes.c_uint32 * max(n, 1))()
    bad = ctypes.c_int64(-1)
    rc = _lib().xhe_partition_batch(blob, offsets, n, world_size, shard_of, ctypes.byref(bad))
    if rc == 1:
        raise MalformedRecord(int(bad.value))
    if rc != 0:
        raise XheError("xhe_partition_batch failed: %d" % rc)
    shards = [[] for _ in range(world_size)]
    for i in range(n):
        shards[shard_of[i]].append(i)
    return shards


def plan(records, world_size):
    """(transaction indices per rank, consecutive_ranges) as xhe_verify_batch_sharded itself cuts a batch (xhe_plan_shards):
    account groups, or consecutive ranges of equal weight when the groups do not balance."""
    n = len(records)
    blob = b"".join(records)
    offs = [0]
    for r in records:
        offs.append(offs[-1] + len(r))
    offsets = (ctypes.c_uint64 * (n + 1))(*offs)
    shard_of = (ctypes.c_uint32 * max(n, 1))()
    bad, ranges = ctypes.c_int64(-1), ctypes.c_int32(0)
    rc = _lib().xhe_plan_shards(blob, offsets, n, world_size, shard_of, ctypes.byref(ranges), ctypes.byref(bad))
    if rc == 1:
        raise MalformedRecord(int(bad.value))
    if rc != 0:
        raise XheError("xhe_plan_shards failed: %d" % rc)
    shards = [[] for _ in range(world_size)]
    for i in range(n):
        shards[shard_of[i]].append(i)
    return shards, bool(ranges.value)


def cut_for_ranks(batch, world_size):
    """A synthetic batch (api.SyntheticBatch) cut into one piece per rank inside the library (xhe_plan_shards + xhe_slice_shard): per
    rank a dict of numpy arrays -- records (uint8), offsets (uint64), index (uint64: the transactions' positions in the whole batch),
    and the ledger's keys, assets, cts (uint8) and balance_account (uint32).  Every rank gets the whole ledger arrays of the
    synthetic batch (a few kilobytes per account group); the records, which are the bulk, are cut."""
    import numpy as np
    lib = _lib()
    n = batch.n_txs
    shard_of = (ctypes.c_uint32 * max(n, 1))()
    bad, ranges = ctypes.c_int64(-1), ctypes.c_int32(0)
    rc = lib.xhe_plan_shards(batch.records, batch.offsets, n, world_size, shard_of, ctypes.byref(ranges), ctypes.byref(bad))
    if rc != 0:
        raise XheError("xhe_plan_shards failed: %d (first malformed record %d)" % (rc, bad.value))
    if ranges.value:
        raise XheError("the batch's account groups do not balance over %d ranks: consecutive ranges need the ledger passed along (verify_batch_sharded)" % world_size)

    class RawView(ctypes.Structure):
        _fields_ = [("n_accounts", ctypes.c_uint64), ("account_keys", ctypes.c_void_p), ("account_nonces", ctypes.c_void_p), ("n_balances", ctypes.c_uint64),
                    ("balance_account", ctypes.c_void_p), ("balance_assets", ctypes.c_void_p), ("balance_cts", ctypes.c_void_p)]
    rv = ctypes.cast(batch.view, ctypes.POINTER(RawView)).contents
    ledger = {"keys": np.frombuffer(ctypes.string_at(rv.account_keys, 32 * rv.n_accounts), dtype=np.uint8),
              "assets": np.frombuffer(ctypes.string_at(rv.balance_assets, 32 * rv.n_balances), dtype=np.uint8),
              "cts": np.frombuffer(ctypes.string_at(rv.balance_cts, 64 * rv.n_balances), dtype=np.uint8),
              "balance_account": np.frombuffer(ctypes.string_at(rv.balance_account, 4 * rv.n_balances), dtype=np.uint32)}
    pieces = []
    for s in range(world_size):
        n_local, n_bytes = ctypes.c_uint64(), ctypes.c_uint64()
        lib.xhe_slice_shard(batch.records, batch.offsets, n, shard_of, s, None, None, None, ctypes.byref(n_local), ctypes.byref(n_bytes))
        recs = np.zeros(max(n_bytes.value, 1), dtype=np.uint8)
        offs = np.zeros(n_local.value + 1, dtype=np.uint64)
        index = np.zeros(max(n_local.value, 1), dtype=np.uint64)
        rc = lib.xhe_slice_shard(batch.records, batch.offsets, n, shard_of, s, recs.ctypes.data, offs.ctypes.data, index.ctypes.data, ctypes.byref(n_local), ctypes.byref(n_bytes))
        if rc != 0:
            raise XheError("xhe_slice_shard failed: %d" % rc)
        piece = {"records": recs[:n_bytes.value], "offsets": offs, "index": index[:n_local.value]}
        piece.update(ledger)
        pieces.append(piece)
    return pieces


def verdict_key(accept, err_code, first_bad_local, shard):
    """The 63-bit key a rank contributes to the MIN all-reduce (xhe_verdict_key): per-transaction errors order by their index in
    the whole batch, then GenericProof, then RangeProof."""
    from .api import Verdict
    v = Verdict(1 if accept else 0, int(err_code), int(first_bad_local))
    g = (ctypes.c_uint64 * max(len(shard), 1))(*shard)
    return int(_lib().xhe_verdict_key(ctypes.byref(v), g, len(shard)))


def decode_key(key):
    """(accept, err_code, first_bad_tx) of a reduced key (xhe_verdict_from_key)."""
    from .api import Verdict
    v = Verdict()
    _lib().xhe_verdict_from_key(ctypes.c_uint64(key), ctypes.byref(v))
    return bool(v.accept), int(v.err_code), int(v.first_bad_tx)


def all_reduce_min(value, dist, device=None):
    import torch
    t = torch.tensor([value], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.cpu()[0])


def merge_ledger_bytes(own_cts, owned_mask, dist, device=None):
    """Resulting ciphertexts of every ledger slot: each slot is written by at most one rank (its account's component lives
    there), so the ranks exchange only the slots they own -- (slot index, 64 bytes) per touched slot, not the whole ledger.
    Returns (bytes of all slots with the owned ones filled in, zero elsewhere; per slot the number of ranks that own it)."""
    n = len(owned_mask)
    mine = [i for i, m in enumerate(owned_mask) if m]
    payload = (mine, b"".join(own_cts[64 * i:64 * i + 64] for i in mine))
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, payload)
    merged, count = bytearray(64 * n), [0] * n
    for idx, blob in gathered:
        for k, i in enumerate(idx):
            merged[64 * i:64 * i + 64] = blob[64 * k:64 * k + 64]
            count[i] += 1
    return bytes(merged), count


def _verify_ranges(verify_fn, apply_fn, txs, ledger, shards, dist, device):
    """Consecutive ranges: the ledger's ciphertexts pass from rank to rank (one broadcast per rank, in order), every rank verifies
    its range against the state it started from.  A rank whose apply fails says so with its broadcast: the ranks behind it do
    not verify (its own verification reports the error, which comes first)."""
    import torch
    from .api import LedgerOut
    rank, world = dist.get_rank(), dist.get_world_size()
    f = ledger.flatten()
    n_slots = len(f["slots"])
    cur = bytearray(f["balance_cts"])
    my_start, chain_ok = None, True
    for s in range(world):
        if s == rank and chain_ok:             # the chain is intact up to here: this rank has a state to start from
            my_start = bytes(cur)
        payload = torch.zeros(64 * n_slots + 1, dtype=torch.uint8)
        if s == rank and chain_ok:
            after = apply_fn([txs[i] for i in shards[s]], ledger, bytes(cur))
            if after is None:
                payload[0] = 1
            else:
                payload[1:] = torch.frombuffer(bytearray(after), dtype=torch.uint8)
        elif s == rank:
            payload[0] = 1
        if device is not None:
            payload = payload.to(device)
        dist.broadcast(payload, src=s)
        payload = payload.cpu()
        if int(payload[0]) != 0:
            chain_ok = False
        elif chain_ok:
            cur = bytearray(payload[1:].numpy().tobytes())
    # this rank's range against the state it started from (ranks behind a broken chain have no such state: they contribute nothing;
    # the rank the chain broke at verifies and names the error)
    key = ACCEPT_KEY
    if my_start is not None:
        start = ledger.clone()
        for i, (acc_i, asset) in enumerate(f["slots"]):
            start.accounts[f["keys"][acc_i]].balances[asset] = my_start[64 * i:64 * i + 64]
        f2 = start.flatten()
        out = LedgerOut((ctypes.c_uint8 * max(64 * n_slots, 1))(), (ctypes.c_uint64 * max(len(f2["keys"]), 1))(), (ctypes.c_int64 * max(len(f2["keys"]), 1))())
        accept, code, first_bad = verify_fn([txs[i] for i in shards[rank]], f2, out)
        key = verdict_key(accept, code, first_bad, shards[rank])
    key = all_reduce_min(key, dist, device)
    ok, code, first = decode_key(key)
    if not ok:
        return False, code, first
    if not chain_ok:                           # cannot happen with an accepting verdict: an apply failure is a verification failure
        return False, codes.ERR_STATE, -1
    ledger.absorb(f, bytes(cur), [int(x) for x in f["nonces"]][:len(f["keys"])], [-1] * len(f["keys"]), txs)
    return True, codes.OK, -1


def verify_batch_sharded(verify_fn, txs, ledger, dist, device=None, apply_fn=None):
    """Transaction::verify_batch over all ranks of `dist` (an initialised torch.distributed).

    verify_fn(shard_txs, flat_ledger, out: LedgerOut) -> (accept, err_code, first_bad_local) verifies one
    shard against the flattened ledger and, on accept, fills `out`.  apply_fn(range_txs, ledger, start_cts) -> the ledger's
    ciphertexts (bytes, slot order of ledger.flatten()) after the range was applied to start_cts, or None when the apply fails;
    with it, batches whose account groups do not balance are cut into consecutive ranges (see the module's head).  Returns the
    batch's (accept, err_code, first_bad_tx), identical on every rank; on accept `ledger` holds the merged state.
    """
    from .api import LedgerOut
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        if apply_fn is not None:
            shards, ranges = plan([t.record for t in txs], world)
            if ranges:
                return _verify_ranges(verify_fn, apply_fn, txs, ledger, shards, dist, device)
        else:
            shards = partition([t.record for t in txs], world)
    except MalformedRecord as e:              # every rank sees the same records: the same verdict everywhere, nothing to reduce
        return False, codes.ERR_MALFORMED, e.index
    mine = shards[rank]
    f = ledger.flatten()
    out_cts = (ctypes.c_uint8 * max(64 * len(f["slots"]), 1))()
    out_nonces = (ctypes.c_uint64 * max(len(f["keys"]), 1))()
    ms_set = (ctypes.c_int64 * max(len(f["keys"]), 1))()
    out = LedgerOut(out_cts, out_nonces, ms_set)
    my_txs = [txs[i] for i in mine]
    accept, code, first_bad = verify_fn(my_txs, f, out)
    key = all_reduce_min(verdict_key(accept, code, first_bad, mine), dist, device)
    ok, code, first = decode_key(key)
    if not ok:
        return False, code, first
    # ledger write-back: this rank owns the slots of the accounts its transactions touch
    touched = set()
    for t in my_txs:
        src, dests = tx_accounts(t.record)
        touched.add(src)
        touched.update(dests)
    owned = [f["keys"][acc] in touched for (acc, _) in f["slots"]]
    merged, count = merge_ledger_bytes(bytes(out_cts), owned, dist, device)
    assert all(c <= 1 for c in count), "an account was verified on two ranks"
    final = bytearray(f["balance_cts"])
    for i, c in enumerate(count):
        if c:
            final[64 * i:64 * i + 64] = merged[64 * i:64 * i + 64]
    # multisig configurations set by this rank's transactions, as global transaction indices
    local_ms = [mine[v] if v >= 0 else -1 for v in list(ms_set)[:len(f["keys"])]]
    import torch
    ms_t = torch.tensor(local_ms if local_ms else [-1], dtype=torch.int64)
    if device is not None:
        ms_t = ms_t.to(device)
    dist.all_reduce(ms_t, op=dist.ReduceOp.MAX)
    ledger.absorb(f, bytes(final), [int(x) for x in f["nonces"]][:len(f["keys"])], [int(x) for x in ms_t.cpu()][:len(f["keys"])], txs)
    return True, codes.OK, -1


def gpu_verify_fn(engine, seed=None):
    """verify_fn over the HIP path (xhe_verify_batch of include/xhe.h)."""
    from .api import Verdict, _fresh_seed, _lib

    def run(shard_txs, f, out):
        blob = b"".join(t.record for t in shard_txs)
        offs = [0]
        for t in shard_txs:
            offs.append(offs[-1] + len(t.record))
        offsets = (ctypes.c_uint64 * len(offs))(*offs)
        v = Verdict()
        engine._check(_lib().xhe_verify_batch(engine.ctx, blob, offsets, len(shard_txs), ctypes.byref(f["view"]), ctypes.byref(out), ctypes.byref(v),
                                              _fresh_seed(seed)))
        return bool(v.accept), int(v.err_code), int(v.first_bad_tx)
    return run


def gpu_apply_fn(engine):
    """apply_fn over the HIP path: Transaction::apply_without_verify for a range (xhe_batch_upload with the starting ciphertexts,
    xhe_batch_apply, xhe_batch_read_ledger)."""
    from .api import LedgerOut, Verdict, _lib

    def run(range_txs, ledger, start_cts):
        lib = _lib()
        lib.xhe_batch_upload.restype = lib.xhe_batch_apply.restype = lib.xhe_batch_read_ledger.restype = lib.xhe_batch_set_ledger.restype = ctypes.c_int32
        f = ledger.flatten()
        blob = b"".join(t.record for t in range_txs)
        offs = [0]
        for t in range_txs:
            offs.append(offs[-1] + len(t.record))
        offsets = (ctypes.c_uint64 * len(offs))(*offs)
        ctx, handle, v = ctypes.c_void_p(engine.ctx), ctypes.c_void_p(), Verdict()
        engine._check(lib.xhe_batch_upload(ctx, blob, offsets, ctypes.c_uint64(len(range_txs)), ctypes.byref(f["view"]), ctypes.byref(handle), ctypes.byref(v)))
        try:
            if not v.accept:
                return None
            engine._check(lib.xhe_batch_set_ledger(ctx, handle, start_cts))
            engine._check(lib.xhe_batch_apply(ctx, handle, ctypes.byref(v)))
            if not v.accept:
                return None
            cts = (ctypes.c_uint8 * max(64 * len(f["slots"]), 1))()
            out = LedgerOut(cts, None, None)
            engine._check(lib.xhe_batch_read_ledger(ctx, handle, ctypes.byref(out)))
            return bytes(cts)[:64 * len(f["slots"])]
        finally:
            lib.xhe_batch_free(ctx, handle)
    return run
