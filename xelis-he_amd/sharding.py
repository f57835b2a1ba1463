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
"""
import ctypes

from . import codes

ACCEPT_KEY = (1 << 63) - 1
_AGGREGATE_STAGE = {codes.ERR_GENERIC_PROOF: 1, codes.ERR_RANGE_PROOF: 2}


def tx_accounts(record: bytes):
    """(source, [destinations]) of a flat transaction record (include/xhe.h)."""
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


def tx_weight(record: bytes):
    """Dynamic multiscalar terms of a transaction: 7 per source commitment, 8 per transfer, 4 + 2 lg + m of the range proof."""
    n_items = int.from_bytes(record[2:4], "little") if record[0] == codes.TX_TRANSFERS else 0
    n_cm = int.from_bytes(record[4:6], "little")
    m = 1
    while m < n_cm + n_items:
        m *= 2
    lg = 6 + m.bit_length() - 1
    return 7 * n_cm + 8 * n_items + 4 + 2 * lg + m


def partition(records, world_size):
    """Transaction indices per rank: connected components of the account graph, heaviest first onto the
    lightest rank; indices ascending inside a rank (batch order is preserved)."""
    parent = {}

    def find(x):
        while parent.setdefault(x, x) != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for r in records:
        src, dests = tx_accounts(r)
        a = find(src)
        for d in dests:
            b = find(d)
            if a != b:
                parent[b] = a
    comps = {}
    for i, r in enumerate(records):
        c = comps.setdefault(find(tx_accounts(r)[0]), [0, []])
        c[0] += tx_weight(r)
        c[1].append(i)
    shards, load = [[] for _ in range(world_size)], [0] * world_size
    for w, idx in sorted(comps.values(), key=lambda c: (-c[0], c[1][0])):
        k = min(range(world_size), key=lambda j: (load[j], j))
        shards[k].extend(idx)
        load[k] += w
    return [sorted(s) for s in shards]


def verdict_key(accept, err_code, first_bad_local, shard):
    """The 63-bit key a rank contributes to the MIN all-reduce."""
    if accept:
        return ACCEPT_KEY
    stage = _AGGREGATE_STAGE.get(err_code, 0)
    index = shard[first_bad_local] if (stage == 0 and first_bad_local >= 0) else 0
    return (stage << 60) | (index << 8) | (err_code & 0xff)


def decode_key(key):
    """(accept, err_code, first_bad_tx) of a reduced key."""
    if key == ACCEPT_KEY:
        return True, codes.OK, -1
    stage = key >> 60
    return False, key & 0xff, ((key >> 8) & ((1 << 52) - 1)) if stage == 0 else -1


def all_reduce_min(value, dist, device=None):
    import torch
    t = torch.tensor([value], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.cpu()[0])


def merge_ledger_bytes(own_cts, owned_mask, dist, device=None):
    """Resulting ciphertexts of every ledger slot: each slot is written by at most one rank (its account's
    component lives there); slots no rank owns come back as zero bytes and keep their input value."""
    import torch
    n = len(owned_mask)
    t = torch.zeros(64 * n + n, dtype=torch.int32)
    raw = torch.frombuffer(bytearray(own_cts[:64 * n]), dtype=torch.uint8).to(torch.int32).view(n, 64) if n else torch.zeros(0, 64, dtype=torch.int32)
    mask = torch.tensor([1 if m else 0 for m in owned_mask], dtype=torch.int32)
    t[:64 * n] = (raw * mask[:, None]).reshape(-1)
    t[64 * n:] = mask
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t.cpu()
    return bytes(t[:64 * n].to(torch.uint8).tolist()), [int(x) for x in t[64 * n:]]


def verify_batch_sharded(verify_fn, txs, ledger, dist, device=None):
    """Transaction::verify_batch over all ranks of `dist` (an initialised torch.distributed).

    verify_fn(shard_txs, flat_ledger, out: LedgerOut) -> (accept, err_code, first_bad_local) verifies one
    shard against the flattened ledger and, on accept, fills `out`.  Returns the batch's
    (accept, err_code, first_bad_tx), identical on every rank; on accept `ledger` holds the merged state.
    """
    from .api import LedgerOut
    rank, world = dist.get_rank(), dist.get_world_size()
    shards = partition([t.record for t in txs], world)
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
