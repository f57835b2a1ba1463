#!/usr/bin/env python3
"""xhe_verify_batches over a run of identical synthetic batches from host buffers; XHE_UPLOAD_TIMES=1 prints the timeline.
usage: tools/exp_batches.py <txs> <transfers> <batches> [repeats]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xelis_he_amd
from xelis_he_amd.api import SyntheticBatch, Verdict, LedgerOut, _lib
n, t, nb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
lib = _lib()
eng = xelis_he_amd.Engine(0)
batch = SyntheticBatch(n, t, min(1024, n), seed=b"batches", threads=16, engine=eng)


class BatchArgs(ctypes.Structure):
    _fields_ = [("records", ctypes.c_char_p), ("offsets", ctypes.POINTER(ctypes.c_uint64)), ("n_txs", ctypes.c_uint64), ("view", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("seed", ctypes.c_char_p)]


view = batch.view.contents
out = LedgerOut((ctypes.c_uint8 * max(64 * view.n_balances, 1))(), (ctypes.c_uint64 * max(view.n_accounts, 1))(), (ctypes.c_int64 * max(view.n_accounts, 1))())
lib.xhe_verify_batches.restype = ctypes.c_int32
for r in range(reps):
    arr = (BatchArgs * nb)()
    seeds = [(3000 + 16 * r + i).to_bytes(8, "little") * 4 for i in range(nb)]
    for i in range(nb):
        arr[i].records, arr[i].offsets, arr[i].n_txs = batch.records, batch.offsets, batch.n_txs
        arr[i].view, arr[i].out, arr[i].seed = ctypes.cast(batch.view, ctypes.c_void_p), ctypes.cast(ctypes.pointer(out), ctypes.c_void_p), seeds[i]
    vs = (Verdict * nb)()
    t0 = time.perf_counter()
    rc = lib.xhe_verify_batches(ctypes.c_void_p(eng.ctx), ctypes.c_uint64(nb), arr, vs)
    dt = time.perf_counter() - t0
    assert rc == 0 and all(v.accept == 1 for v in vs)
    print("BATCHES %d x %d tx: %.3f ms, %.3f ms per batch (%.0f tx/s)" % (nb, n, dt * 1e3, dt * 1e3 / nb, nb * n / dt), flush=True)
