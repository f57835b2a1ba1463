#!/usr/bin/env python3
"""End-to-end time of xhe_verify_batch from host buffers (parse + upload + verify + ledger read-back) on one synthetic
batch, repeated; XHE_UPLOAD_TIMES=1 in the environment prints the upload's phases.
usage: tools/exp_e2e.py <txs> <transfers> [repeats] [parts,parts,...]
The last argument repeats the measurement with XHE_SPLIT_PARTS set to each value (1 = the batch in one piece; default: the
library's own choice)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xelis_he_amd
from xelis_he_amd.api import SyntheticBatch, Verdict, _lib
n, t = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
lib = _lib()
eng = xelis_he_amd.Engine(0)
mixed = os.environ.get("XHE_EXP_MIXED") == "1"           # the mixed batch (groups of 16) instead of transfers
batch = SyntheticBatch(n - n % 16 if mixed else n, t, min(1024, n), seed=b"e2e", threads=16, engine=eng, mixed=mixed)
v = Verdict()
settings = sys.argv[4].split(",") if len(sys.argv) > 4 else [None]
for parts in settings:
    if parts is not None:
        os.environ["XHE_SPLIT_PARTS"] = parts
        lib.xhe_settings_reload()
    for i in range(reps):
        t0 = time.perf_counter()
        rc = lib.xhe_verify_batch(ctypes.c_void_p(eng.ctx), batch.records, batch.offsets, ctypes.c_uint64(batch.n_txs), batch.view, None, ctypes.byref(v), bytes([i + 1]) * 32)
        dt = time.perf_counter() - t0
        assert rc == 0 and v.accept == 1, (rc, v.accept, v.code)
        print("E2E %d tx, parts %s: %.3f ms  (%.0f tx/s)" % (n, parts or "auto", dt * 1e3, n / dt), flush=True)
