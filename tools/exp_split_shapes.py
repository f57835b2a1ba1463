#!/usr/bin/env python3
"""xhe_verify_batch from host buffers over a list of cuts: tools/exp_split_shapes.py <txs> <transfers> <reps> parts:first:second[:last] ...  (two rounds over the list)"""
import ctypes, os, statistics, sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xelis_he_amd
from xelis_he_amd.api import SyntheticBatch, Verdict, _lib
n, t, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lib = _lib()
eng = xelis_he_amd.Engine(0)
eng.set_stage_timing(0)
mixed = os.environ.get("XHE_EXP_MIXED") == "1"
batch = SyntheticBatch(n - n % 16 if mixed else n, t, min(1024, n), seed=b"e2e", threads=16, engine=eng, mixed=mixed)
v = Verdict()
for rnd in range(2):
    for spec in sys.argv[4:]:
        f = spec.split(":")
        env = {"XHE_SPLIT_PARTS": f[0], "XHE_SPLIT_FIRST": f[1], "XHE_SPLIT_SECOND": f[2], "XHE_SPLIT_LAST": f[3] if len(f) > 3 else "100"}
        if f[0] == "auto":
            for k in env: os.environ.pop(k, None)
        else:
            os.environ.update(env)
        lib.xhe_settings_reload()
        ts = []
        for i in range(reps + 2):
            t0 = time.perf_counter()
            rc = lib.xhe_verify_batch(ctypes.c_void_p(eng.ctx), batch.records, batch.offsets, ctypes.c_uint64(batch.n_txs), batch.view, None, ctypes.byref(v), bytes([i + 1]) * 32)
            ts.append(time.perf_counter() - t0)
            assert rc == 0 and v.accept == 1
        print("round %d cut %-14s median %.2f ms, best %.2f ms" % (rnd, spec, statistics.median(ts[2:]) * 1e3, min(ts[2:]) * 1e3), flush=True)
