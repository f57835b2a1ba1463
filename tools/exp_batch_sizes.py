#!/usr/bin/env python3
"""Resident verification time (xhe_batch_verify on an uploaded batch) and the call from host buffers over batch sizes.
usage: tools/exp_batch_sizes.py [transfers=1] [sizes=64,256,1024,4096,16384,32768,65536]"""
import ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xelis_he_amd
from xelis_he_amd.api import SyntheticBatch, Verdict, _lib
t = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 256, 1024, 4096, 16384, 32768, 65536]
lib = _lib()
lib.xhe_batch_upload.restype = lib.xhe_batch_verify.restype = lib.xhe_verify_batch.restype = ctypes.c_int32
eng = xelis_he_amd.Engine(0)
eng.set_stage_timing(0)
ctx = ctypes.c_void_p(eng.ctx)
for n in sizes:
    batch = SyntheticBatch(n, t, min(1024, n), seed=b"sizes", threads=16, engine=eng if n >= 4096 else None)
    v, h = Verdict(), ctypes.c_void_p()
    assert lib.xhe_batch_upload(ctx, batch.records, batch.offsets, ctypes.c_uint64(batch.n_txs), batch.view, ctypes.byref(h), ctypes.byref(v)) == 0
    res, e2e = [], []
    for i in range(14):
        t0 = time.perf_counter()
        assert lib.xhe_batch_verify(ctx, h, ctypes.byref(v), bytes([i + 1]) * 32) == 0 and v.accept == 1
        res.append(time.perf_counter() - t0)
    lib.xhe_batch_free(ctx, h)
    for i in range(10):
        t0 = time.perf_counter()
        assert lib.xhe_verify_batch(ctx, batch.records, batch.offsets, ctypes.c_uint64(batch.n_txs), batch.view, None, ctypes.byref(v), bytes([i + 1]) * 32) == 0 and v.accept == 1
        e2e.append(time.perf_counter() - t0)
    print("%6d transactions x %d transfer(s): resident %.3f ms, from host buffers %.3f ms (medians)" % (n, t, statistics.median(res[4:]) * 1e3, statistics.median(e2e[3:]) * 1e3), flush=True)
    batch.close()
