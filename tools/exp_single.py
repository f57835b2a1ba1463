#!/usr/bin/env python3
"""Latency of one transaction through the batch entry points (the Transaction::verify shape, src/tx/verify.rs:519-543):
xhe_verify_batch from host buffers and xhe_batch_verify on the uploaded one-transaction batch, median of the repeats.
usage: tools/exp_single.py [transfers] [repeats]"""
import ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xelis_he_amd
from xelis_he_amd.api import SyntheticBatch, Verdict, _lib
t = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
lib = _lib()
lib.xhe_verify_batch.restype = lib.xhe_batch_upload.restype = lib.xhe_batch_verify.restype = ctypes.c_int32
eng = xelis_he_amd.Engine(0)
eng.set_stage_timing(0)
batch = SyntheticBatch(1, t, 1, seed=b"single", threads=1)
ctx, v = ctypes.c_void_p(eng.ctx), Verdict()
e2e = []
for i in range(reps + 5):
    t0 = time.perf_counter()
    rc = lib.xhe_verify_batch(ctx, batch.records, batch.offsets, ctypes.c_uint64(1), batch.view, None, ctypes.byref(v), bytes([i + 1]) * 32)
    e2e.append(time.perf_counter() - t0)
    assert rc == 0 and v.accept == 1
h = ctypes.c_void_p()
assert lib.xhe_batch_upload(ctx, batch.records, batch.offsets, ctypes.c_uint64(1), batch.view, ctypes.byref(h), ctypes.byref(v)) == 0
res = []
for i in range(reps + 5):
    t0 = time.perf_counter()
    rc = lib.xhe_batch_verify(ctx, h, ctypes.byref(v), bytes([i + 1]) * 32)
    res.append(time.perf_counter() - t0)
    assert rc == 0 and v.accept == 1
lib.xhe_batch_free(ctx, h)
print("single transaction, %d transfer(s): xhe_verify_batch %.3f ms, xhe_batch_verify (resident) %.3f ms (medians of %d)"
      % (t, statistics.median(e2e[5:]) * 1e3, statistics.median(res[5:]) * 1e3, reps))
