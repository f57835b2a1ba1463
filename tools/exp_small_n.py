#!/usr/bin/env python3
"""Where the GPU call overtakes one CPU thread: xhe_verify_batch from host buffers (median latency) for batches of 1..256
transactions against the CPU restatement of Transaction::verify_batch (src/tx/verify.rs:487-517) on one host thread, same bytes.
A measurement tool (the CPU restatement is the checker the tests use, timed here as the bench's cpu_baseline leg times it).
usage: tools/exp_small_n.py [transfers=1] [repeats=30] > curve.json"""
import ctypes, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import xelis_he_amd
from xelis_he_amd.api import LedgerOut, SyntheticBatch, Verdict, _lib
import helpers

transfers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lib = _lib()
lib.xhe_verify_batch.restype = ctypes.c_int32
oracle = helpers.load_oracle()
eng = xelis_he_amd.Engine(0)
eng.set_stage_timing(0)
ctx = ctypes.c_void_p(eng.ctx)
rows = []
for n in (1, 2, 4, 8, 16, 24, 32, 48, 64, 128, 256):
    batch = SyntheticBatch(n, transfers, 1, seed=b"small-n", threads=8)
    v = Verdict()
    gpu = []
    for i in range(reps + 5):
        t0 = time.perf_counter()
        rc = lib.xhe_verify_batch(ctx, batch.records, batch.offsets, ctypes.c_uint64(n), batch.view, None, ctypes.byref(v), bytes([i + 1]) * 32)
        gpu.append(time.perf_counter() - t0)
        assert rc == 0 and v.accept == 1
    cpu = []
    for i in range(3 if n > 32 else 5):
        t0 = time.perf_counter()
        oracle.xo_verify_batch(batch.records, batch.offsets, ctypes.c_uint64(n), batch.view, None, ctypes.byref(v), bytes([i + 1]) * 32)
        cpu.append(time.perf_counter() - t0)
        assert v.accept == 1
    row = {"txs": n, "transfers": transfers, "gpu_call_ms": round(statistics.median(gpu[5:]) * 1e3, 3), "cpu_one_thread_ms": round(statistics.median(cpu) * 1e3, 3)}
    # the reference's published per-transaction time on one thread (README: 0.4 ms for 1 transfer, 1.2 ms for 8) scaled to the batch
    ref = {1: 0.4, 8: 1.2}.get(transfers)
    if ref:
        row["reference_published_ms"] = round(ref * n, 3)
    rows.append(row)
    print(json.dumps(row), file=sys.stderr, flush=True)
print(json.dumps({"what": "latency of one verify_batch call by batch size: GPU call from host buffers (median) against one CPU thread", "rows": rows}, indent=1))
