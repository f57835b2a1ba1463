#!/usr/bin/env python3
"""Many xhe_verify_batch calls of alternating sizes on one context: every verdict accepted, device memory in use at the end close to
what it was after the first round (the pool's blocks are reused, nothing leaks).  usage: tools/soak_calls.py [rounds=30]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xelis_he_amd
from xelis_he_amd.api import SyntheticBatch, Verdict, _lib
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lib = _lib(); lib.xhe_verify_batch.restype = ctypes.c_int32
eng = xelis_he_amd.Engine(0)
batches = [SyntheticBatch(n, t, min(256, n), seed=b"soak%d" % n, threads=16, engine=eng if n >= 4096 else None) for n, t in ((64, 1), (3000, 8), (30000, 1), (40000, 8))]
v = Verdict()
first = None
for r in range(rounds):
    for b in batches:
        rc = lib.xhe_verify_batch(ctypes.c_void_p(eng.ctx), b.records, b.offsets, ctypes.c_uint64(b.n_txs), b.view, None, ctypes.byref(v), bytes([r + 1]) * 32)
        assert rc == 0 and v.accept == 1, (r, b.n_txs, rc, v.accept, v.err_code)
    free, total = torch.cuda.mem_get_info()
    used = (total - free) >> 20
    if first is None: first = used
    if r % 5 == 0 or r == rounds - 1: print("round %d: device memory in use %d MiB" % (r, used), flush=True)
assert used <= first + 256, (first, used)
print("soak ok: %d calls" % (rounds * len(batches)))
