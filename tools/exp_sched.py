#!/usr/bin/env python3
"""Stage times of one synthetic batch (generated once, verified several times), optionally under variants of the library
selected through the environment: variant bit 0 = one-pass multiscalar sort (XHE_MSM_ONEPASS), bit 1 = signatures behind the bulk
decode on its stream instead of a stream of their own (XHE_SIG_STREAM_MIN_TXS=4294967295).
usage: tools/exp_sched.py <txs> <transfers> <variant values...>"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xelis_he_amd
from xelis_he_amd.api import SyntheticBatch, Verdict, _lib
n, t = int(sys.argv[1]), int(sys.argv[2])
exps = [int(x) for x in sys.argv[3:]] or [0]
lib = _lib(); lib.xhe_batch_upload.restype = ctypes.c_int32; lib.xhe_batch_verify.restype = ctypes.c_int32
eng = xelis_he_amd.Engine(0)
batch = SyntheticBatch(n, t, min(1024, n), seed=b"exp", threads=16)
h = ctypes.c_void_p(); sv = Verdict()
eng._check(lib.xhe_batch_upload(ctypes.c_void_p(eng.ctx), batch.records, batch.offsets, ctypes.c_uint64(batch.n_txs), batch.view, ctypes.byref(h), ctypes.byref(sv)))
v = Verdict()
for e in exps:
    if e & 1:
        os.environ["XHE_MSM_ONEPASS"] = "1"
    else:
        os.environ.pop("XHE_MSM_ONEPASS", None)
    if e & 2:
        os.environ["XHE_SIG_STREAM_MIN_TXS"] = "4294967295"
    else:
        os.environ.pop("XHE_SIG_STREAM_MIN_TXS", None)
    lib.xhe_settings_reload()
    acc = {}
    reps = 10
    t0 = None
    for i in range(reps + 2):
        if i == 2: t0 = time.perf_counter()
        eng._check(lib.xhe_batch_verify(ctypes.c_void_p(eng.ctx), h, ctypes.byref(v), bytes([i + 1]) * 32))
        assert v.accept == 1
        if i >= 2:
            for k, ms in eng.stage_times(): acc[k] = acc.get(k, 0) + ms / reps
    wall = (time.perf_counter() - t0) / reps * 1e3
    print("EXP", e, "wall_ms %.3f" % wall, " ".join("%s=%.3f" % (k, x) for k, x in acc.items()), flush=True)
