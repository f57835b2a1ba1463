#!/usr/bin/env python3
"""xhe_verify_batch from host buffers on the headline batch over the cut's settings: parts x share of the first (and second) part.
usage: tools/exp_split_sweep.py [txs=100000] [transfers=8] [reps=8]"""
import ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xelis_he_amd
from xelis_he_amd.api import SyntheticBatch, Verdict, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
t = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lib = _lib()
eng = xelis_he_amd.Engine(0)
eng.set_stage_timing(0)
batch = SyntheticBatch(n, t, min(1024, n), seed=b"e2e", threads=16, engine=eng)
v = Verdict()
def run():
    ts = []
    for i in range(reps + 2):
        t0 = time.perf_counter()
        rc = lib.xhe_verify_batch(ctypes.c_void_p(eng.ctx), batch.records, batch.offsets, ctypes.c_uint64(batch.n_txs), batch.view, None, ctypes.byref(v), bytes([i + 1]) * 32)
        ts.append(time.perf_counter() - t0)
        assert rc == 0 and v.accept == 1
    return statistics.median(ts[2:]) * 1e3, min(ts[2:]) * 1e3
shapes = [(6, 50, 100, 100), (6, 50, 100, 50), (7, 50, 100, 50), (7, 50, 100, 25), (7, 25, 50, 50), (8, 25, 50, 25), (8, 50, 100, 25), (6, 50, 100, 25), (5, 50, 100, 50)]
if len(sys.argv) > 4 and sys.argv[4] == "full":
    shapes = [(p, f, s2, 100) for p in (4, 6, 8, 10, 12, 16) for f, s2 in ((100, 100), (50, 100), (25, 100), (25, 50), (12, 50), (12, 25))]
for parts, first, second, last in shapes:
    os.environ.update(XHE_SPLIT_PARTS=str(parts), XHE_SPLIT_FIRST=str(first), XHE_SPLIT_SECOND=str(second), XHE_SPLIT_LAST=str(last))
    lib.xhe_settings_reload()
    med, best = run()
    print("parts %2d first %3d second %3d last %3d: median %.2f ms, best %.2f ms" % (parts, first, second, last, med, best), flush=True)
