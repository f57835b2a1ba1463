#!/usr/bin/env python3
"""Balance decoding of 4 096 ciphertexts over the size of the baby-step table (2^b entries, 16 << b bytes): build time, call time,
walk statistics.  usage: tools/exp_ecdlp_tables.py [max_bits=40] [b,b,...=24,26,28]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xelis_he_amd
from xelis_he_amd import EcdlpTable
from xelis_he_amd.api import Keypair
L = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [24, 26, 28]
eng = xelis_he_amd.Engine(0)
kp = Keypair.keygen(b"k".ljust(32, b"\0"))
rnd = random.Random(3)
n = 4096
amounts = [rnd.randrange(1 << L) for _ in range(n)]
cts = b"".join(kp.encrypt(a, seed=i.to_bytes(4, "little") + bytes(28)) for i, a in enumerate(amounts))
for b in bs:
    t0 = time.perf_counter()
    t = EcdlpTable(eng, b)
    build = time.perf_counter() - t0
    calls = []
    for _ in range(4):
        t0 = time.perf_counter()
        got = t.decrypt_decode(kp.secret, cts, L)
        calls.append(time.perf_counter() - t0)
        assert got == amounts
    print("table 2^%d (%d MiB) built in %.3f s; %d amounts below 2^%d: %.3f ms per call, %s" % (b, (16 << b) >> 20, build, n, L, min(calls[1:]) * 1e3, t.last_stats()), flush=True)
    t.close()
