#!/usr/bin/env python3
"""Balance decoding over table sizes, ranges and batch sizes: amounts known from encryption must come back; prints the walk's
statistics (giant steps = hash-table look-ups, GPU time).  usage: tools/exp_ecdlp.py"""
import sys, random
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xelis_he_amd
from xelis_he_amd import EcdlpTable
from xelis_he_amd.api import Keypair
eng = xelis_he_amd.Engine(0)
kp = Keypair.keygen(b"k".ljust(32, b"\0"))
rnd = random.Random(3)
def run(b, L, n, fixed=()):
    t = EcdlpTable(eng, b)
    amounts = list(fixed) + [rnd.randrange(1 << L) for _ in range(n - len(fixed))]
    cts = b"".join(kp.encrypt(a, seed=i.to_bytes(4, "little") + bytes(28)) for i, a in enumerate(amounts))
    got = t.decrypt_decode(kp.secret, cts, L)
    bad = [(i, amounts[i], got[i]) for i in range(n) if got[i] != amounts[i]]
    print("b", b, "L", L, "n", n, "bad", len(bad), bad[:6], t.last_stats(), flush=True)
    t.close()
run(16, 28, 8)
run(16, 36, 8)
run(16, 36, 300)
run(22, 28, 8)
run(22, 36, 8)
run(22, 40, 8, (0, (1 << 40) - 1, 1 << 39, (1 << 23) + 1))
run(22, 40, 64)
run(22, 40, 1024)
run(22, 40, 4096)
