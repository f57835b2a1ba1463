"""The HIP path against the committed fixtures (tests/golden/), through the C ABI; needs no CPU checker at
run time.  Bit-exact: verdict, error code, first failing index, ledger bytes, multiscalar results."""
import ctypes
import hashlib
import random

import pytest

import golden_io
from helpers import le32

pytestmark = pytest.mark.gpu
L = 2**252 + 27742317777372353535851937790883648493


@pytest.fixture(scope="module")
def engine():
    import xelis_he_amd
    e = xelis_he_amd.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("case", golden_io.tx_cases(), ids=lambda c: c["name"])
def test_tx_cases_gpu(engine, case):
    from xelis_he_amd.api import _lib
    c = golden_io.FlatCase(case)
    for seed in (c.seed, bytes(range(32))):
        rc = _lib().xhe_verify_batch(engine.ctx, c.blob, c.offsets, len(c.records), ctypes.byref(c.view), ctypes.byref(c.out), ctypes.byref(c.verdict), seed)
        engine._check(rc)
        c.check()


def test_generator_multiples_gpu(engine):
    g = golden_io.primitives()
    base = bytes.fromhex(g["rfc9496_generator_multiples"][1])
    for k, enc in enumerate(g["rfc9496_generator_multiples"]):
        got, ident = engine.msm_vartime(le32(k), base)
        assert got.hex() == enc and ident == (k == 0)
    ks = [int(v["k"], 16) for v in g["spec_generator_multiples"]]
    for v, k in zip(g["spec_generator_multiples"], ks):
        got, _ = engine.msm_vartime(le32(k), base)
        assert got.hex() == v["encoding"]
    # the same as one sum: (sum k_i) * B
    got, _ = engine.msm_vartime(b"".join(le32(k) for k in ks), base * len(ks))
    want, _ = engine.msm_vartime(le32(sum(ks) % L), base)
    assert got == want


def test_decode_vectors_gpu(engine):
    bad = golden_io.primitives()["decode_accepts"]
    _, ok = engine.decompress_batch(b"".join(bytes.fromhex(v["bytes"]) for v in bad))
    assert [int(x) for x in ok] == [v["ok"] for v in bad]


def test_msm_vectors_gpu(engine):
    """Inputs regenerated from the seed on the GPU itself (k_i * B through the library), checked by hash."""
    g = golden_io.primitives()
    base = bytes.fromhex(g["rfc9496_generator_multiples"][1])
    for v in g["msm"]:
        r, n = random.Random(v["seed"]), v["n"]
        pool = [engine.msm_vartime(le32(r.randrange(L)), base)[0] for _ in range(min(max(n, 1), 32))]
        pts = b"".join(pool[i % len(pool)] for i in range(n))
        scal = b"".join(le32(r.randrange(L)) for _ in range(n))
        assert hashlib.sha3_256(scal + pts).hexdigest() == v["input_sha3_256"]
        got, _ = engine.msm_vartime(scal, pts)
        assert got.hex() == v["result"], n
