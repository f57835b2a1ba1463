"""The CPU oracle and the CPU build of the product's arithmetic headers against the committed fixtures
(tests/golden/): public known answers, big-integer specification values, and frozen transaction cases."""
import ctypes
import hashlib

import pytest

import golden_io
from helpers import buf, le32, merlin_script


def test_primitives_oracle(oracle):
    g = golden_io.primitives()
    base = bytes.fromhex(g["rfc9496_generator_multiples"][1])
    for k, enc in enumerate(g["rfc9496_generator_multiples"]):
        o, idf = buf(32), buf(1)
        assert oracle.xo_msm_vartime(1, le32(k), base, o, idf) == 0 and bytes(o).hex() == enc
    for v in g["spec_generator_multiples"]:
        o, idf = buf(32), buf(1)
        oracle.xo_msm_vartime(1, le32(int(v["k"], 16)), base, o, idf)
        assert bytes(o).hex() == v["encoding"]
    h = buf(32)
    oracle.xo_pedersen_h(h)
    assert bytes(h).hex() == g["pedersen_blinding_generator"] == "8c9240b456a9e6dc65c377a1048d745f94a08cdb7f44cbcd7b46f34048871134"
    gg, hh = buf(128), buf(128)
    oracle.xo_bp_generators(4, gg, hh)
    assert [bytes(gg)[32 * i:32 * i + 32].hex() for i in range(4)] == g["bulletproof_generators_party0"]["G"]
    assert [bytes(hh)[32 * i:32 * i + 32].hex() for i in range(4)] == g["bulletproof_generators_party0"]["H"]
    for msg, digest in g["blake3"].items():
        o = buf(32)
        oracle.xo_blake3(bytes.fromhex(msg), len(msg) // 2, o)
        assert bytes(o).hex() == digest
    for v in g["merlin_test_protocol"]:
        ops = [(op[0], op[1].encode(), bytes.fromhex(op[2]) if op[0] == "A" else op[2]) for op in v["ops"]]
        script = merlin_script(ops)
        o = buf(64)
        n = oracle.xo_merlin_script(b"test protocol", script, len(script), o, 64)
        assert bytes(o)[:n].hex() == v["challenge"]
    bad = g["decode_accepts"]
    blob = b"".join(bytes.fromhex(v["bytes"]) for v in bad)
    ok, xyzt = buf(len(bad)), buf(128 * len(bad))
    oracle.xo_ristretto_decompress_batch(len(bad), blob, xyzt, ok)
    for i, v in enumerate(bad):
        assert int(ok[i]) == v["ok"] == v["spec_ok"]


def test_msm_vectors_oracle(oracle):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden import msm_inputs
    for v in golden_io.primitives()["msm"]:
        scal, pts = msm_inputs(oracle, v["seed"], v["n"])
        assert hashlib.sha3_256(scal + pts).hexdigest() == v["input_sha3_256"]
        o, idf = buf(32), buf(1)
        assert oracle.xo_msm_vartime(v["n"], scal, pts, o, idf) == 0
        assert bytes(o).hex() == v["result"]


@pytest.mark.parametrize("case", golden_io.tx_cases(), ids=lambda c: c["name"])
def test_tx_cases_oracle(oracle, case):
    c = golden_io.FlatCase(case)
    oracle.xo_verify_batch(c.blob, c.offsets, ctypes.c_uint64(len(c.records)), ctypes.byref(c.view), ctypes.byref(c.out), ctypes.byref(c.verdict), c.seed)
    c.check()
    for rec, want in zip(c.records, c.expect["challenges"]):
        cb = buf(32 * 64)
        n = oracle.xo_tx_challenges(rec, ctypes.c_uint64(len(rec)), cb, ctypes.c_uint64(64))
        assert (bytes(cb)[:32 * n].hex() if n >= 0 else None) == want
    # the verdict does not depend on the batch factors
    c2 = golden_io.FlatCase(case)
    oracle.xo_verify_batch(c2.blob, c2.offsets, ctypes.c_uint64(len(c2.records)), ctypes.byref(c2.view), ctypes.byref(c2.out), ctypes.byref(c2.verdict), bytes(range(32)))
    c2.check()
