"""Pins the CPU oracle's primitives: RFC 9496 vectors, the published Pedersen blinding generator,
hashlib SHA3/SHAKE, Merlin and BLAKE3 vectors, and the big-integer spec (oracle/spec_ristretto.py).
The reference holds no golden vectors of its own (every test of /root/reference/src/lib.rs draws
fresh random keys), so these public vectors are what pins the oracle."""
import hashlib
import os

import pytest

from helpers import buf, from_le, le32, merlin_script, rng
from oracle import spec_ristretto as S

# RFC 9496 appendix A.1: multiples 0..3 of the generator
RFC9496_MULTIPLES = [
    "0000000000000000000000000000000000000000000000000000000000000000",
    "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76",
    "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919",
    "94741f5d5d52755ece4f23f044ee27d5d1ea1e2bd196b462166b16152a9d0259",
]
# PedersenGens::default().B_blinding of dalek-bulletproofs = the reference's H (elgamal.rs:16-24)
PEDERSEN_H = "8c9240b456a9e6dc65c377a1048d745f94a08cdb7f44cbcd7b46f34048871134"


def test_generator_multiples(oracle):
    pts = b"".join(bytes.fromhex(RFC9496_MULTIPLES[1]) for _ in range(4))
    for k in range(4):
        out, ident = buf(32), buf(1)
        rc = oracle.xo_msm_vartime(1, le32(k), bytes.fromhex(RFC9496_MULTIPLES[1]), out, ident)
        assert rc == 0 and bytes(out).hex() == RFC9496_MULTIPLES[k]
        assert bool(ident[0]) == (k == 0)
    # l * B = identity
    out, ident = buf(32), buf(1)
    oracle.xo_msm_vartime(2, le32(S.L - 1) + le32(1), pts[:64], out, ident)
    assert bytes(out) == bytes(32) and ident[0] == 1


def test_pedersen_h(oracle):
    out = buf(32)
    oracle.xo_pedersen_h(out)
    assert bytes(out).hex() == PEDERSEN_H
    assert S.encode(S.H_POINT).hex() == PEDERSEN_H


def test_bad_encodings(oracle):
    p = S.P
    bad = [
        le32(p),                # non-canonical field element
        le32(p + 2),            # non-canonical
        le32(2**255 - 1),       # bit pattern above p
        le32(1),                # negative (odd) s
        le32(p - 1),            # canonical but odd
        bytes([0xFF] * 32),     # high bit set
    ]
    r = rng(3)
    cand = [r.randbytes(32) for _ in range(200)]
    bad += [c for c in cand if S.decode(c) is None][:40]
    n = len(bad)
    out, ok = buf(128 * n), buf(n)
    oracle.xo_ristretto_decompress_batch(n, b"".join(bad), out, ok)
    assert not any(ok)


def test_decode_encode_vs_spec(oracle):
    r = rng(5)
    pts = [S.mul(r.randrange(S.L), S.BASE) for _ in range(32)]
    enc = [S.encode(p) for p in pts] + [bytes(32)]
    n = len(enc)
    out, ok = buf(128 * n), buf(n)
    oracle.xo_ristretto_decompress_batch(n, b"".join(enc), out, ok)
    assert all(ok)
    for i, e in enumerate(enc):
        sp = S.decode(e)
        x, y = from_le(out[128 * i:128 * i + 32]), from_le(out[128 * i + 32:128 * i + 64])
        assert (x, y) == (sp[0], sp[1])
    # encode from projective (non-affine) representatives, including torsion-shifted ones
    reps = []
    for p in pts[:8]:
        q = S.add(p, p)
        reps.append(q)
    inb = b"".join(b"".join(le32(c) for c in q) for q in reps)
    o = buf(32 * len(reps))
    oracle.xo_ristretto_compress_batch(len(reps), inb, o)
    for i, q in enumerate(reps):
        assert bytes(o[32 * i:32 * i + 32]) == S.encode(q)


def test_field_and_scalar(oracle):
    r = rng(7)
    for _ in range(300):
        a, b = r.randrange(2**255), r.randrange(2**255)
        o = buf(32)
        oracle.xo_fe_mul(le32(a), le32(b), o)
        assert from_le(o) == a * b % S.P
        a, b = r.randrange(2**256), r.randrange(2**256)
        oracle.xo_sc_mul(le32(a), le32(b), o)
        assert from_le(o) == a * b % S.L
        oracle.xo_sc_add(le32(a), le32(b), o)
        assert from_le(o) == (a + b) % S.L
        w = r.randrange(2**512)
        oracle.xo_sc_from_wide(w.to_bytes(64, "little"), o)
        assert from_le(o) == w % S.L
    for x in (1, 2, S.L - 1, r.randrange(1, S.L)):
        o = buf(32)
        oracle.xo_sc_invert(le32(x), o)
        assert from_le(o) * x % S.L == 1
    for x in (1, 2, S.P - 1, r.randrange(1, S.P)):
        o = buf(32)
        oracle.xo_fe_invert(le32(x), o)
        assert from_le(o) * x % S.P == 1


@pytest.mark.parametrize("n", [1, 2, 3, 17, 189, 190, 499, 500, 800])
def test_msm_vs_spec(oracle, n):
    r = rng(100 + n)
    base_pts = [S.mul(r.randrange(S.L), S.BASE) for _ in range(min(n, 12))]
    pts = [base_pts[i % len(base_pts)] for i in range(n)]
    scal = [r.randrange(S.L) for _ in range(n)]
    out, ident = buf(32), buf(1)
    rc = oracle.xo_msm_vartime(n, b"".join(le32(s) for s in scal), b"".join(S.encode(p) for p in pts), out, ident)
    assert rc == 0
    # group by base point to keep the big-int side cheap
    acc = S.IDENT
    for j, bp in enumerate(base_pts):
        k = sum(scal[i] for i in range(n) if i % len(base_pts) == j) % S.L
        acc = S.add(acc, S.mul(k, bp))
    assert bytes(out) == S.encode(acc)


def test_from_uniform_bytes(oracle):
    r = rng(11)
    for _ in range(16):
        u = r.randbytes(64)
        o = buf(32)
        oracle.xo_from_uniform_bytes(u, o)
        assert bytes(o) == S.encode(S.from_uniform_bytes(u))


def test_sha3_shake(oracle):
    r = rng(13)
    for n in [0, 1, 71, 72, 73, 135, 136, 137, 200, 1361, 3793]:
        d = r.randbytes(n)
        o = buf(64); oracle.xo_sha3_512(d, n, o); assert bytes(o) == hashlib.sha3_512(d).digest()
        o = buf(32); oracle.xo_sha3_256(d, n, o); assert bytes(o) == hashlib.sha3_256(d).digest()
        o = buf(500); oracle.xo_shake256(d, n, o, 500); assert bytes(o) == hashlib.shake_256(d).digest(500)


def test_blake3_vectors(oracle):
    o = buf(32)
    oracle.xo_blake3(b"", 0, o)
    assert bytes(o).hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    oracle.xo_blake3(b"abc", 3, o)
    assert bytes(o).hex() == "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"


def test_merlin_vectors(oracle):
    # "test protocol" vectors shared by the Merlin implementations (simple and 32-round transcripts)
    s = merlin_script([("A", b"some label", b"some data"), ("C", b"challenge", 32)])
    o = buf(32)
    assert oracle.xo_merlin_script(b"test protocol", s, len(s), o, 32) == 32
    assert bytes(o).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    ops, data, chl = [("A", b"step1", b"some data")], bytes([99]) * 1024, None
    for _ in range(32):
        ops = ops + [("C", b"challenge", 32)]
        s = merlin_script(ops)
        big = buf(32 * 40)
        n = oracle.xo_merlin_script(b"test protocol", s, len(s), big, 32 * 40)
        chl = bytes(big[n - 32:n])
        ops = ops + [("A", b"bigdata", data), ("A", b"challengedata", chl)]
    assert chl.hex() == "a8c933f54fae76e3f9bea93648c1308e7dfa2152dd51674ff3ca438351cf003c"


def test_bp_generators_chain(oracle):
    # BulletproofGens share j, index i = i-th 64-byte SHAKE256("GeneratorsChain" || 'G'/'H' || LE32(j)) block
    n = 130
    g, h = buf(32 * n), buf(32 * n)
    oracle.xo_bp_generators(n, g, h)
    for which, out in ((b"G", g), (b"H", h)):
        for j in range(3):
            stream = hashlib.shake_256(b"GeneratorsChain" + which + j.to_bytes(4, "little")).digest(64 * 64)
            for i in (0, 1, 63):
                idx = 64 * j + i
                if idx >= n:
                    continue
                assert bytes(out[32 * idx:32 * idx + 32]) == S.encode(S.from_uniform_bytes(stream[64 * i:64 * i + 64]))
