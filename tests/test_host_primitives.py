"""CPU build of the product's arithmetic headers (radix-2^25.5 field, radix-2^29 Montgomery scalars,
ristretto, Keccak/Merlin/BLAKE3) against the oracle, hashlib and the big-integer spec."""
import hashlib

from helpers import buf, from_le, le32, merlin_script, rng
from oracle import spec_ristretto as S

EDGE_FE = [0, 1, 2, 19, S.P - 1, S.P - 2, S.P, S.P + 1, 2**255 - 1, 2**255 - 20, 2**26 - 1, 2**26, 2**51 - 1, (1 << 255) - (1 << 230)]


def test_field_mul_sq_invert(hostlib):
    r = rng(21)
    vals = EDGE_FE + [r.randrange(2**255) for _ in range(400)]
    o = buf(32)
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        hostlib.xt_fe_mul(le32(a), le32(b), o); assert from_le(o) == a * b % S.P
        hostlib.xt_fe_sq(le32(a), o); assert from_le(o) == a * a % S.P
        hostlib.xt_fe_addsub_mul(le32(a), le32(b), o); assert from_le(o) == (a + b) * (a - b) % S.P
        hostlib.xt_fe_roundtrip(le32(a), o); assert from_le(o) == a % S.P
    for a in [1, 2, S.P - 1, r.randrange(1, S.P)]:
        hostlib.xt_fe_invert(le32(a), o); assert from_le(o) * a % S.P == 1


def test_scalar_ops(hostlib):
    r = rng(22)
    edge = [0, 1, 2, S.L - 1, S.L, S.L + 1, 2**252, 2**256 - 1, 2**253 - 1]
    vals = edge + [r.randrange(2**256) for _ in range(300)]
    o = buf(32)
    for i, a in enumerate(vals):
        b = vals[(i * 5 + 1) % len(vals)]
        c = vals[(i * 11 + 2) % len(vals)]
        d = vals[(i * 3 + 7) % len(vals)]
        hostlib.xt_sc_mul(le32(a), le32(b), o); assert from_le(o) == a * b % S.L
        hostlib.xt_sc_add(le32(a), le32(b), o); assert from_le(o) == (a + b) % S.L
        hostlib.xt_sc_sub(le32(a), le32(b), o); assert from_le(o) == (a - b) % S.L
        hostlib.xt_sc_muladd(le32(a), le32(b), le32(c), le32(d), o); assert from_le(o) == (a * b + c * d + a * d) % S.L
        w = r.randrange(2**512)
        hostlib.xt_sc_from_wide(w.to_bytes(64, "little"), o); assert from_le(o) == w % S.L
        assert bool(hostlib.xt_sc_canonical(le32(a))) == (a < S.L)
    hostlib.xt_sc_from_wide((2**512 - 1).to_bytes(64, "little"), o); assert from_le(o) == (2**512 - 1) % S.L
    for a in [1, 2, S.L - 1, r.randrange(1, S.L)]:
        hostlib.xt_sc_invert(le32(a), o); assert from_le(o) * a % S.L == 1


def test_ristretto_vs_oracle(hostlib, oracle):
    r = rng(23)
    good = [S.encode(S.mul(r.randrange(S.L), S.BASE)) for _ in range(24)] + [bytes(32)]
    cand = good + [r.randbytes(32) for _ in range(200)] + [le32(S.P), le32(1), le32(S.P - 1), bytes([255] * 32)]
    n = len(cand)
    ref_out, ref_ok = buf(128 * n), buf(n)
    oracle.xo_ristretto_decompress_batch(n, b"".join(cand), ref_out, ref_ok)
    for i, e in enumerate(cand):
        o = buf(128)
        ok = hostlib.xt_ristretto_decode(e, o)
        assert bool(ok) == bool(ref_ok[i]), e.hex()
        if ok:
            assert bytes(o) == bytes(ref_out[128 * i:128 * i + 128])
            enc = buf(32)
            hostlib.xt_ristretto_encode(bytes(o), enc)
            assert bytes(enc) == e
        else:  # failed decodes yield the identity
            assert from_le(o[0:32]) == 0 and from_le(o[32:64]) == 1
    for _ in range(16):
        u = r.randbytes(64)
        a, b = buf(32), buf(32)
        hostlib.xt_from_uniform_bytes(u, a); oracle.xo_from_uniform_bytes(u, b)
        assert bytes(a) == bytes(b)


def test_point_ops_scalarmul(hostlib):
    r = rng(24)
    for k in [0, 1, 2, S.L - 1, S.L, r.randrange(S.L), r.randrange(2**256)]:
        p = S.mul(r.randrange(1, S.L), S.BASE)
        o = buf(32)
        assert hostlib.xt_scalarmul(le32(k), S.encode(p), o) == 0
        # double-and-add over all 256 bits computes (k mod 8l) * P; ristretto quotient makes it k mod l
        assert bytes(o) == S.encode(S.mul(k % S.L, p))


def test_hashes(hostlib):
    r = rng(25)
    for n in [0, 1, 71, 72, 73, 135, 136, 137, 1361, 3793]:
        d = r.randbytes(n)
        o = buf(64); hostlib.xt_sha3_512(d, n, o); assert bytes(o) == hashlib.sha3_512(d).digest()
        o = buf(32); hostlib.xt_sha3_256(d, n, o); assert bytes(o) == hashlib.sha3_256(d).digest()
        o = buf(300); hostlib.xt_shake256(d, n, o, 300); assert bytes(o) == hashlib.shake_256(d).digest(300)


def test_blake3_vs_oracle(hostlib, oracle):
    r = rng(26)
    for n in [0, 1, 3, 63, 64, 65, 1023, 1024, 1025, 2048, 2049, 3072, 3073, 4096, 4097, 5000, 8192, 8193, 16384, 31744, 100000]:
        d = r.randbytes(n) if n != 3 else b"abc"
        a, b = buf(32), buf(32)
        hostlib.xt_blake3(d, n, a); oracle.xo_blake3(d, n, b)
        assert bytes(a) == bytes(b), n
    a = buf(32); hostlib.xt_blake3(b"abc", 3, a)
    assert bytes(a).hex() == "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"


def test_merlin_vs_oracle(hostlib, oracle):
    s = merlin_script([("A", b"some label", b"some data"), ("C", b"challenge", 32)])
    o = buf(32)
    assert hostlib.xt_merlin_script(b"test protocol", s, len(s), o, 32) == 32
    assert bytes(o).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    r = rng(27)
    ops = []
    for i in range(60):
        if r.random() < 0.3:
            ops.append(("C", b"c%d" % i, r.choice([1, 32, 64, 165, 166, 167, 400])))
        else:
            ops.append(("A", b"label-%d" % i, r.randbytes(r.choice([0, 1, 8, 32, 160, 166, 200, 700]))))
    s = merlin_script(ops)
    a, b = buf(8192), buf(8192)
    na = hostlib.xt_merlin_script(b"transaction-proof", s, len(s), a, 8192)
    nb = oracle.xo_merlin_script(b"transaction-proof", s, len(s), b, 8192)
    assert na == nb and na > 0 and bytes(a[:na]) == bytes(b[:nb])


def test_transcript_template_matches_transcript(oracle, hostlib):
    """The unrolled transcript (block descriptors replayed over a plain Keccak state) yields the challenges
    of the operation-by-operation Merlin transcript, for every kind of transaction in the fixtures."""
    import ctypes
    import golden_io
    hostlib.xt_transcript_template_challenges.restype = ctypes.c_longlong
    seen = 0
    for case in golden_io.tx_cases():
        for rec_hex, want in zip(case["records"], case["expect"]["challenges"]):
            rec = bytes.fromhex(rec_hex)
            out = buf(32 * 64)
            blocks, zeros = ctypes.c_uint32(), ctypes.c_uint32()
            n = hostlib.xt_transcript_template_challenges(rec, ctypes.c_uint64(len(rec)), out, ctypes.c_uint64(64), ctypes.byref(blocks), ctypes.byref(zeros))
            if case["name"] == "reject_range_proof_rounds":
                assert n < 0, "a proof length that disagrees with the value count has no template (the serial transcript handles it)"
                continue
            assert n * 64 == len(want) and bytes(out)[:32 * n].hex() == want, case["name"]
            # the all-zero encodings a template reports are the ones validate_and_append_point rejects
            assert zeros.value == (1 if "identity_in" in case["name"] else 0) and blocks.value > 10, case["name"]
            seen += 1
    assert seen >= 15


def test_ingest_layout_counts(hostlib):
    """Host ingest (xhe_ingest.h) of synthetic batches on the CPU: the sizes of the tables it lays out for the kernels
    follow from the batch's shape (terms of the merged multiscalar check, challenges, signatures, transcript tiles)."""
    import ctypes
    from xelis_he_amd.api import SyntheticBatch
    hostlib.xt_ingest.restype = ctypes.c_double
    cnt = (ctypes.c_uint64 * 8)()
    for n, t in [(96, 1), (40, 3)]:
        sb = SyntheticBatch(n, n_transfers=t, n_senders=8, seed=b"ingest-%d" % t, threads=4)
        assert hostlib.xt_ingest(sb.records, sb.offsets, ctypes.c_uint64(sb.n_txs), sb.view, cnt) >= 0
        ok, n_points, n_terms, n_sigs, n_ops, n_tiles, n_chal, stream_words = list(cnt)
        m = 1
        while m < 1 + t:
            m *= 2
        lg = 6 + m.bit_length() - 1
        assert ok == 1 and n_sigs == n and n_tiles == (n + 63) // 64
        assert n_terms == n * (7 + 8 * t + 4 + 2 * lg + 1 + t) + 2 + 2 * 64 * m
        assert n_chal == n * (2 + 2 * t + 4 + lg)
        # per transaction: initial balance (first touch only), t transfers out, the closing fee/emit op; t receiver credits
        assert n_ops >= n * (t + 1 + t) and stream_words % (21 * 64) == 0
    sb = SyntheticBatch(32, seed=b"ingest-mixed", mixed=True, threads=4)
    assert hostlib.xt_ingest(sb.records, sb.offsets, ctypes.c_uint64(sb.n_txs), sb.view, cnt) >= 0
    assert cnt[0] == 1 and cnt[3] == 32 + 2 * 2 * 2 and cnt[2] == 2 * 590 + 2 + 2 * 256      # co-signatures: 2 per co-signed transfer
    assert cnt[5] >= 5                                    # one transcript tile per shape at least


def test_ingest_pool_and_arena(hostlib):
    """The host threads and the memory a context keeps between uploads: a parallel pass covers its range exactly once for any
    thread count, the arena hands out aligned disjoint memory and settles on one block; an ingest through both lays out the
    tables of the plain ingest."""
    import ctypes
    hostlib.xt_pool_selftest.restype = ctypes.c_int
    assert hostlib.xt_pool_selftest() == 1
    from xelis_he_amd.api import SyntheticBatch
    hostlib.xt_ingest_pooled_matches.restype = ctypes.c_int
    for sb in (SyntheticBatch(64, seed=b"pooled-mixed", mixed=True), SyntheticBatch(24, n_transfers=3, n_senders=2, seed=b"pooled-chain")):
        assert hostlib.xt_ingest_pooled_matches(sb.records, sb.offsets, ctypes.c_uint64(sb.n_txs), sb.view) == 1
