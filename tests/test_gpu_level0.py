"""Parity of the level-0 HIP entry points (decompress, compress, multiscalar multiplication) with the
CPU oracle, through the C ABI.  Integer work: the bar is bit-exact."""
import os

import pytest

from helpers import buf, le32, rng
from oracle import spec_ristretto as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import xelis_he_amd
    e = xelis_he_amd.Engine(0)
    yield e
    e.close()


def _oracle_points(oracle, r, n):
    """n valid encodings: small multiples walk (cheap) of random base points."""
    scal = b"".join(le32(r.randrange(S.L)) for _ in range(n))
    out = []
    base = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
    for i in range(n):
        o, idf = buf(32), buf(1)
        oracle.xo_msm_vartime(1, scal[32 * i:32 * i + 32], base, o, idf)
        out.append(bytes(o))
    return out


def test_decompress_compress_parity(engine, oracle):
    r = rng(41)
    good = _oracle_points(oracle, r, 300) + [bytes(32)]
    bad = [r.randbytes(32) for _ in range(300)] + [le32(S.P), le32(1), le32(S.P - 1), bytes([255] * 32)]
    enc = good + bad
    r.shuffle(enc)
    n = len(enc)
    blob = b"".join(enc)
    xyzt, ok = engine.decompress_batch(blob)
    ref_xyzt, ref_ok = buf(128 * n), buf(n)
    oracle.xo_ristretto_decompress_batch(n, blob, ref_xyzt, ref_ok)
    assert ok == bytes(ref_ok)
    for i in range(n):
        if ok[i]:
            assert xyzt[128 * i:128 * i + 128] == bytes(ref_xyzt[128 * i:128 * i + 128]), i
    # compress what decoded, expect the original encodings
    valid = [i for i in range(n) if ok[i]]
    packed = b"".join(xyzt[128 * i:128 * i + 128] for i in valid)
    back = engine.compress_batch(packed)
    assert back == b"".join(enc[i] for i in valid)
    assert engine.decompress_batch(b"") == (b"", b"")


@pytest.mark.parametrize("n", [0, 1, 2, 3, 64, 189, 190, 255, 256, 499, 800, 4096, 5000, 40000, 199999, 200000, 262144])
def test_msm_parity(engine, oracle, n):
    r = rng(1000 + n)
    pool = _oracle_points(oracle, r, min(max(n, 1), 64))
    pts = b"".join(pool[i % len(pool)] for i in range(n))
    kinds = [0, 1, S.L - 1, 2**252, S.L, 2**256 - 1, (S.L - 1) // 2, (S.L + 1) // 2, (S.L - 3) // 2, 2**251, 2**251 - 1, S.L - 2**251]
    scal = b"".join(le32(kinds[i % len(kinds)] if i % 17 == 0 else r.randrange(S.L)) for i in range(n))
    got, ident = engine.msm_vartime(scal, pts)
    o, idf = buf(32), buf(1)
    assert oracle.xo_msm_vartime(n, scal, pts, o, idf) == 0
    assert got == bytes(o)
    assert ident == bool(idf[0])


@pytest.mark.parametrize("n,kind", [(3000, "equal"), (3000, "two"), (20000, "small"), (70000, "equal"), (210000, "equal"), (210000, "two"), (210000, "small")])
def test_msm_skewed_digits(engine, oracle, n, kind):
    """Digit distributions that put most terms into a handful of buckets (buckets spanning many chunks; from 200 000
    terms on the sort runs in two passes and all of a window's terms meet in one coarse bin)."""
    r = rng(5000 + n)
    pool = _oracle_points(oracle, r, 32)
    pts = b"".join(pool[r.randrange(32)] for _ in range(n))
    if kind == "equal":
        k = r.randrange(S.L)
        vals = [k] * n
    elif kind == "two":
        a, b = r.randrange(S.L), r.randrange(S.L)
        vals = [a if r.random() < 0.9 else b for _ in range(n)]
    else:
        vals = [r.randrange(4) for _ in range(n)]
    scal = b"".join(le32(v) for v in vals)
    got, ident = engine.msm_vartime(scal, pts)
    o, idf = buf(32), buf(1)
    assert oracle.xo_msm_vartime(n, scal, pts, o, idf) == 0
    assert got == bytes(o)
    assert ident == bool(idf[0])


@pytest.mark.parametrize("n,kind", [(262144, "random"), (210000, "equal"), (210000, "small")])
def test_msm_two_pass_sort_without_lds_staging(engine, oracle, switch, n, kind):
    """The two-pass bucket sort in its direct form (every entry written where its counter says, XHE_MSM_SORT_STAGED=0) and in the
    default form (entries ordered inside LDS, written out as runs) give the oracle's sum."""
    r = rng(9000 + n)
    pool = _oracle_points(oracle, r, 32)
    pts = b"".join(pool[r.randrange(32)] for _ in range(n))
    if kind == "equal":
        vals = [r.randrange(S.L)] * n
    elif kind == "small":
        vals = [r.randrange(4) for _ in range(n)]
    else:
        vals = [r.randrange(S.L) for _ in range(n)]
    scal = b"".join(le32(v) for v in vals)
    o, idf = buf(32), buf(1)
    assert oracle.xo_msm_vartime(n, scal, pts, o, idf) == 0
    for staged in ("0", None):
        switch("XHE_MSM_SORT_STAGED", staged)
        got, ident = engine.msm_vartime(scal, pts)
        assert got == bytes(o) and ident == bool(idf[0]), staged


def test_msm_identity_and_cancellation(engine, oracle):
    r = rng(77)
    pool = _oracle_points(oracle, r, 8)
    n = 1000
    half = [r.randrange(S.L) for _ in range(n // 2)]
    scal = b"".join(le32(k) for k in half) + b"".join(le32((S.L - k) % S.L) for k in half)
    pts = b"".join(pool[i % 8] for i in range(n // 2)) * 2
    got, ident = engine.msm_vartime(scal, pts)
    assert got == bytes(32) and ident
    # identity points with arbitrary scalars
    got, ident = engine.msm_vartime(b"".join(le32(r.randrange(S.L)) for _ in range(10)), bytes(320))
    assert got == bytes(32) and ident


def test_msm_rejects_bad_point(engine):
    import xelis_he_amd
    with pytest.raises(xelis_he_amd.XheError):
        engine.msm_vartime(le32(5) + le32(7), le32(1) + bytes(32))


def test_wide_arithmetic_selftest():
    """The one-point-per-wavefront arithmetic of the Horner stage (xhe_wide.h) against the one-point-per-lane arithmetic, on
    the device: products at the magnitude limits, record loads, a chain of doublings and additions (tools/test_wide.hip,
    built by __graft_entry__.build())."""
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "test_wide")
    if not os.path.exists(exe):
        pytest.skip("tools/test_wide is not built (__graft_entry__.build() builds it)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "PASS" in r.stdout, (r.returncode, r.stdout, r.stderr)
