"""xhe_msm_vartime_static (no allocation: staging and workspace live in the library's static device array) against
xhe_msm_vartime through the C ABI.  Integer work: the bar is bit-exact.  Two calls in a row on one context, the second
with a smaller input, so stale contents of the shared array would show."""
import pytest

from helpers import buf, le32, rng
from oracle import spec_ristretto as S

pytestmark = pytest.mark.gpu

MAX_TERMS = 4096      # XHE_MSM_STATIC_MAX_TERMS of include/xhe.h


def _points(oracle, r, n):
    base = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
    out = []
    for _ in range(n):
        o, idf = buf(32), buf(1)
        oracle.xo_msm_vartime(1, le32(r.randrange(S.L)), base, o, idf)
        out.append(bytes(o))
    return out


def _inputs(oracle, seed, n):
    r = rng(seed)
    pool = _points(oracle, r, 48)
    pts = b"".join(pool[r.randrange(len(pool))] for _ in range(n))
    kinds = [0, 1, S.L - 1, 2**252, S.L, 2**256 - 1, (S.L - 1) // 2, (S.L + 1) // 2]
    scal = b"".join(le32(kinds[i % len(kinds)] if i % 13 == 0 else r.randrange(S.L)) for i in range(n))
    return scal, pts


def test_static_msm_matches_allocating_entry_point(oracle):
    import xelis_he_amd
    engine = xelis_he_amd.Engine(0)
    try:
        for seed, n in [(9001, MAX_TERMS), (9002, 300), (9003, 4095), (9004, 1), (9005, 0)]:
            scal, pts = _inputs(oracle, seed, n)
            want = engine.msm_vartime(scal, pts)
            got = engine.msm_vartime_static(scal, pts)
            assert got == want, (n, got, want)
            o, idf = buf(32), buf(1)
            assert oracle.xo_msm_vartime(n, scal, pts, o, idf) == 0
            assert got == (bytes(o), bool(idf[0])), n
        # a sum that cancels, right after a large call: the identity flag must not be stale
        r = rng(9100)
        pool = _points(oracle, r, 4)
        half = [r.randrange(S.L) for _ in range(100)]
        scal = b"".join(le32(k) for k in half) + b"".join(le32((S.L - k) % S.L) for k in half)
        pts = b"".join(pool[i % 4] for i in range(100)) * 2
        assert engine.msm_vartime_static(scal, pts) == (bytes(32), True)
        scal, pts = _inputs(oracle, 9006, 77)
        assert engine.msm_vartime_static(scal, pts) == engine.msm_vartime(scal, pts)
        # beyond the stated maximum: a clear error, no allocation, nothing launched
        scal, pts = _inputs(oracle, 9007, MAX_TERMS + 1)
        with pytest.raises(xelis_he_amd.XheError, match="XHE_MSM_STATIC_MAX_TERMS"):
            engine.msm_vartime_static(scal, pts)
        # an undecodable point is refused as by the allocating entry point
        with pytest.raises(xelis_he_amd.XheError):
            engine.msm_vartime_static(le32(5) + le32(7), le32(1) + bytes(32))
        scal, pts = _inputs(oracle, 9008, 64)
        assert engine.msm_vartime_static(scal, pts) == engine.msm_vartime(scal, pts)
    finally:
        engine.close()
