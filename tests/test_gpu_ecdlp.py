"""Balance decoding on the GPU (ECDLPInstance::decode, src/elgamal.rs:68-92, after ElGamalSecretKey::decrypt, :140-145) against the
CPU restatement's baby-step/giant-step (oracle/xo_capi.cpp: xo_ecdlp_decode) and against amounts known from encryption.
The reference's own decoder and table files belong to a dependency that is not among its sources: parity unpinned - the contract
(the integer, or none) is what is compared."""
import ctypes
import random

import pytest

from xelis_he_amd import EcdlpTable
from xelis_he_amd.api import ECDLPInstance, Keypair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import xelis_he_amd
    e = xelis_he_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def table(engine):
    t = EcdlpTable(engine, 16)
    yield t
    t.close()


def _points(oracle, amounts):
    """amount*G for every amount, through encryption and the CPU decryption (so the points carry whatever representative
    the ciphertext arithmetic leaves them with)."""
    kp = Keypair.keygen(b"ecdlp-key".ljust(32, b"\0"))
    oracle.xo_decrypt_point.restype = ctypes.c_int32
    pts, cts = [], []
    for i, a in enumerate(amounts):
        ct = kp.encrypt(a, seed=bytes([i & 255, i >> 8 & 255]) + bytes(30))
        out = (ctypes.c_uint8 * 32)()
        assert oracle.xo_decrypt_point(kp.secret, ct, out) == 0
        assert kp.decrypt(ct).as_point() == bytes(out)            # the library's host decryption gives the same encoding
        pts.append(bytes(out))
        cts.append(ct)
    return kp, pts, cts


@pytest.mark.timeout(600)
def test_decode_equals_the_cpu_baby_step_giant_step(engine, table, oracle):
    rnd = random.Random(7)
    amounts = [0, 1, 2, 65535, 65536, 65537, (1 << 17) - 1, 1 << 17, (1 << 24) - 1, 1 << 24, (1 << 28) - 1] + [rnd.randrange(1 << 28) for _ in range(13)]
    kp, pts, cts = _points(oracle, amounts)
    oracle.xo_ecdlp_decode.restype = ctypes.c_int64
    want = [oracle.xo_ecdlp_decode(p, ctypes.c_uint32(28)) for p in pts]
    assert want == amounts
    got = table.decode(b"".join(pts), 28)
    assert got == want
    # the same through the ciphertexts (decryption on the device)
    assert table.decrypt_decode(kp.secret, b"".join(cts), 28) == want
    # one instance through the reference's names
    assert ECDLPInstance(pts[5]).decode(table, 28) == amounts[5]


@pytest.mark.timeout(600)
def test_out_of_range_and_invalid_points(engine, table, oracle):
    kp, pts, cts = _points(oracle, [5, (1 << 20) + 3, (1 << 30) + 1])
    oracle.xo_ecdlp_decode.restype = ctypes.c_int64
    # 2^20 + 3 and 2^30 + 1 are not below 2^20: none, as on the CPU
    assert [oracle.xo_ecdlp_decode(p, ctypes.c_uint32(20)) for p in pts] == [5, -1, -1]
    assert table.decode(b"".join(pts), 20) == [5, None, None]
    assert table.decode(b"".join(pts), 31) == [5, (1 << 20) + 3, (1 << 30) + 1]
    # a point that is no multiple of G within the range (a public key), an invalid encoding, the identity (amount 0)
    bad = bytes([0xff]) * 32
    assert table.decode(kp.pubkey + bad + bytes(32), 24) == [None, None, 0]
    # ciphertexts whose commitment or handle is no valid encoding: none, the valid one beside them is decoded
    assert table.decrypt_decode(kp.secret, cts[0][:32] + bad + cts[0] + bad + cts[0][32:], 24) == [None, 5, None]
    # max_bits below the table's own reach
    assert table.decode(pts[0] + pts[1], 3) == [5, None]
    assert table.decode(b"", 24) == []
    with pytest.raises(Exception):
        table.decode(pts[0], 63)
    with pytest.raises(Exception):
        table.decode(pts[0], 16 + 26)           # more giant steps than a table of 2^16 allows


@pytest.mark.timeout(600)
def test_many_instances_and_wide_range(engine):
    """4096 balances below 2^40 with a table of 2^22 baby steps: every amount comes back (known from encryption); smaller
    calls on the same table (shorter items, candidate list); one instance alone (its range spread over the lanes)."""
    rnd = random.Random(11)
    t = EcdlpTable(engine, 22)
    try:
        kp = Keypair.keygen(b"ecdlp-many".ljust(32, b"\0"))
        amounts = [rnd.randrange(1 << 40) for _ in range(4096)]
        amounts[:4] = [0, (1 << 40) - 1, 1 << 39, (1 << 23) + 1]
        cts = b"".join(kp.encrypt(a, seed=i.to_bytes(4, "little") + bytes(28)) for i, a in enumerate(amounts))
        assert t.decrypt_decode(kp.secret, cts, 40) == amounts
        st = t.last_stats()
        assert st["instances"] == 4096 and 0 < st["steps"] <= 4096 << (40 - 22 - 1)
        assert t.decrypt_decode(kp.secret, cts[:64 * 100], 40) == amounts[:100]
        # 256 instances: items of 64 steps, several per lane, candidates through the list
        assert t.decrypt_decode(kp.secret, cts[:64 * 256], 40) == amounts[:256]
        assert t.last_stats()["steps"] == 256 << (40 - 22 - 1)
        assert t.decrypt_decode(kp.secret, cts[64:128], 40) == amounts[1:2]
        assert t.decrypt_decode(kp.secret, cts[:64 * 3], 39) == [0, None, None]
    finally:
        t.close()


@pytest.mark.timeout(600)
def test_48_bit_balances_with_a_gigabyte_table(engine):
    """A table of 2^26 baby steps (1 GiB) and amounts below 2^48, the width a wallet balance in atomic units needs."""
    rnd = random.Random(13)
    t = EcdlpTable(engine, 26)
    try:
        kp = Keypair.keygen(b"ecdlp-48".ljust(32, b"\0"))
        amounts = [0, (1 << 48) - 1, 1 << 47, (1 << 27) - 1, 1 << 27, (1 << 27) + 1] + [rnd.randrange(1 << 48) for _ in range(58)]
        cts = b"".join(kp.encrypt(a, seed=i.to_bytes(4, "little") + bytes(28)) for i, a in enumerate(amounts))
        assert t.decrypt_decode(kp.secret, cts, 48) == amounts
        assert t.decrypt_decode(kp.secret, cts[:64 * 3], 47) == [0, None, None]
    finally:
        t.close()
