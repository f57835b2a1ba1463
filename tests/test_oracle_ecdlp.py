"""The CPU restatement of balance decoding (ElGamalSecretKey::decrypt, src/elgamal.rs:140-145, then ECDLPInstance::decode, :68-92):
amounts known from encryption come back; amounts outside the range and points that are no small multiple of G give none."""
import ctypes

from xelis_he_amd.api import Keypair


def test_decrypt_then_decode_returns_the_encrypted_amount(oracle):
    oracle.xo_decrypt_point.restype = ctypes.c_int32
    oracle.xo_ecdlp_decode.restype = ctypes.c_int64
    kp = Keypair.keygen(b"oracle-ecdlp".ljust(32, b"\0"))
    for i, amount in enumerate([0, 1, 65535, 65536, 99999, (1 << 20) + 17, (1 << 24) - 1]):
        ct = kp.encrypt(amount, seed=bytes([i]) * 32)
        out = (ctypes.c_uint8 * 32)()
        assert oracle.xo_decrypt_point(kp.secret, ct, out) == 0
        assert kp.decrypt(ct).as_point() == bytes(out)
        assert oracle.xo_ecdlp_decode(bytes(out), ctypes.c_uint32(24)) == amount
        assert oracle.xo_ecdlp_decode(bytes(out), ctypes.c_uint32(16)) == (amount if amount < (1 << 16) else -1)
    assert oracle.xo_ecdlp_decode(kp.pubkey, ctypes.c_uint32(20)) == -1
    assert oracle.xo_ecdlp_decode(bytes([0xff]) * 32, ctypes.c_uint32(20)) == -1
    assert oracle.xo_ecdlp_decode(bytes(32), ctypes.c_uint32(20)) == 0
