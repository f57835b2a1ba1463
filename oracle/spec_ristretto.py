"""Big-integer restatement of ristretto255 (RFC 9496) and the scalar field, for cross-checking
the C++ oracle and the HIP path on small cases.  TEST INFRASTRUCTURE ONLY: nothing in the product
package may import this file.

Follows the published algorithms the reference's dependency curve25519-dalek implements
(reference call sites: /root/reference/src/compressed.rs:17-107 decode/encode,
/root/reference/src/elgamal.rs:16-24 one-way map for H).  Constants are derived, not transcribed.
"""
import hashlib

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)


def _is_neg(x):
    return (x % P) & 1


def _abs(x):
    x %= P
    return P - x if x & 1 else x


def sqrt_ratio_m1(u, v):
    """RFC 9496 section 4.2 SQRT_RATIO_M1: (was_square, sqrt(u/v) or sqrt(i*u/v)), non-negative."""
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (-u) % P
    flipped_i = check == (-u * SQRT_M1) % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    r = _abs(r)
    return (correct or flipped), r


def _sqrt(x):
    ok, r = sqrt_ratio_m1(x, 1)
    assert ok
    return r


# derived ristretto constants (signs fixed as in RFC 9496 section 4.1)
INVSQRT_A_MINUS_D = sqrt_ratio_m1(1, (-1 - D) % P)[1]
SQRT_AD_MINUS_ONE = P - _sqrt((-D - 1) % P)      # the RFC's value is the odd ("negative") root
ONE_MINUS_D_SQ = (1 - D * D) % P
D_MINUS_ONE_SQ = (D - 1) * (D - 1) % P

IDENT = (0, 1, 1, 0)


def add(p, q):
    """Extended twisted Edwards addition, a = -1 (unified)."""
    X1, Y1, Z1, T1 = p
    X2, Y2, Z2, T2 = q
    A = (Y1 - X1) * (Y2 - X2) % P
    B = (Y1 + X1) * (Y2 + X2) % P
    C = T1 * 2 * D % P * T2 % P
    Dd = Z1 * 2 * Z2 % P
    E, F, G, H = B - A, Dd - C, Dd + C, B + A
    return (E * F % P, G * H % P, F * G % P, E * H % P)


def neg(p):
    X, Y, Z, T = p
    return ((-X) % P, Y, Z, (-T) % P)


def mul(k, p):
    k %= L
    r = IDENT
    while k:
        if k & 1:
            r = add(r, p)
        p = add(p, p)
        k >>= 1
    return r


def decode(b):
    """RFC 9496 4.3.1.  Returns a point or None."""
    s = int.from_bytes(b, "little")
    if s >= P or (s & 1):
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2s = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2s) % P
    ok, invsqrt = sqrt_ratio_m1(1, v * u2s % P)
    den_x = invsqrt * u2 % P
    den_y = invsqrt * den_x % P * v % P
    x = _abs(2 * s * den_x % P)
    y = u1 * den_y % P
    t = x * y % P
    if (not ok) or _is_neg(t) or y == 0:
        return None
    return (x, y, 1, t)


def encode(p):
    """RFC 9496 4.3.2."""
    x0, y0, z0, t0 = p
    u1 = (z0 + y0) * (z0 - y0) % P
    u2 = x0 * y0 % P
    _, invsqrt = sqrt_ratio_m1(1, u1 * u2 % P * u2 % P)
    den1 = invsqrt * u1 % P
    den2 = invsqrt * u2 % P
    z_inv = den1 * den2 % P * t0 % P
    ix0 = x0 * SQRT_M1 % P
    iy0 = y0 * SQRT_M1 % P
    enchanted = den1 * INVSQRT_A_MINUS_D % P
    rotate = _is_neg(t0 * z_inv % P)
    if rotate:
        x, y, den_inv = iy0, ix0, enchanted
    else:
        x, y, den_inv = x0, y0, den2
    if _is_neg(x * z_inv % P):
        y = (-y) % P
    s = _abs(den_inv * ((z0 - y) % P) % P)
    return s.to_bytes(32, "little")


def elligator(t):
    """RFC 9496 4.3.4 MAP."""
    r = SQRT_M1 * t % P * t % P
    u = (r + 1) * ONE_MINUS_D_SQ % P
    v = (-1 - r * D) % P * ((r + D) % P) % P
    was_square, s = sqrt_ratio_m1(u, v)
    s_prime = (-_abs(s * t % P)) % P
    if not was_square:
        s = s_prime
        c = r
    else:
        c = P - 1
    n = (c * ((r - 1) % P) % P * D_MINUS_ONE_SQ - v) % P
    w0 = 2 * s * v % P
    w1 = n * SQRT_AD_MINUS_ONE % P
    w2 = (1 - s * s) % P
    w3 = (1 + s * s) % P
    return (w0 * w3 % P, w2 * w1 % P, w1 * w3 % P, w0 * w2 % P)


def from_uniform_bytes(b):
    assert len(b) == 64
    r0 = int.from_bytes(b[:32], "little") & ((1 << 255) - 1)
    r1 = int.from_bytes(b[32:], "little") & ((1 << 255) - 1)
    return add(elligator(r0 % P), elligator(r1 % P))


def eq(p, q):
    X1, Y1, _, _ = p
    X2, Y2, _, _ = q
    return (X1 * Y2 - Y1 * X2) % P == 0 or (Y1 * Y2 - X1 * X2) % P == 0


B_ENC = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
BASE = decode(B_ENC)
H_POINT = from_uniform_bytes(hashlib.sha3_512(B_ENC).digest())


def msm(scalars, points):
    acc = IDENT
    for k, p in zip(scalars, points):
        acc = add(acc, mul(k, p))
    return acc


if __name__ == "__main__":
    print("B*2", encode(mul(2, BASE)).hex())
    print("B*3", encode(mul(3, BASE)).hex())
    print("H  ", encode(H_POINT).hex())
    print("l*B", encode(mul(L - 1, BASE)).hex(), encode(add(mul(L - 1, BASE), BASE)).hex())
