// Extended twisted-Edwards points (a = -1) and the ristretto255 group for gfx950 (host + device).
//
// HBM formats:
//   compressed point  32 bytes, RFC 9496 encoding (what the reference stores and signs)
//   extended point    4 x 32 bytes: canonical X, Y, Z, T
//   affine Niels      3 x 32 bytes (y+x, y-x, 2d*x*y) padded to 128 bytes so one gather touches one
//                     128-byte line; a ristretto decode yields Z = 1, so every decoded input point of
//                     the multiscalar multiplication is stored this way and bucket additions are
//                     mixed additions (7 field multiplications).
// Restates curve25519-dalek's EdwardsPoint/RistrettoPoint behaviour as the reference uses it at
// /root/reference/src/compressed.rs:17-107, /root/reference/src/elgamal.rs:16-24,280-377 and
// /root/reference/src/proofs.rs:49-67 (is_identity on the multiscalar result).
#pragma once
#include "xhe_fe.h"

namespace xhe {

struct ge { fe X, Y, Z, T; };
struct ge_niels { fe ypx, ymx, xy2d; };

XHE_HD ge ge_identity() { ge r; r.X = fe_zero(); r.Y = fe_one(); r.Z = fe_one(); r.T = fe_zero(); return r; }
XHE_HD ge_niels ge_niels_identity() { ge_niels r; r.ypx = fe_one(); r.ymx = fe_one(); r.xy2d = fe_zero(); return r; }

// affine (Z = 1) extended point -> Niels form
XHE_HD ge_niels ge_to_niels_affine(const ge &p) {
    ge_niels r;
    r.ypx = fe_reduce(fe_add(p.Y, p.X));
    r.ymx = fe_reduce(fe_sub(p.Y, p.X));
    r.xy2d = fe_mul(p.T, fe_c_d2());
    return r;
}

template <int ME, int MF, int MG, int MH>
XHE_HD ge ge_from_completed(const fe_t<ME> &E, const fe_t<MF> &F, const fe_t<MG> &G, const fe_t<MH> &H) {
    ge r;
    r.X = fe_mul(E, F); r.Y = fe_mul(G, H); r.Z = fe_mul(F, G); r.T = fe_mul(E, H);
    return r;
}

// p + q (unified, complete for the prime-order subgroup and its cosets)
XHE_HD ge ge_add(const ge &p, const ge &q) {
    fe A = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    fe B = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    fe C = fe_mul(fe_mul(p.T, fe_c_d2()), q.T);
    fe ZZ = fe_mul(p.Z, q.Z);
    auto D = fe_add(ZZ, ZZ);
    return ge_from_completed(fe_sub(B, A), fe_sub(D, C), fe_add(D, C), fe_add(B, A));
}
XHE_HD ge ge_neg(const ge &p) { ge r; r.X = fe_negate(p.X); r.Y = p.Y; r.Z = p.Z; r.T = fe_negate(p.T); return r; }
XHE_HD ge ge_sub(const ge &p, const ge &q) { return ge_add(p, ge_neg(q)); }

// p + q with q in affine Niels form: 7 multiplications
XHE_HD ge ge_madd(const ge &p, const ge_niels &q) {
    fe A = fe_mul(fe_sub(p.Y, p.X), q.ymx);
    fe B = fe_mul(fe_add(p.Y, p.X), q.ypx);
    fe C = fe_mul(p.T, q.xy2d);
    auto D = fe_add(p.Z, p.Z);
    return ge_from_completed(fe_sub(B, A), fe_sub(D, C), fe_add(D, C), fe_add(B, A));
}
// p - q with q in affine Niels form
XHE_HD ge ge_msub(const ge &p, const ge_niels &q) {
    fe A = fe_mul(fe_sub(p.Y, p.X), q.ypx);
    fe B = fe_mul(fe_add(p.Y, p.X), q.ymx);
    fe C = fe_mul(p.T, q.xy2d);
    auto D = fe_add(p.Z, p.Z);
    return ge_from_completed(fe_sub(B, A), fe_add(D, C), fe_sub(D, C), fe_add(B, A));
}
// sign-selected mixed addition without a branch on the sign
XHE_HD ge ge_madd_signed(const ge &p, const ge_niels &q, bool negative) {
    ge_niels s;
    s.ypx = fe_select(q.ypx, q.ymx, negative);
    s.ymx = fe_select(q.ymx, q.ypx, negative);
    s.xy2d = fe_select(q.xy2d, fe_negate(q.xy2d), negative);
    return ge_madd(p, s);
}

XHE_HD ge ge_double(const ge &p) {
    fe A = fe_sq(p.X), B = fe_sq(p.Y), ZZ = fe_sq(p.Z);
    auto C = fe_add(ZZ, ZZ);
    fe S = fe_sq(fe_add(p.X, p.Y));
    auto H = fe_add(A, B);                 // = -(H of the a = -1 formulas)
    auto E = fe_sub(S, H);
    auto G = fe_sub(B, A);
    fe F = fe_reduce(fe_sub(C, G));        // = -(F of the formulas); all four outputs change sign together
    return ge_from_completed(E, F, G, H);
}

// ristretto255 equality (RFC 9496 4.3.3): X1*Y2 == Y1*X2 or Y1*Y2 == X1*X2
XHE_HD bool ge_ristretto_eq(const ge &p, const ge &q) {
    return fe_eq(fe_mul(p.X, q.Y), fe_mul(p.Y, q.X)) || fe_eq(fe_mul(p.Y, q.Y), fe_mul(p.X, q.X));
}
// the identity coset is the 4-torsion subgroup: X == 0 or Y == 0
XHE_HD bool ge_ristretto_is_identity(const ge &p) { return fe_is_zero(p.X) || fe_is_zero(p.Y); }

// RFC 9496 4.3.1.  `in` as eight little-endian 32-bit words.  On failure the output is the identity.
XHE_HD_NOINLINE inline bool ristretto_decode_words(ge &out, const u32 in[8]) {
    fe s = fe_unpack(in);
    u32 chk[8]; fe_pack(chk, s);
    u32 diff = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) diff |= chk[i] ^ in[i];
    bool ok = (diff == 0) && ((in[0] & 1) == 0);         // canonical and non-negative
    fe ss = fe_sq(s);
    fe u1 = fe_reduce(fe_sub(fe_one(), ss));
    auto u2 = fe_add(fe_one(), ss);
    fe u2s = fe_sq(u2);
    fe du1s = fe_mul(fe_c_d(), fe_sq(u1));
    fe v = fe_reduce(fe_sub(fe_sub(fe_zero(), du1s), u2s));
    fe invsqrt;
    bool was_square = fe_sqrt_ratio_m1(invsqrt, fe_one(), fe_mul(v, u2s));
    fe den_x = fe_mul(invsqrt, u2);
    fe den_y = fe_mul(fe_mul(invsqrt, den_x), v);
    fe x = fe_abs(fe_mul(fe_add(s, s), den_x));
    fe y = fe_mul(u1, den_y);
    fe t = fe_mul(x, y);
    ok = ok && was_square && !fe_is_negative(t) && !fe_is_zero(y);
    ge id = ge_identity();
    out.X = fe_select(id.X, x, ok); out.Y = fe_select(id.Y, y, ok); out.Z = fe_one(); out.T = fe_select(id.T, t, ok);
    return ok;
}
XHE_HD bool ristretto_decode(ge &out, const u8 in[32]) {
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = (u32)in[4 * i] | ((u32)in[4 * i + 1] << 8) | ((u32)in[4 * i + 2] << 16) | ((u32)in[4 * i + 3] << 24);
    return ristretto_decode_words(out, w);
}

// RFC 9496 4.3.2, output as eight 32-bit words
XHE_HD_NOINLINE inline void ristretto_encode_words(u32 out[8], const ge &p) {
    fe u1 = fe_mul(fe_add(p.Z, p.Y), fe_sub(p.Z, p.Y));
    fe u2 = fe_mul(p.X, p.Y);
    fe invsqrt;
    fe_sqrt_ratio_m1(invsqrt, fe_one(), fe_mul(u1, fe_sq(u2)));
    fe den1 = fe_mul(invsqrt, u1), den2 = fe_mul(invsqrt, u2);
    fe z_inv = fe_mul(fe_mul(den1, den2), p.T);
    fe ix0 = fe_mul(p.X, fe_c_sqrt_m1()), iy0 = fe_mul(p.Y, fe_c_sqrt_m1());
    fe enchanted = fe_mul(den1, fe_c_invsqrt_a_minus_d());
    bool rotate = fe_is_negative(fe_mul(p.T, z_inv));
    fe x = fe_select(p.X, iy0, rotate), y = fe_select(p.Y, ix0, rotate), den_inv = fe_select(den2, enchanted, rotate);
    bool neg_y = fe_is_negative(fe_mul(x, z_inv));
    y = fe_select(y, fe_negate(y), neg_y);
    fe s = fe_abs(fe_mul(den_inv, fe_sub(p.Z, y)));
    fe_pack(out, s);
}
XHE_HD void ristretto_encode(u8 out[32], const ge &p) {
    u32 w[8]; ristretto_encode_words(w, p);
#pragma unroll
    for (int i = 0; i < 8; i++) { out[4 * i] = (u8)w[i]; out[4 * i + 1] = (u8)(w[i] >> 8); out[4 * i + 2] = (u8)(w[i] >> 16); out[4 * i + 3] = (u8)(w[i] >> 24); }
}

// RFC 9496 4.3.4 MAP (Elligator 2 on the Jacobi quartic)
XHE_HD_NOINLINE inline ge ristretto_elligator(const fe &t) {
    fe r = fe_mul(fe_c_sqrt_m1(), fe_sq(t));
    fe u = fe_mul(fe_add(r, fe_one()), fe_c_one_minus_d_sq());
    fe rd = fe_mul(r, fe_c_d());
    fe v = fe_mul(fe_sub(fe_sub(fe_zero(), fe_one()), rd), fe_add(r, fe_c_d()));
    fe s;
    bool was_square = fe_sqrt_ratio_m1(s, u, v);
    fe s_prime = fe_negate(fe_abs(fe_mul(s, t)));
    fe minus_one = fe_negate(fe_one());
    s = fe_select(s_prime, s, was_square);
    fe c = fe_select(r, minus_one, was_square);
    fe N = fe_reduce(fe_sub(fe_mul(fe_mul(c, fe_sub(r, fe_one())), fe_c_d_minus_one_sq()), v));
    fe w0 = fe_mul(fe_add(s, s), v);
    fe w1 = fe_mul(N, fe_c_sqrt_ad_minus_one());
    fe ss = fe_sq(s);
    fe w2 = fe_reduce(fe_sub(fe_one(), ss));
    auto w3 = fe_add(fe_one(), ss);
    ge p; p.X = fe_mul(w0, w3); p.Y = fe_mul(w2, w1); p.Z = fe_mul(w1, w3); p.T = fe_mul(w0, w2);
    return p;
}
// RistrettoPoint::from_uniform_bytes
XHE_HD ge ristretto_from_uniform_bytes(const u8 b[64]) {
    return ge_add(ristretto_elligator(fe_frombytes(b)), ristretto_elligator(fe_frombytes(b + 32)));
}

// ---- HBM load/store helpers ------------------------------------------------------------------------
XHE_HD void ge_store(u32 *dst /*32 words*/, const ge &p) { fe_pack(dst, p.X); fe_pack(dst + 8, p.Y); fe_pack(dst + 16, p.Z); fe_pack(dst + 24, p.T); }
XHE_HD ge ge_load(const u32 *src) { ge p; p.X = fe_unpack(src); p.Y = fe_unpack(src + 8); p.Z = fe_unpack(src + 16); p.T = fe_unpack(src + 24); return p; }
XHE_HD void ge_niels_store(u32 *dst /*32 words, last 8 unused*/, const ge_niels &p) { fe_pack(dst, p.ypx); fe_pack(dst + 8, p.ymx); fe_pack(dst + 16, p.xy2d); }
XHE_HD ge_niels ge_niels_load(const u32 *src) { ge_niels p; p.ypx = fe_unpack(src); p.ymx = fe_unpack(src + 8); p.xy2d = fe_unpack(src + 16); return p; }

}  // namespace xhe
