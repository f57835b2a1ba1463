// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under xelis-he_amd/ may include, link or call this.
//
// Extended twisted-Edwards points (a = -1) on Curve25519 and the ristretto255 group: decode,
// encode, one-way map, equality, scalar multiplication and variable-time multiscalar
// multiplication.  Restates the curve25519-dalek behaviour the reference calls at
// /root/reference/src/compressed.rs:17-107 (decode/encode), /root/reference/src/elgamal.rs:16-24
// (from_uniform_bytes), /root/reference/src/proofs.rs:49-67 (vartime_multiscalar_mul +
// is_identity).  Algorithms: RFC 9496 sections 4.3.1-4.3.4, Hisil et al. extended coordinates.
#pragma once
#include <vector>
#include "xo_fe.h"
#include "xo_sc.h"

namespace xo {

struct ge { fe X, Y, Z, T; };

static inline ge ge_identity() { ge r = {fe_zero(), fe_one(), fe_one(), fe_zero()}; return r; }

static inline ge ge_add(const ge &p, const ge &q) {
    const fe_consts &K = fe_constants();
    fe A = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    fe B = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    fe C = fe_mul(fe_mul(p.T, K.d2), q.T);
    fe D = fe_mul(fe_add(p.Z, p.Z), q.Z);
    fe E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    ge r = {fe_mul(E, F), fe_mul(G, H), fe_mul(F, G), fe_mul(E, H)};
    return r;
}
static inline ge ge_neg(const ge &p) { ge r = {fe_neg(p.X), p.Y, p.Z, fe_neg(p.T)}; return r; }
static inline ge ge_sub(const ge &p, const ge &q) { return ge_add(p, ge_neg(q)); }

static inline ge ge_double(const ge &p) {
    fe A = fe_sq(p.X), B = fe_sq(p.Y), C = fe_add(fe_sq(p.Z), fe_sq(p.Z));
    fe D = fe_neg(A);                                   // a = -1
    fe E = fe_sub(fe_sub(fe_sq(fe_add(p.X, p.Y)), A), B);
    fe G = fe_add(D, B), F = fe_sub(G, C), H = fe_sub(D, B);
    ge r = {fe_mul(E, F), fe_mul(G, H), fe_mul(F, G), fe_mul(E, H)};
    return r;
}

// Ristretto equality: X1*Y2 == Y1*X2 or Y1*Y2 == X1*X2
static inline bool ge_ristretto_eq(const ge &p, const ge &q) {
    return fe_eq(fe_mul(p.X, q.Y), fe_mul(p.Y, q.X)) || fe_eq(fe_mul(p.Y, q.Y), fe_mul(p.X, q.X));
}
static inline bool ge_ristretto_is_identity(const ge &p) { return ge_ristretto_eq(p, ge_identity()); }

// RFC 9496 4.3.1
static inline bool ristretto_decode(ge &out, const u8 in[32]) {
    const fe_consts &K = fe_constants();
    fe s = fe_frombytes(in);
    u8 chk[32]; fe_tobytes(chk, s);
    if (memcmp(chk, in, 32) != 0) return false;   // non-canonical (>= p or bit 255 set)
    if (in[0] & 1) return false;                  // negative
    fe ss = fe_sq(s);
    fe u1 = fe_sub(fe_one(), ss), u2 = fe_add(fe_one(), ss);
    fe u2s = fe_sq(u2);
    fe v = fe_sub(fe_neg(fe_mul(K.d, fe_sq(u1))), u2s);
    fe invsqrt;
    bool ok = fe_sqrt_ratio_m1(invsqrt, fe_one(), fe_mul(v, u2s));
    fe den_x = fe_mul(invsqrt, u2);
    fe den_y = fe_mul(fe_mul(invsqrt, den_x), v);
    fe x = fe_abs(fe_mul(fe_add(s, s), den_x));
    fe y = fe_mul(u1, den_y);
    fe t = fe_mul(x, y);
    if (!ok || fe_is_negative(t) || fe_is_zero(y)) return false;
    out.X = x; out.Y = y; out.Z = fe_one(); out.T = t;
    return true;
}

// RFC 9496 4.3.2
static inline void ristretto_encode(u8 out[32], const ge &p) {
    const fe_consts &K = fe_constants();
    fe u1 = fe_mul(fe_add(p.Z, p.Y), fe_sub(p.Z, p.Y));
    fe u2 = fe_mul(p.X, p.Y);
    fe invsqrt;
    fe_sqrt_ratio_m1(invsqrt, fe_one(), fe_mul(u1, fe_sq(u2)));
    fe den1 = fe_mul(invsqrt, u1), den2 = fe_mul(invsqrt, u2);
    fe z_inv = fe_mul(fe_mul(den1, den2), p.T);
    fe ix0 = fe_mul(p.X, K.sqrt_m1), iy0 = fe_mul(p.Y, K.sqrt_m1);
    fe enchanted = fe_mul(den1, K.invsqrt_a_minus_d);
    bool rotate = fe_is_negative(fe_mul(p.T, z_inv));
    fe x = rotate ? iy0 : p.X, y = rotate ? ix0 : p.Y, den_inv = rotate ? enchanted : den2;
    if (fe_is_negative(fe_mul(x, z_inv))) y = fe_neg(y);
    fe s = fe_abs(fe_mul(den_inv, fe_sub(p.Z, y)));
    fe_tobytes(out, s);
}

// RFC 9496 4.3.4 MAP
static inline ge ristretto_elligator(const fe &t) {
    const fe_consts &K = fe_constants();
    fe r = fe_mul(K.sqrt_m1, fe_sq(t));
    fe u = fe_mul(fe_add(r, fe_one()), K.one_minus_d_sq);
    fe v = fe_mul(fe_sub(fe_neg(fe_one()), fe_mul(r, K.d)), fe_add(r, K.d));
    fe s;
    bool was_square = fe_sqrt_ratio_m1(s, u, v);
    fe s_prime = fe_neg(fe_abs(fe_mul(s, t)));
    fe c = fe_neg(fe_one());
    if (!was_square) { s = s_prime; c = r; }
    fe N = fe_sub(fe_mul(fe_mul(c, fe_sub(r, fe_one())), K.d_minus_one_sq), v);
    fe w0 = fe_mul(fe_add(s, s), v);
    fe w1 = fe_mul(N, K.sqrt_ad_minus_one);
    fe ss = fe_sq(s);
    fe w2 = fe_sub(fe_one(), ss), w3 = fe_add(fe_one(), ss);
    ge p = {fe_mul(w0, w3), fe_mul(w2, w1), fe_mul(w1, w3), fe_mul(w0, w2)};
    return p;
}
// RistrettoPoint::from_uniform_bytes
static inline ge ristretto_from_uniform_bytes(const u8 b[64]) {
    u8 h0[32], h1[32];
    memcpy(h0, b, 32); memcpy(h1, b + 32, 32);
    h0[31] &= 0x7f; h1[31] &= 0x7f;
    // fe_frombytes does not reduce; fe arithmetic accepts values in [p, 2^255)
    return ge_add(ristretto_elligator(fe_frombytes(h0)), ristretto_elligator(fe_frombytes(h1)));
}

const ge &ge_basepoint();      // ristretto255 generator (decode of RFC 9496 B)
const ge &ge_pedersen_h();     // H = from_uniform_bytes(SHA3-512(enc(B)))  (elgamal.rs:16-24)

// Variable-base scalar multiplication, 4-bit fixed window (variable time; the oracle does not
// need the reference's constant-time ladder, the result is the same group element).
static inline ge ge_scalarmul(const sc &k, const ge &p) {
    ge tbl[16];
    tbl[0] = ge_identity(); tbl[1] = p;
    for (int i = 2; i < 16; i++) tbl[i] = (i & 1) ? ge_add(tbl[i - 1], p) : ge_double(tbl[i / 2]);
    ge r = ge_identity();
    for (int i = 63; i >= 0; i--) {
        if (i != 63) { r = ge_double(r); r = ge_double(r); r = ge_double(r); r = ge_double(r); }
        unsigned nib = (unsigned)(k.v[i >> 4] >> ((i & 15) * 4)) & 15;
        if (nib) r = ge_add(r, tbl[nib]);
    }
    return r;
}

// signed radix-2^w digits of a scalar < 2^255 (dalek Scalar::as_radix_2w): digits in [-2^(w-1), 2^(w-1))
static inline std::vector<int> sc_radix_2w(const sc &k, int w) {
    int digits = (256 + w - 1) / w;
    std::vector<int> out(digits + 1, 0);
    int carry = 0;
    for (int i = 0; i < digits; i++) {
        int bit = i * w, limb = bit >> 6, off = bit & 63;
        u64 raw = k.v[limb] >> off;
        if (off + w > 64 && limb < 3) raw |= k.v[limb + 1] << (64 - off);
        int coef = carry + (int)(raw & (((u64)1 << w) - 1));
        carry = (coef + (1 << (w - 1))) >> w;
        out[i] = coef - (carry << w);
    }
    out[digits] = carry;
    return out;
}

// Pippenger bucket method (dalek's choice for >= 190 terms; w = 6/7/8 by size)
static inline ge ge_msm_pippenger(const std::vector<sc> &scalars, const std::vector<ge> &points) {
    size_t n = scalars.size();
    int w = n < 500 ? 6 : (n < 800 ? 7 : 8);
    size_t nb = (size_t)1 << (w - 1);
    std::vector<std::vector<int>> digs(n);
    for (size_t i = 0; i < n; i++) digs[i] = sc_radix_2w(scalars[i], w);
    int ndig = (int)digs.empty() ? 0 : (int)digs[0].size();
    ge total = ge_identity();
    std::vector<ge> buckets(nb);
    for (int d = ndig - 1; d >= 0; d--) {
        for (int k = 0; k < w; k++) total = ge_double(total);
        for (size_t b = 0; b < nb; b++) buckets[b] = ge_identity();
        for (size_t i = 0; i < n; i++) {
            int dg = digs[i][d];
            if (dg > 0) buckets[dg - 1] = ge_add(buckets[dg - 1], points[i]);
            else if (dg < 0) buckets[-dg - 1] = ge_sub(buckets[-dg - 1], points[i]);
        }
        ge run = ge_identity(), sum = ge_identity();
        for (size_t b = nb; b-- > 0;) { run = ge_add(run, buckets[b]); sum = ge_add(sum, run); }
        total = ge_add(total, sum);
    }
    return total;
}

// Straus interleaved width-5 NAF (dalek's choice for < 190 terms)
static inline std::vector<int8_t> sc_naf(const sc &k, int w) {
    std::vector<int8_t> naf(257, 0);
    u64 x[5] = {k.v[0], k.v[1], k.v[2], k.v[3], 0};
    u64 width = (u64)1 << w, mask = width - 1;
    int pos = 0; u64 carry = 0;
    while (pos < 256) {
        int idx = pos >> 6, off = pos & 63;
        u64 bits = x[idx] >> off;
        if (off + w > 64) bits |= x[idx + 1] << (64 - off);
        u64 window = carry + (bits & mask);
        if ((window & 1) == 0) { pos += 1; continue; }
        if (window < width / 2) { carry = 0; naf[pos] = (int8_t)window; }
        else { carry = 1; naf[pos] = (int8_t)((int64_t)window - (int64_t)width); }
        pos += w;
    }
    naf[256] = (int8_t)carry;
    return naf;
}
static inline ge ge_msm_straus(const std::vector<sc> &scalars, const std::vector<ge> &points) {
    size_t n = scalars.size();
    std::vector<std::vector<int8_t>> nafs(n);
    std::vector<std::vector<ge>> tbl(n);   // odd multiples 1P,3P,...,15P
    for (size_t i = 0; i < n; i++) {
        nafs[i] = sc_naf(scalars[i], 5);
        ge p2 = ge_double(points[i]);
        tbl[i].resize(8);
        tbl[i][0] = points[i];
        for (int j = 1; j < 8; j++) tbl[i][j] = ge_add(tbl[i][j - 1], p2);
    }
    ge r = ge_identity();
    for (int pos = 256; pos >= 0; pos--) {
        r = ge_double(r);
        for (size_t i = 0; i < n; i++) {
            int d = nafs[i][pos];
            if (d > 0) r = ge_add(r, tbl[i][d >> 1]);
            else if (d < 0) r = ge_sub(r, tbl[i][(-d) >> 1]);
        }
    }
    return r;
}
// RistrettoPoint::vartime_multiscalar_mul
static inline ge ge_msm_vartime(const std::vector<sc> &scalars, const std::vector<ge> &points) {
    if (scalars.size() < 190) return ge_msm_straus(scalars, points);
    return ge_msm_pippenger(scalars, points);
}

}  // namespace xo
