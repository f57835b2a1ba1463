// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under xelis-he_amd/ may include, link or call this.
//
// GF(2^255-19) in radix 2^51 (five 64-bit limbs, 128-bit products), the representation
// curve25519-dalek's 64-bit serial backend uses.  The reference (/root/reference/Cargo.toml:10)
// takes its field arithmetic from that crate, which is not vendored; this file restates the
// published algorithms (RFC 7748 section 5 field, RFC 9496 section 4.2 SQRT_RATIO_M1).
// Parity: unpinned against the reference itself (it cannot be built here); pinned by RFC 9496
// vectors and by the big-integer spec in oracle/spec_ristretto.py.
#pragma once
#include <stdint.h>
#include <string.h>

namespace xo {

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

struct fe { u64 v[5]; };

static const u64 MASK51 = ((u64)1 << 51) - 1;

static inline fe fe_zero() { fe r = {{0, 0, 0, 0, 0}}; return r; }
static inline fe fe_one() { fe r = {{1, 0, 0, 0, 0}}; return r; }

// weak reduction: limbs back below 2^51 + small
static inline fe fe_carry(fe a) {
    u64 c;
    c = a.v[0] >> 51; a.v[0] &= MASK51; a.v[1] += c;
    c = a.v[1] >> 51; a.v[1] &= MASK51; a.v[2] += c;
    c = a.v[2] >> 51; a.v[2] &= MASK51; a.v[3] += c;
    c = a.v[3] >> 51; a.v[3] &= MASK51; a.v[4] += c;
    c = a.v[4] >> 51; a.v[4] &= MASK51; a.v[0] += c * 19;
    return a;
}

static inline fe fe_add(const fe &a, const fe &b) {
    fe r;
    for (int i = 0; i < 5; i++) r.v[i] = a.v[i] + b.v[i];
    return fe_carry(r);
}

// a - b, computed as a + 16p - b so limbs never go negative
static inline fe fe_sub(const fe &a, const fe &b) {
    fe r;
    r.v[0] = a.v[0] + 36028797018963664ULL - b.v[0];   // 16*(2^51-19)
    r.v[1] = a.v[1] + 36028797018963952ULL - b.v[1];   // 16*(2^51-1)
    r.v[2] = a.v[2] + 36028797018963952ULL - b.v[2];
    r.v[3] = a.v[3] + 36028797018963952ULL - b.v[3];
    r.v[4] = a.v[4] + 36028797018963952ULL - b.v[4];
    return fe_carry(r);
}

static inline fe fe_neg(const fe &a) { return fe_sub(fe_zero(), a); }

static inline fe fe_mul(const fe &a, const fe &b) {
    const u64 a0 = a.v[0], a1 = a.v[1], a2 = a.v[2], a3 = a.v[3], a4 = a.v[4];
    const u64 b0 = b.v[0], b1 = b.v[1], b2 = b.v[2], b3 = b.v[3], b4 = b.v[4];
    const u64 b1_19 = b1 * 19, b2_19 = b2 * 19, b3_19 = b3 * 19, b4_19 = b4 * 19;
    u128 c0 = (u128)a0 * b0 + (u128)a4 * b1_19 + (u128)a3 * b2_19 + (u128)a2 * b3_19 + (u128)a1 * b4_19;
    u128 c1 = (u128)a1 * b0 + (u128)a0 * b1 + (u128)a4 * b2_19 + (u128)a3 * b3_19 + (u128)a2 * b4_19;
    u128 c2 = (u128)a2 * b0 + (u128)a1 * b1 + (u128)a0 * b2 + (u128)a4 * b3_19 + (u128)a3 * b4_19;
    u128 c3 = (u128)a3 * b0 + (u128)a2 * b1 + (u128)a1 * b2 + (u128)a0 * b3 + (u128)a4 * b4_19;
    u128 c4 = (u128)a4 * b0 + (u128)a3 * b1 + (u128)a2 * b2 + (u128)a1 * b3 + (u128)a0 * b4;
    fe r;
    c1 += (u64)(c0 >> 51); r.v[0] = (u64)c0 & MASK51;
    c2 += (u64)(c1 >> 51); r.v[1] = (u64)c1 & MASK51;
    c3 += (u64)(c2 >> 51); r.v[2] = (u64)c2 & MASK51;
    c4 += (u64)(c3 >> 51); r.v[3] = (u64)c3 & MASK51;
    u64 carry = (u64)(c4 >> 51); r.v[4] = (u64)c4 & MASK51;
    r.v[0] += carry * 19;
    r.v[1] += r.v[0] >> 51; r.v[0] &= MASK51;
    return r;
}

static inline fe fe_sq(const fe &a) { return fe_mul(a, a); }

static inline fe fe_pow2k(fe a, int k) {
    for (int i = 0; i < k; i++) a = fe_sq(a);
    return a;
}

// canonical little-endian 32 bytes
static inline void fe_tobytes(u8 out[32], const fe &a) {
    fe t = fe_carry(fe_carry(a));
    // compute q = floor((t + 19) / 2^255) to decide whether t >= p
    u64 q = (t.v[0] + 19) >> 51;
    q = (t.v[1] + q) >> 51;
    q = (t.v[2] + q) >> 51;
    q = (t.v[3] + q) >> 51;
    q = (t.v[4] + q) >> 51;
    t.v[0] += 19 * q;
    u64 c;
    c = t.v[0] >> 51; t.v[0] &= MASK51; t.v[1] += c;
    c = t.v[1] >> 51; t.v[1] &= MASK51; t.v[2] += c;
    c = t.v[2] >> 51; t.v[2] &= MASK51; t.v[3] += c;
    c = t.v[3] >> 51; t.v[3] &= MASK51; t.v[4] += c;
    t.v[4] &= MASK51;
    u64 w0 = t.v[0] | (t.v[1] << 51);
    u64 w1 = (t.v[1] >> 13) | (t.v[2] << 38);
    u64 w2 = (t.v[2] >> 26) | (t.v[3] << 25);
    u64 w3 = (t.v[3] >> 39) | (t.v[4] << 12);
    memcpy(out, &w0, 8); memcpy(out + 8, &w1, 8); memcpy(out + 16, &w2, 8); memcpy(out + 24, &w3, 8);
}

// loads 255 bits (top bit ignored), value may be >= p (callers check canonicity by re-encoding)
static inline fe fe_frombytes(const u8 in[32]) {
    u64 w0, w1, w2, w3;
    memcpy(&w0, in, 8); memcpy(&w1, in + 8, 8); memcpy(&w2, in + 16, 8); memcpy(&w3, in + 24, 8);
    fe r;
    r.v[0] = w0 & MASK51;
    r.v[1] = ((w0 >> 51) | (w1 << 13)) & MASK51;
    r.v[2] = ((w1 >> 38) | (w2 << 26)) & MASK51;
    r.v[3] = ((w2 >> 25) | (w3 << 39)) & MASK51;
    r.v[4] = (w3 >> 12) & MASK51;
    return r;
}

static inline bool fe_eq(const fe &a, const fe &b) {
    u8 x[32], y[32];
    fe_tobytes(x, a); fe_tobytes(y, b);
    return memcmp(x, y, 32) == 0;
}
static inline bool fe_is_zero(const fe &a) {
    u8 x[32]; fe_tobytes(x, a);
    u8 acc = 0; for (int i = 0; i < 32; i++) acc |= x[i];
    return acc == 0;
}
static inline bool fe_is_negative(const fe &a) { u8 x[32]; fe_tobytes(x, a); return x[0] & 1; }
static inline fe fe_abs(const fe &a) { return fe_is_negative(a) ? fe_neg(a) : a; }
static inline fe fe_small(u64 x) { fe r = fe_zero(); r.v[0] = x & MASK51; r.v[1] = x >> 51; return r; }

// a^(2^252 - 3) = a^((p-5)/8)
static inline fe fe_pow22523(const fe &z) {
    fe t0 = fe_sq(z);                       // 2
    fe t1 = fe_pow2k(t0, 2);                // 8
    t1 = fe_mul(z, t1);                     // 9
    t0 = fe_mul(t0, t1);                    // 11
    t0 = fe_sq(t0);                         // 22
    t0 = fe_mul(t1, t0);                    // 31 = 2^5-1
    t1 = fe_pow2k(t0, 5); t0 = fe_mul(t1, t0);      // 2^10-1
    t1 = fe_pow2k(t0, 10); t1 = fe_mul(t1, t0);     // 2^20-1
    fe t2 = fe_pow2k(t1, 20); t1 = fe_mul(t2, t1);  // 2^40-1
    t1 = fe_pow2k(t1, 10); t0 = fe_mul(t1, t0);     // 2^50-1
    t1 = fe_pow2k(t0, 50); t1 = fe_mul(t1, t0);     // 2^100-1
    t2 = fe_pow2k(t1, 100); t1 = fe_mul(t2, t1);    // 2^200-1
    t1 = fe_pow2k(t1, 50); t0 = fe_mul(t1, t0);     // 2^250-1
    t0 = fe_pow2k(t0, 2);                            // 2^252-4
    return fe_mul(t0, z);                            // 2^252-3
}

// a^(p-2)
static inline fe fe_invert(const fe &z) {
    // z^(2^255-21) = (z^(2^252-3))^8 * z^3
    fe t = fe_pow22523(z);
    t = fe_pow2k(t, 3);
    return fe_mul(t, fe_mul(fe_sq(z), z));
}

struct fe_consts {
    fe d, d2, sqrt_m1, invsqrt_a_minus_d, sqrt_ad_minus_one, one_minus_d_sq, d_minus_one_sq;
};
const fe_consts &fe_constants();

// RFC 9496 4.2: returns was_square and r = sqrt(u/v) (or sqrt(i*u/v)), r non-negative
static inline bool fe_sqrt_ratio_m1(fe &r_out, const fe &u, const fe &v) {
    const fe_consts &K = fe_constants();
    fe v3 = fe_mul(fe_sq(v), v);
    fe v7 = fe_mul(fe_sq(v3), v);
    fe r = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
    fe check = fe_mul(v, fe_sq(r));
    fe neg_u = fe_neg(u);
    bool correct = fe_eq(check, u);
    bool flipped = fe_eq(check, neg_u);
    bool flipped_i = fe_eq(check, fe_mul(neg_u, K.sqrt_m1));
    if (flipped || flipped_i) r = fe_mul(r, K.sqrt_m1);
    r_out = fe_abs(r);
    return correct || flipped;
}

}  // namespace xo
