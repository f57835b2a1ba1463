// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under xelis-he_amd/ may include, link or call this.
//
// Scalars modulo the ristretto255 group order l = 2^252 + 27742317777372353535851937790883648493,
// four 64-bit limbs, Montgomery multiplication (R = 2^256) with results kept in normal form.
// Restates curve25519-dalek's Scalar semantics used at /root/reference/src/proofs.rs:162-198,
// /root/reference/src/transcript.rs:46-51 (from_bytes_mod_order_wide) and
// /root/reference/src/elgamal.rs:64.
#pragma once
#include "xo_fe.h"

namespace xo {

struct sc { u64 v[4]; };

static const sc SC_L = {{0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0x0000000000000000ULL, 0x1000000000000000ULL}};
// -l^{-1} mod 2^64 by Newton iteration on the low limb
static inline u64 sc_compute_lfactor() {
    u64 inv = 1;
    for (int i = 0; i < 7; i++) inv *= 2 - SC_L.v[0] * inv;
    return (u64)0 - inv;
}
static const u64 SC_LFACTOR = sc_compute_lfactor();

static inline sc sc_zero() { sc r = {{0, 0, 0, 0}}; return r; }
static inline sc sc_one() { sc r = {{1, 0, 0, 0}}; return r; }
static inline sc sc_from_u64(u64 x) { sc r = {{x, 0, 0, 0}}; return r; }

static inline bool sc_eq(const sc &a, const sc &b) { return memcmp(a.v, b.v, 32) == 0; }
static inline bool sc_is_zero(const sc &a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }

// a >= b ?
static inline bool sc_geq(const sc &a, const sc &b) {
    for (int i = 3; i >= 0; i--) {
        if (a.v[i] > b.v[i]) return true;
        if (a.v[i] < b.v[i]) return false;
    }
    return true;
}
static inline sc sc_raw_sub(const sc &a, const sc &b, u64 *borrow_out) {
    sc r; u64 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a.v[i] - b.v[i] - borrow;
        r.v[i] = (u64)t; borrow = (u64)(t >> 64) & 1;
    }
    if (borrow_out) *borrow_out = borrow;
    return r;
}
static inline sc sc_raw_add(const sc &a, const sc &b, u64 *carry_out) {
    sc r; u64 carry = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a.v[i] + b.v[i] + carry;
        r.v[i] = (u64)t; carry = (u64)(t >> 64);
    }
    if (carry_out) *carry_out = carry;
    return r;
}
static inline sc sc_add(const sc &a, const sc &b) {   // inputs < l
    sc r = sc_raw_add(a, b, nullptr);                 // < 2l < 2^254, no overflow
    if (sc_geq(r, SC_L)) r = sc_raw_sub(r, SC_L, nullptr);
    return r;
}
static inline sc sc_sub(const sc &a, const sc &b) {
    u64 borrow; sc r = sc_raw_sub(a, b, &borrow);
    if (borrow) r = sc_raw_add(r, SC_L, nullptr);
    return r;
}
static inline sc sc_neg(const sc &a) { return sc_sub(sc_zero(), a); }

// Montgomery product a*b/R mod l (CIOS); inputs < 2^256 with a*b < l*R
static inline sc sc_montmul(const sc &a, const sc &b) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u64 carry = 0;
        for (int j = 0; j < 4; j++) {
            u128 p = (u128)a.v[j] * b.v[i] + t[j] + carry;
            t[j] = (u64)p; carry = (u64)(p >> 64);
        }
        u128 s = (u128)t[4] + carry; t[4] = (u64)s; t[5] = (u64)(s >> 64);
        u64 m = t[0] * SC_LFACTOR;
        u128 p = (u128)m * SC_L.v[0] + t[0]; carry = (u64)(p >> 64);
        for (int j = 1; j < 4; j++) {
            p = (u128)m * SC_L.v[j] + t[j] + carry;
            t[j - 1] = (u64)p; carry = (u64)(p >> 64);
        }
        s = (u128)t[4] + carry; t[3] = (u64)s; t[4] = t[5] + (u64)(s >> 64);
    }
    sc r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || sc_geq(r, SC_L)) r = sc_raw_sub(r, SC_L, nullptr);
    return r;
}

struct sc_consts { sc R, RR, RRR; };   // R = 2^256 mod l, RR = R^2, RRR = R^3 (all mod l)
const sc_consts &sc_constants();

static inline sc sc_mul(const sc &a, const sc &b) {
    return sc_montmul(sc_montmul(a, b), sc_constants().RR);
}
static inline sc sc_sq(const sc &a) { return sc_mul(a, a); }

static inline sc sc_from_bytes_raw(const u8 b[32]) { sc r; memcpy(r.v, b, 32); return r; }
static inline void sc_tobytes(u8 out[32], const sc &a) { memcpy(out, a.v, 32); }
static inline bool sc_is_canonical_bytes(const u8 b[32]) { sc r = sc_from_bytes_raw(b); return !sc_geq(r, SC_L); }

// 256-bit value mod l
static inline sc sc_from_bytes_mod_order(const u8 b[32]) {
    sc x = sc_from_bytes_raw(b);
    // x*R/R : montmul(x, R) = x mod l   (x < 2^256, R < l so x*R < l*2^256)
    return sc_montmul(x, sc_constants().R);
}
// 512-bit little-endian value mod l (Scalar::from_bytes_mod_order_wide)
static inline sc sc_from_bytes_mod_order_wide(const u8 b[64]) {
    sc lo = sc_from_bytes_raw(b), hi = sc_from_bytes_raw(b + 32);
    const sc_consts &K = sc_constants();
    // lo + hi*2^256 = montmul(lo, R) + montmul(hi, R^2)
    return sc_add(sc_montmul(lo, K.R), sc_montmul(hi, K.RR));
}

static inline sc sc_pow(const sc &a, const sc &e) {
    sc r = sc_one();
    for (int i = 255; i >= 0; i--) {
        r = sc_sq(r);
        if ((e.v[i >> 6] >> (i & 63)) & 1) r = sc_mul(r, a);
    }
    return r;
}
static inline sc sc_invert(const sc &a) {
    sc e = sc_raw_sub(SC_L, sc_from_u64(2), nullptr);
    return sc_pow(a, e);
}

}  // namespace xo
