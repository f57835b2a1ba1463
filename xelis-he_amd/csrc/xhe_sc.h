// Scalars modulo l = 2^252 + 27742317777372353535851937790883648493 for gfx950 (host + device).
//
// HBM format: canonical value, four little-endian 64-bit limbs (32 bytes), as the reference's
// Scalar::to_bytes (/root/reference/src/proofs.rs:213-222).  Register format: nine unsigned 29-bit
// limbs in Montgomery form with R = 2^261.  Because R is 2^9 times larger than l, a Montgomery
// product of two values below 2^257 is below 1.02*l without any conditional subtraction, so products
// chain freely; additions are limb-wise and lazy (one level), canonical reduction happens only when
// a scalar is stored.  Column sums of 29-bit products fit a 64-bit accumulator, so the product and
// the reduction are plain chains of v_mad_u64_u32 (see xhe_fe.h for the measured instruction rates).
//
// Reference semantics restated: curve25519-dalek Scalar arithmetic used at
// /root/reference/src/proofs.rs:162-198,304-347 and Scalar::from_bytes_mod_order_wide at
// /root/reference/src/transcript.rs:46-51, /root/reference/src/elgamal.rs:64.
#pragma once
#include "xhe_fe.h"

namespace xhe {

// Montgomery-form scalar.  Invariant kept by every function here: value < 2^257, limbs < 2^30.
struct sc { u32 v[9]; };
static constexpr u32 SC_MASK29 = (1u << 29) - 1;

#include "xhe_sc_consts.inc"

XHE_HD sc sc_zero() { sc r; for (int i = 0; i < 9; i++) r.v[i] = 0; return r; }
XHE_HD sc sc_one() { return sc_c_R(); }

// limb-wise sum; operands must be normalised (outputs of sc_mul / sc_norm / loads)
XHE_HD sc sc_add_lazy(const sc &a, const sc &b) {
    sc r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// propagate carries so limbs are < 2^29 again (value unchanged, no modular reduction)
XHE_HD sc sc_norm(const sc &a) {
    sc r; u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { u32 t = a.v[i] + c; r.v[i] = t & SC_MASK29; c = t >> 29; }
    r.v[8] = a.v[8] + c;
    return r;
}
XHE_HD sc sc_add(const sc &a, const sc &b) { return sc_norm(sc_add_lazy(a, b)); }

// Montgomery product a*b/R mod l, result < l*(1+2^-4) for operands < 2^258
XHE_HD sc sc_mul(const sc &a, const sc &b) {
    u64 t[18];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        u64 acc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j > 8) continue;
            acc = mad64(a.v[i], b.v[j], acc);
        }
        t[k] = acc;
    }
    t[17] = 0;
    const u32 lv[9] = {SC_L0, SC_L1, SC_L2, SC_L3, SC_L4, SC_L5, SC_L6, SC_L7, SC_L8};
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 m = ((u32)t[i] * SC_NPRIME) & SC_MASK29;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            if (lv[j] == 0) continue;
            t[i + j] = mad64(m, lv[j], t[i + j]);
        }
        t[i + 1] += t[i] >> 29;
    }
    sc r; u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { u64 x = t[9 + i] + c; r.v[i] = (u32)x & SC_MASK29; c = x >> 29; }
    r.v[8] = (u32)(t[17] + c);
    return r;
}
XHE_HD sc sc_sq(const sc &a) { return sc_mul(a, a); }

// raw 256-bit value (eight 32-bit words) -> nine 29-bit limbs, not in Montgomery form
XHE_HD sc sc_limbs_from_words(const u32 w[8]) {
    sc r;
    r.v[0] = w[0] & SC_MASK29;
    r.v[1] = ((w[0] >> 29) | (w[1] << 3)) & SC_MASK29;
    r.v[2] = ((w[1] >> 26) | (w[2] << 6)) & SC_MASK29;
    r.v[3] = ((w[2] >> 23) | (w[3] << 9)) & SC_MASK29;
    r.v[4] = ((w[3] >> 20) | (w[4] << 12)) & SC_MASK29;
    r.v[5] = ((w[4] >> 17) | (w[5] << 15)) & SC_MASK29;
    r.v[6] = ((w[5] >> 14) | (w[6] << 18)) & SC_MASK29;
    r.v[7] = ((w[6] >> 11) | (w[7] << 21)) & SC_MASK29;
    r.v[8] = w[7] >> 8;
    return r;
}
XHE_HD void sc_words_from_limbs(u32 w[8], const sc &a) {   // limbs < 2^29, value < 2^256
    w[0] = a.v[0] | (a.v[1] << 29);
    w[1] = (a.v[1] >> 3) | (a.v[2] << 26);
    w[2] = (a.v[2] >> 6) | (a.v[3] << 23);
    w[3] = (a.v[3] >> 9) | (a.v[4] << 20);
    w[4] = (a.v[4] >> 12) | (a.v[5] << 17);
    w[5] = (a.v[5] >> 15) | (a.v[6] << 14);
    w[6] = (a.v[6] >> 18) | (a.v[7] << 11);
    w[7] = (a.v[7] >> 21) | (a.v[8] << 8);
}

// any 256-bit value -> Montgomery form (value mod l)
XHE_HD sc sc_from_words(const u32 w[8]) { return sc_mul(sc_limbs_from_words(w), sc_c_RR()); }
// 512-bit little-endian value -> Montgomery form (Scalar::from_bytes_mod_order_wide)
XHE_HD sc sc_from_wide_words(const u32 w[16]) {
    sc lo = sc_mul(sc_limbs_from_words(w), sc_c_RR());
    sc hi = sc_mul(sc_limbs_from_words(w + 8), sc_c_RR_2_256());
    return sc_add(lo, hi);
}
XHE_HD sc sc_from_u64(u64 x) {
    u32 w[8] = {(u32)x, (u32)(x >> 32), 0, 0, 0, 0, 0, 0};
    return sc_from_words(w);
}

// Montgomery form -> canonical value in [0, l) as eight 32-bit words
XHE_HD void sc_to_words(u32 w[8], const sc &a) {
    sc one_raw = sc_zero(); one_raw.v[0] = 1;
    sc x = sc_mul(a, one_raw);          // < l*(1+2^-4), limbs < 2^29
    // conditional subtraction of l
    const u32 lv[9] = {SC_L0, SC_L1, SC_L2, SC_L3, SC_L4, SC_L5, SC_L6, SC_L7, SC_L8};
    sc d; u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 t = x.v[i] - lv[i] - borrow;
        borrow = (t >> 31) & 1;          // limbs < 2^29, so a negative difference sets bit 31
        d.v[i] = t & SC_MASK29;
    }
    // borrow out of limb 8: limb 8 holds bits 232.. and is not masked to 29 bits of significance; x < 2^253
    sc r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = borrow ? x.v[i] : d.v[i];
    sc_words_from_limbs(w, r);
}
XHE_HD void sc_tobytes(u8 out[32], const sc &a) {
    u32 w[8]; sc_to_words(w, a);
#pragma unroll
    for (int i = 0; i < 8; i++) { out[4 * i] = (u8)w[i]; out[4 * i + 1] = (u8)(w[i] >> 8); out[4 * i + 2] = (u8)(w[i] >> 16); out[4 * i + 3] = (u8)(w[i] >> 24); }
}
XHE_HD sc sc_frombytes(const u8 in[32]) {
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = (u32)in[4 * i] | ((u32)in[4 * i + 1] << 8) | ((u32)in[4 * i + 2] << 16) | ((u32)in[4 * i + 3] << 24);
    return sc_from_words(w);
}
XHE_HD sc sc_from_wide_bytes(const u8 in[64]) {
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = (u32)in[4 * i] | ((u32)in[4 * i + 1] << 8) | ((u32)in[4 * i + 2] << 16) | ((u32)in[4 * i + 3] << 24);
    return sc_from_wide_words(w);
}
// is the 256-bit value below l ? (canonical-encoding check, Scalar::from_canonical_bytes)
XHE_HD bool sc_words_canonical(const u32 w[8]) {
    const u32 lw[8] = {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0, 0, 0, 0x10000000u};
    for (int i = 7; i >= 0; i--) {
        if (w[i] < lw[i]) return true;
        if (w[i] > lw[i]) return false;
    }
    return false;
}

// -a mod l: one Montgomery product with -1 (valid for any operand a product accepts)
XHE_HD sc sc_neg(const sc &a) { return sc_mul(a, sc_c_MINUS_R()); }
XHE_HD sc sc_sub(const sc &a, const sc &b) { return sc_add(a, sc_neg(b)); }

XHE_HD bool sc_eq(const sc &a, const sc &b) {
    u32 x[8], y[8]; sc_to_words(x, a); sc_to_words(y, b);
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= x[i] ^ y[i];
    return d == 0;
}

// a^(l-2), square-and-multiply over the bits of l-2 (no table, so nothing spills to scratch)
XHE_HD_NOINLINE inline sc sc_invert(const sc &a) {
    const u32 e[8] = {0x5cf5d3ebu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0, 0, 0, 0x10000000u};   // l - 2
    sc r = a;                                  // bit 252
    for (int i = 251; i >= 0; i--) {
        r = sc_sq(r);
        if ((e[i >> 5] >> (i & 31)) & 1) r = sc_mul(r, a);
    }
    return r;
}

}  // namespace xhe
