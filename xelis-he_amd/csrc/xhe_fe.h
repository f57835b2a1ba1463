// GF(2^255-19) for gfx950, shared by device kernels and the host-side prover.
//
// HBM format: a field element is its canonical value as four little-endian 64-bit limbs (32 bytes).
// Register format: ten unsigned limbs of alternating 26/25 bits (radix 2^25.5).  The choice follows
// the measured VALU rates of MI355X (tools/ubench_valu.hip): v_mad_u64_u32 issues in ~5.5 cycles per
// wave while every carry-propagating instruction (v_add_co/v_addc_co, 64-bit adds) costs ~4.5-5, so a
// saturated 8x32 schoolbook spends more cycles on carries than on products.  With 25.5-bit limbs a
// column of the product is a plain chain of v_mad_u64_u32 into one 64-bit accumulator, the modular
// fold (2^255 = 19) rides along as a premultiplied operand, and add/sub are ten carry-free 32-bit ops.
//
// Magnitudes are tracked in the type: fe_t<M> has even limbs <= M*2^26 and odd limbs <= M*(2^25+2^18).
// fe_mul/fe_sq static_assert the operand magnitudes that keep every 64-bit column below 2^64, so a
// formula that would overflow does not compile.
//
// Reference behaviour restated: curve25519-dalek FieldElement as used by the reference through
// /root/reference/src/compressed.rs:17-107 and /root/reference/src/elgamal.rs:280-377.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define XHE_HD __host__ __device__ __forceinline__
#define XHE_HD_NOINLINE __host__ __device__
#else
#define XHE_HD inline
#define XHE_HD_NOINLINE
#endif

namespace xhe {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

template <int M>
struct fe_t {
    u32 v[10];
    XHE_HD fe_t() {}
    template <int N>
    XHE_HD fe_t(const fe_t<N> &o) {
        static_assert(N <= M, "magnitude can only be widened implicitly");
#pragma unroll
        for (int i = 0; i < 10; i++) v[i] = o.v[i];
    }
};
typedef fe_t<1> fe;

static constexpr u32 FE_MASK26 = (1u << 26) - 1, FE_MASK25 = (1u << 25) - 1;

XHE_HD fe fe_zero() { fe r; for (int i = 0; i < 10; i++) r.v[i] = 0; return r; }
XHE_HD fe fe_one() { fe r = fe_zero(); r.v[0] = 1; return r; }
XHE_HD fe fe_small(u32 x) { fe r = fe_zero(); r.v[0] = x & FE_MASK26; r.v[1] = x >> 26; return r; }

XHE_HD u64 mad64(u32 a, u32 b, u64 c) { return (u64)a * b + c; }

template <int A, int B>
XHE_HD fe_t<A + B> fe_add(const fe_t<A> &a, const fe_t<B> &b) {
    fe_t<A + B> r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}

// a - b computed as a + k*p - b with k*p spread over the limbs so no limb goes negative.
// k = 2 covers b of magnitude 1, k = 4 magnitude <= 3, k = 8 magnitude <= 7.
template <int B> struct fe_bias { static constexpr int K = (B <= 1) ? 2 : ((B <= 3) ? 4 : 8); static_assert(B <= 7, "subtrahend magnitude too large"); };
template <int A, int B>
XHE_HD fe_t<A + fe_bias<B>::K> fe_sub(const fe_t<A> &a, const fe_t<B> &b) {
    constexpr u32 K = fe_bias<B>::K;
    fe_t<A + fe_bias<B>::K> r;
    r.v[0] = a.v[0] + (K * ((1u << 26) - 19)) - b.v[0];
#pragma unroll
    for (int i = 1; i < 10; i++) r.v[i] = a.v[i] + (K * ((i & 1) ? FE_MASK25 : FE_MASK26)) - b.v[i];
    return r;
}

// carry chain in 32-bit arithmetic: any magnitude <= 32 back to magnitude 1
template <int A>
XHE_HD fe fe_reduce(const fe_t<A> &a) {
    static_assert(A <= 32, "limbs would not fit 32 bits");
    fe r;
    u32 c = a.v[9] >> 25;
    u32 t9 = a.v[9] & FE_MASK25;
    u32 t = a.v[0] + 19 * c;   // c < 2^7 * 32, 19c < 2^17
    c = t >> 26; r.v[0] = t & FE_MASK26;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        t = a.v[i] + c;
        if (i & 1) { c = t >> 25; r.v[i] = t & FE_MASK25; } else { c = t >> 26; r.v[i] = t & FE_MASK26; }
    }
    r.v[9] = t9 + c;           // < 2^25 + 2^7: within magnitude 1
    return r;
}

// carry the ten 64-bit columns into a magnitude-1 element
XHE_HD fe fe_carry_columns(u64 h[10]) {
    fe r;
    u64 c;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        if (i & 1) { c = h[i] >> 25; r.v[i] = (u32)h[i] & FE_MASK25; } else { c = h[i] >> 26; r.v[i] = (u32)h[i] & FE_MASK26; }
        h[i + 1] += c;
    }
    c = h[9] >> 25; r.v[9] = (u32)h[9] & FE_MASK25;
    u64 t = (u64)r.v[0] + 19 * c;          // c < 2^39
    r.v[0] = (u32)t & FE_MASK26;
    r.v[1] += (u32)(t >> 26);              // < 2^25 + 2^18
    return r;
}

// f*g.  g is the operand that gets premultiplied by 19, so it must have magnitude <= 3.
template <int A, int B>
XHE_HD fe fe_mul_ordered(const fe_t<A> &f, const fe_t<B> &g) {
    static_assert(B <= 3, "19*g must fit 32 bits");
    static_assert(A * B <= 24 && A <= 24, "64-bit column would overflow");
    u32 g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { g19[i] = 19 * g.v[i]; f2[i] = 2 * f.v[i]; }
    u64 h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        u64 acc = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            const bool wrap = (i + j) >= 10;
            const bool both_odd = (i & 1) && (j & 1);
            acc = mad64(both_odd ? f2[i] : f.v[i], wrap ? g19[j] : g.v[j], acc);
        }
        h[k] = acc;
    }
    return fe_carry_columns(h);
}
template <int A, int B>
XHE_HD fe fe_mul(const fe_t<A> &f, const fe_t<B> &g) {
    if constexpr (B <= 3) return fe_mul_ordered(f, g);
    else return fe_mul_ordered(g, f);
}

template <int A>
XHE_HD fe fe_sq(const fe_t<A> &f) {
    static_assert(A <= 3, "operand magnitude too large for squaring");
    u32 f2[10], f4[10], f19[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { f2[i] = 2 * f.v[i]; f4[i] = 4 * f.v[i]; f19[i] = 19 * f.v[i]; }
    u64 h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        u64 acc = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            if (i > j) continue;
            const bool wrap = (i + j) >= 10;
            const bool both_odd = (i & 1) && (j & 1);
            if (i == j) acc = mad64(both_odd ? f2[i] : f.v[i], wrap ? f19[j] : f.v[j], acc);
            else acc = mad64(both_odd ? f4[i] : f2[i], wrap ? f19[j] : f.v[j], acc);
        }
        h[k] = acc;
    }
    return fe_carry_columns(h);
}

XHE_HD fe fe_pow2k(fe a, int k) {
    for (int i = 0; i < k; i++) a = fe_sq(a);
    return a;
}

// a^(2^252-3)
XHE_HD_NOINLINE inline fe fe_pow22523(const fe &z) {
    fe t0 = fe_sq(z);
    fe t1 = fe_pow2k(t0, 2);
    t1 = fe_mul(z, t1);
    t0 = fe_mul(t0, t1);
    t0 = fe_sq(t0);
    t0 = fe_mul(t1, t0);
    t1 = fe_pow2k(t0, 5); t0 = fe_mul(t1, t0);
    t1 = fe_pow2k(t0, 10); t1 = fe_mul(t1, t0);
    fe t2 = fe_pow2k(t1, 20); t1 = fe_mul(t2, t1);
    t1 = fe_pow2k(t1, 10); t0 = fe_mul(t1, t0);
    t1 = fe_pow2k(t0, 50); t1 = fe_mul(t1, t0);
    t2 = fe_pow2k(t1, 100); t1 = fe_mul(t2, t1);
    t1 = fe_pow2k(t1, 50); t0 = fe_mul(t1, t0);
    t0 = fe_pow2k(t0, 2);
    return fe_mul(t0, z);
}
XHE_HD fe fe_invert(const fe &z) {
    fe t = fe_pow2k(fe_pow22523(z), 3);
    return fe_mul(t, fe_mul(fe_sq(z), z));
}

// canonical value as eight 32-bit words (= four little-endian 64-bit limbs)
template <int A>
XHE_HD void fe_pack(u32 w[8], const fe_t<A> &a) {
    fe h = fe_reduce(a);
    // q = 1 iff h >= p
    u32 q = (h.v[0] + 19) >> 26;
#pragma unroll
    for (int i = 1; i < 10; i++) q = (h.v[i] + q) >> ((i & 1) ? 25 : 26);
    u32 t = h.v[0] + 19 * q, c;
    c = t >> 26; h.v[0] = t & FE_MASK26;
#pragma unroll
    for (int i = 1; i < 10; i++) {
        t = h.v[i] + c;
        if (i & 1) { c = t >> 25; h.v[i] = t & FE_MASK25; } else { c = t >> 26; h.v[i] = t & FE_MASK26; }
    }
    // limb bit offsets: 0 26 51 77 102 128 153 179 204 230
    w[0] = h.v[0] | (h.v[1] << 26);
    w[1] = (h.v[1] >> 6) | (h.v[2] << 19);
    w[2] = (h.v[2] >> 13) | (h.v[3] << 13);
    w[3] = (h.v[3] >> 19) | (h.v[4] << 6);
    w[4] = h.v[5] | (h.v[6] << 25);
    w[5] = (h.v[6] >> 7) | (h.v[7] << 19);
    w[6] = (h.v[7] >> 13) | (h.v[8] << 12);
    w[7] = (h.v[8] >> 20) | (h.v[9] << 6);
}
// loads 255 bits (bit 255 ignored); the value may be >= p
XHE_HD fe fe_unpack(const u32 w[8]) {
    fe r;
    r.v[0] = w[0] & FE_MASK26;
    r.v[1] = ((w[0] >> 26) | (w[1] << 6)) & FE_MASK25;
    r.v[2] = ((w[1] >> 19) | (w[2] << 13)) & FE_MASK26;
    r.v[3] = ((w[2] >> 13) | (w[3] << 19)) & FE_MASK25;
    r.v[4] = (w[3] >> 6) & FE_MASK26;
    r.v[5] = w[4] & FE_MASK25;
    r.v[6] = ((w[4] >> 25) | (w[5] << 7)) & FE_MASK26;
    r.v[7] = ((w[5] >> 19) | (w[6] << 13)) & FE_MASK25;
    r.v[8] = ((w[6] >> 12) | (w[7] << 20)) & FE_MASK26;
    r.v[9] = (w[7] >> 6) & FE_MASK25;
    return r;
}
XHE_HD void fe_tobytes(u8 out[32], const fe &a) {
    u32 w[8]; fe_pack(w, a);
#pragma unroll
    for (int i = 0; i < 8; i++) { out[4 * i] = (u8)w[i]; out[4 * i + 1] = (u8)(w[i] >> 8); out[4 * i + 2] = (u8)(w[i] >> 16); out[4 * i + 3] = (u8)(w[i] >> 24); }
}
XHE_HD fe fe_frombytes(const u8 in[32]) {
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = (u32)in[4 * i] | ((u32)in[4 * i + 1] << 8) | ((u32)in[4 * i + 2] << 16) | ((u32)in[4 * i + 3] << 24);
    return fe_unpack(w);
}

template <int A, int B>
XHE_HD bool fe_eq(const fe_t<A> &a, const fe_t<B> &b) {
    u32 x[8], y[8]; fe_pack(x, a); fe_pack(y, b);
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= x[i] ^ y[i];
    return d == 0;
}
template <int A> XHE_HD bool fe_is_zero(const fe_t<A> &a) {
    u32 x[8]; fe_pack(x, a);
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= x[i];
    return d == 0;
}
template <int A> XHE_HD bool fe_is_negative(const fe_t<A> &a) { u32 x[8]; fe_pack(x, a); return x[0] & 1; }

// select b when flag, else a (no branch)
XHE_HD fe fe_select(const fe &a, const fe &b, bool flag) {
    fe r; u32 m = flag ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] ^ (m & (a.v[i] ^ b.v[i]));
    return r;
}
XHE_HD fe fe_negate(const fe &a) { return fe_reduce(fe_sub(fe_zero(), a)); }
XHE_HD fe fe_abs(const fe &a) { return fe_select(a, fe_negate(a), fe_is_negative(a)); }

// ---- curve constants as limb literals (generated by tools/gen_consts.py) ----
#include "xhe_fe_consts.inc"

// RFC 9496 4.2 SQRT_RATIO_M1
XHE_HD_NOINLINE inline bool fe_sqrt_ratio_m1(fe &r_out, const fe &u, const fe &v) {
    const fe sqrt_m1 = fe_c_sqrt_m1();
    fe v3 = fe_mul(fe_sq(v), v);
    fe v7 = fe_mul(fe_sq(v3), v);
    fe r = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
    fe check = fe_mul(v, fe_sq(r));
    fe neg_u = fe_negate(u);
    bool correct = fe_eq(check, u);
    bool flipped = fe_eq(check, neg_u);
    bool flipped_i = fe_eq(check, fe_mul(neg_u, sqrt_m1));
    fe r_prime = fe_mul(r, sqrt_m1);
    r = fe_select(r, r_prime, flipped || flipped_i);
    r_out = fe_abs(r);
    return correct || flipped;
}

}  // namespace xhe
