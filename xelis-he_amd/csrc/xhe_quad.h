// Extended twisted-Edwards arithmetic on four cooperating lanes (device only).
//
// A dependent chain of point operations (the Horner stage of the multiscalar multiplication, a Schnorr
// verification, a running sum) is latency: a lone wavefront issues the eight field products of a
// doubling one after the other.  The four products of each half of the a = -1 extended formulas are
// independent, so here the four lanes of a quad (lanes 4q .. 4q+3) share one point: lane r holds
// coordinate r of (X, Y, Z, T), computes one product per half, and the lanes trade limbs through DPP
// quad permutes (register-to-register inside the SIMD, no LDS, no barrier).  A doubling costs one squaring
// plus one multiplication per lane instead of four plus four.
//
// All four lanes of a quad must be active whenever these functions run (a DPP read of a disabled lane
// leaves the destination unchanged); callers keep control flow uniform per quad.
//
// Same formulas as ge_double / ge_add / ge_madd of xhe_ge.h, which restate curve25519-dalek's
// EdwardsPoint arithmetic as used through /root/reference/src/elgamal.rs:280-377 and
// /root/reference/src/proofs.rs:49-67.
#pragma once
#include "xhe_ge.h"

namespace xhe {

struct qd_point { fe c; };                   // this lane's coordinate: role 0..3 = X, Y, Z, T

#define XHE_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

// every lane of the quad reads the lane named for its position
template <int CTRL>
__device__ __forceinline__ u32 qd_pull(u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false); }
// as qd_pull where keep is all-ones, zero where it is zero (the DPP bank mask works on groups of four lanes, not
// on positions inside a quad, so the positions are masked afterwards)
template <int CTRL>
__device__ __forceinline__ u32 qd_pull_masked(u32 v, u32 keep) { return qd_pull<CTRL>(v) & keep; }

template <int K>
__device__ __forceinline__ u32 qd_bias_limb(int i) { return i == 0 ? K * ((1u << 26) - 19) : K * ((i & 1) ? FE_MASK25 : FE_MASK26); }

__device__ __forceinline__ u32 qd_role() { return threadIdx.x & 3; }

// (E, F, G, H) spread over the quad -> (E*F, G*H, F*G, E*H) = (X, Y, Z, T)
__device__ __forceinline__ qd_point qd_from_completed(const fe &e) {
    fe o1, o2;
#pragma unroll
    for (int i = 0; i < 10; i++) { o1.v[i] = qd_pull<XHE_QP(0, 2, 1, 0)>(e.v[i]); o2.v[i] = qd_pull<XHE_QP(1, 3, 2, 3)>(e.v[i]); }
    qd_point r; r.c = fe_mul(o1, o2);
    return r;
}

// 2p
__device__ __forceinline__ qd_point qd_double(const qd_point &p, u32 role) {
    // squared: X, Y, Z, X + Y
    const u32 only3 = role == 3 ? 0xffffffffu : 0u, only01 = role < 2 ? 0xffffffffu : 0u;
    fe_t<2> op;
#pragma unroll
    for (int i = 0; i < 10; i++) op.v[i] = qd_pull<XHE_QP(0, 1, 2, 0)>(p.c.v[i]) + qd_pull_masked<XHE_QP(0, 0, 0, 1)>(p.c.v[i], only3);
    fe s = fe_sq(op);                                  // A, B, ZZ, S
    // E = S - A - B, F = 2 ZZ + A - B, G = B - A, H = A + B   (the signs of ge_double)
    const bool add_a = (role & 1) != 0, add_b = role >= 2;
    const u32 sh = role == 1 ? 1u : 0u;
    fe_t<6> r;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        u32 a = qd_pull<XHE_QP(0, 0, 0, 0)>(s.v[i]), b = qd_pull<XHE_QP(1, 1, 1, 1)>(s.v[i]);
        u32 t = qd_pull_masked<XHE_QP(3, 2, 0, 0)>(s.v[i], only01);
        u32 sa = add_a ? a : qd_bias_limb<2>(i) - a, sb = add_b ? b : qd_bias_limb<2>(i) - b;
        r.v[i] = (t << sh) + sa + sb;
    }
    return qd_from_completed(fe_reduce(r));
}

// this lane's operand of the first half of an addition: (Y - X, Y + X, Z, T)
__device__ __forceinline__ fe_t<3> qd_add_operand(const fe &c, u32 role) {
    const u32 only01 = role < 2 ? 0xffffffffu : 0u;
    fe_t<3> out;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        u32 own = c.v[i];
        u32 other = qd_pull_masked<XHE_QP(1, 0, 0, 0)>(own, only01);    // lane 0 <- Y, lane 1 <- X
        out.v[i] = (role == 0 ? qd_bias_limb<2>(i) - own : own) + other;
    }
    return out;
}
// products (A, B, ZZ, C) spread over the quad -> the sum
__device__ __forceinline__ qd_point qd_add_finish(const fe &prod, u32 role) {
    // E = B - A, F = 2 ZZ - C, G = 2 ZZ + C, H = B + A
    const u32 sh = (role == 1 || role == 2) ? 1u : 0u;
    fe_t<6> r;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        u32 x = qd_pull<XHE_QP(0, 2, 2, 0)>(prod.v[i]) << sh, y = qd_pull<XHE_QP(1, 3, 3, 1)>(prod.v[i]);
        u32 sx = role == 0 ? qd_bias_limb<4>(i) - x : x, sy = role == 1 ? qd_bias_limb<2>(i) - y : y;
        r.v[i] = sx + sy;
    }
    return qd_from_completed(fe_reduce(r));
}
// p + q, both spread over the quad
__device__ __forceinline__ qd_point qd_add(const qd_point &p, const qd_point &q, u32 role) {
    fe_t<3> a = qd_add_operand(p.c, role), b = qd_add_operand(q.c, role);
    fe_t<3> td = fe_mul(a, fe_c_d2());                 // only lane 3 (T1 * 2d) keeps it
    fe_t<3> a2;
#pragma unroll
    for (int i = 0; i < 10; i++) a2.v[i] = role == 3 ? td.v[i] : a.v[i];
    return qd_add_finish(fe_mul(a2, b), role);
}
// p +/- q with q an affine-Niels record in memory (ypx, ymx, xy2d, unused)
__device__ __forceinline__ qd_point qd_madd(const qd_point &p, const u32 *__restrict__ niels, bool negative, u32 role) {
    fe_t<3> a = qd_add_operand(p.c, role);
    // multiplicands: ymx, ypx, 1, xy2d; a negated point swaps the first two and negates the last
    const u32 field = role == 0 ? (negative ? 0u : 1u) : role == 1 ? (negative ? 1u : 0u) : 2u;
    fe m = fe_unpack(niels + 8 * field);
    fe neg = fe_negate(m);
    fe one = fe_one();
    fe b;
#pragma unroll
    for (int i = 0; i < 10; i++) b.v[i] = role == 2 ? one.v[i] : (role == 3 && negative) ? neg.v[i] : m.v[i];
    return qd_add_finish(fe_mul(a, b), role);
}

// memory <-> quad: an XYZT record is four 32-byte coordinates, lane r takes coordinate r
__device__ __forceinline__ qd_point qd_load(const u32 *__restrict__ src, u32 role) { qd_point p; p.c = fe_unpack(src + 8 * role); return p; }
__device__ __forceinline__ void qd_store(u32 *__restrict__ dst, const qd_point &p, u32 role) { fe_pack(dst + 8 * role, p.c); }
__device__ __forceinline__ qd_point qd_identity(u32 role) { qd_point p; p.c = (role == 1 || role == 2) ? fe_one() : fe_zero(); return p; }
__device__ __forceinline__ qd_point qd_from_ge(const ge &g, u32 role) {
    qd_point p;
#pragma unroll
    for (int i = 0; i < 10; i++) p.c.v[i] = role == 0 ? g.X.v[i] : role == 1 ? g.Y.v[i] : role == 2 ? g.Z.v[i] : g.T.v[i];
    return p;
}
// the whole point in every lane of the quad
__device__ __forceinline__ ge qd_gather(const qd_point &p) {
    ge g;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        g.X.v[i] = qd_pull<XHE_QP(0, 0, 0, 0)>(p.c.v[i]); g.Y.v[i] = qd_pull<XHE_QP(1, 1, 1, 1)>(p.c.v[i]);
        g.Z.v[i] = qd_pull<XHE_QP(2, 2, 2, 2)>(p.c.v[i]); g.T.v[i] = qd_pull<XHE_QP(3, 3, 3, 3)>(p.c.v[i]);
    }
    return g;
}

}  // namespace xhe
