// Extended twisted-Edwards arithmetic with one point per wavefront (device only).
//
// The Horner stage of the multiscalar multiplication is one dependent chain of ~240 doublings: its run time
// is the instruction count of one doubling times the chain length, whatever the machine's width.  Here the
// whole wavefront shares one point.  Row r (lanes 16r .. 16r+15) holds coordinate r of (X, Y, Z, T) and lane k
// of a row holds limb k of that coordinate in radix 2^25.5 (k < 10; lanes 10..15 of every row hold zero), so a
// point is ONE 32-bit register across the wave.  A field product is ten multiply-adds per lane: limb i of f is
// broadcast over the row (DPP row_newbcast), g is rotated by i limbs (DPP row_shr for the limbs that stay below
// 2^255, row_shl of 19*g for those that wrap), and every lane accumulates the column of its own limb.  Carries
// run as two parallel rounds (every lane hands its carry to the next at once).  Coordinates cross rows with
// v_permlane16_swap / v_permlane32_swap.  A doubling is two such products plus ~40 cheap instructions,
// ≈ 170 instructions against ≈ 420 per lane for the four-lanes-per-point form (xhe_quad.h) and ≈ 1 300 for a
// point kept in one lane.
//
// All 64 lanes must be active whenever these functions run.
//
// Same formulas as ge_double / ge_add of xhe_ge.h, which restate curve25519-dalek's EdwardsPoint arithmetic as
// used through /root/reference/src/proofs.rs:49-67 (the multiscalar multiplication of BatchCollector::verify).
#pragma once
#include "xhe_ge.h"

namespace xhe {

// per-lane constants of the layout
struct wd_lane {
    u32 k, row;          // limb index inside the row (0..15, limbs are 0..9), coordinate index
    u32 sh;              // bits of this limb: 26 (even k) or 25 (odd k)
    u32 mask;            // (1 << sh) - 1, zero on the idle lanes
    u32 live;            // all-ones on limb lanes, zero on the idle lanes
    u32 even_shift;      // 1 on even limbs: the factor two of odd*odd limb products lands on even columns
    u32 p2;              // limb k of 2p (zero on the idle lanes)
};
__device__ __forceinline__ wd_lane wd_lane_init() {
    wd_lane L;
    u32 lane = threadIdx.x & 63;
    L.k = lane & 15; L.row = lane >> 4;
    bool live = L.k < 10;
    L.sh = (L.k & 1) ? 25 : 26;
    L.mask = live ? ((1u << L.sh) - 1) : 0u;
    L.live = live ? 0xffffffffu : 0u;
    L.even_shift = (L.k & 1) ? 0u : 1u;
    L.p2 = !live ? 0u : L.k == 0 ? 2u * ((1u << 26) - 19) : 2u * ((L.k & 1) ? FE_MASK25 : FE_MASK26);
    return L;
}

// DPP move inside each row of 16 lanes; lanes whose source falls outside the row read zero
template <int CTRL>
__device__ __forceinline__ u32 wd_dpp(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
static constexpr int WD_ROW_SHL = 0x100, WD_ROW_SHR = 0x110, WD_ROW_BCAST = 0x150;   // lane k <- k+n, k-n, n

// what the previous lane carries in: lane k <- lane k-1, and limb 0 <- 19 * (limb 9's carry)
__device__ __forceinline__ u32 wd_carry_in(u32 c, u32 c19) { return wd_dpp<WD_ROW_SHR + 1>(c) | wd_dpp<WD_ROW_SHL + 9>(c19); }

template <int I>
__device__ __forceinline__ void wd_mul_step(u64 &h, u32 f, u32 g, u32 g19, const wd_lane &L) {
    u32 fb = wd_dpp<WD_ROW_BCAST + I>(f);
    if constexpr ((I & 1) != 0) fb <<= L.even_shift;
    u32 t;
    if constexpr (I == 0) t = g;
    else t = wd_dpp<WD_ROW_SHR + I>(g) | wd_dpp<WD_ROW_SHL + 10 - I>(g19);
    h += (u64)fb * t;
}
// f*g per row.  Limb magnitudes (multiples of 2^26 / 2^25): g <= 3.3 (19*g must fit 32 bits), f*g <= 20 (the
// columns must fit 64 bits); the idle lanes of f and g must hold zero.  Result magnitude <= 1.01.
__device__ __forceinline__ u32 wd_mul(u32 f, u32 g, const wd_lane &L) {
    u32 g19 = g * 19u;
    u64 h = 0;
    wd_mul_step<0>(h, f, g, g19, L); wd_mul_step<1>(h, f, g, g19, L); wd_mul_step<2>(h, f, g, g19, L); wd_mul_step<3>(h, f, g, g19, L);
    wd_mul_step<4>(h, f, g, g19, L); wd_mul_step<5>(h, f, g, g19, L); wd_mul_step<6>(h, f, g, g19, L); wd_mul_step<7>(h, f, g, g19, L);
    wd_mul_step<8>(h, f, g, g19, L); wd_mul_step<9>(h, f, g, g19, L);
    if (!L.live) h = 0;                                // the idle lanes accumulated limbs that slid past limb 9
    // first carry pass: carries of up to 39 bits
    u64 c = h >> L.sh;
    u32 lo = (u32)h & L.mask;
    u32 c_lo = (u32)c, c_hi = (u32)(c >> 32);
    u64 c19 = (u64)c_lo * 19u + ((u64)(c_hi * 19u) << 32);
    u32 in_lo = wd_carry_in(c_lo, (u32)c19), in_hi = wd_carry_in(c_hi, (u32)(c19 >> 32));
    u64 h1 = (u64)lo + (((u64)in_hi << 32) | in_lo);
    // second carry pass: carries below 2^18; the idle lanes received limb 9's carry and must not pass anything on
    u32 c1 = (u32)(h1 >> L.sh) & L.live;
    u32 lo1 = (u32)h1 & L.mask;
    return (lo1 + wd_carry_in(c1, c1 * 19u)) & L.live;
}
// one parallel carry round on 32-bit limbs: any magnitude <= 60 back to <= 1.01
__device__ __forceinline__ u32 wd_weak_reduce(u32 v, const wd_lane &L) {
    u32 c = v >> L.sh, lo = v & L.mask;
    return (lo + wd_carry_in(c, c * 19u)) & L.live;
}

// the four rows of v, each spread over the whole wave
__device__ __forceinline__ void wd_rows(u32 v, u32 &r0, u32 &r1, u32 &r2, u32 &r3) {
    // v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second,
    // v_permlane32_swap the upper half of the first with the lower half of the second
    auto s = __builtin_amdgcn_permlane16_swap(v, v, false, false);       // (r0 r0 r2 r2), (r1 r1 r3 r3)
    auto e = __builtin_amdgcn_permlane32_swap(s[0], s[0], false, false); // (r0 x4), (r2 x4)
    auto o = __builtin_amdgcn_permlane32_swap(s[1], s[1], false, false); // (r1 x4), (r3 x4)
    r0 = e[0]; r2 = e[1]; r1 = o[0]; r3 = o[1];
}

struct wd_point { u32 v; };                  // rows: X, Y, Z, T

// 2p (the T row of p is not used)
__device__ __forceinline__ wd_point wd_double(wd_point p, const wd_lane &L) {
    u32 x, y, z, t;
    wd_rows(p.v, x, y, z, t);
    u32 op = L.row == 3 ? x + y : p.v;                 // X, Y, Z, X + Y            (magnitude <= 2.02)
    u32 a, b, zz, s;
    wd_rows(wd_mul(op, op, L), a, b, zz, s);           // A = X^2, B = Y^2, ZZ = Z^2, S = (X + Y)^2
    u32 h = a + b;                                     // H = A + B                 (<= 2.02)
    u32 g = b + (L.p2 - a);                            // G = B - A                 (<= 3.01)
    u32 e = s + (2u * L.p2 - h);                       // E = S - A - B             (<= 5.01)
    u32 f = wd_weak_reduce(2u * zz + (2u * L.p2 - g), L);   // F = 2 ZZ - G         (6.02 -> 1.01)
    // (X, Y, Z, T) = (E F, G H, F G, E H); the second operand is the one multiplied by 19
    u32 m1 = L.row == 1 ? h : L.row == 2 ? g : e;      // E, H, G, E
    u32 m2 = L.row == 1 ? g : L.row == 3 ? h : f;      // F, G, F, H
    wd_point r; r.v = wd_mul(m1, m2, L);
    return r;
}

// the second operand of an addition, prepared once: (Y - X, Y + X, 2d T, 2 Z)
struct wd_cached { u32 v; };
__device__ __forceinline__ u32 wd_const_d2(const wd_lane &L) {
    fe d2 = fe_c_d2();
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) v = L.k == (u32)i ? d2.v[i] : v;
    return v;
}
__device__ __forceinline__ wd_cached wd_to_cached(wd_point q, const wd_lane &L) {
    u32 x, y, z, t;
    wd_rows(q.v, x, y, z, t);
    u32 td = wd_mul(t, wd_const_d2(L), L);             // every row computes 2d T; row 2 keeps it
    wd_cached c;
    c.v = L.row == 0 ? y + (L.p2 - x) : L.row == 1 ? y + x : L.row == 2 ? td : 2u * z;      // <= 3.01
    return c;
}
// p + q
__device__ __forceinline__ wd_point wd_add(wd_point p, wd_cached q, const wd_lane &L) {
    u32 x, y, z, t;
    wd_rows(p.v, x, y, z, t);
    u32 op = L.row == 0 ? y + (L.p2 - x) : L.row == 1 ? y + x : L.row == 2 ? t : z;         // Y-X, Y+X, T, Z
    u32 a, b, c, d;
    wd_rows(wd_mul(op, q.v, L), a, b, c, d);           // A, B, C = 2d T1 T2, D = 2 Z1 Z2
    u32 e = b + (L.p2 - a), h = b + a, f = d + (L.p2 - c), g = d + c;                       // <= 3.01
    u32 m1 = L.row == 1 ? g : L.row == 2 ? f : e;      // E, G, F, E
    u32 m2 = L.row == 0 ? f : L.row == 2 ? g : h;      // F, H, G, H
    wd_point r; r.v = wd_mul(m1, m2, L);
    return r;
}

// memory <-> wave: an XYZT record is four coordinates of eight 32-bit words (255 bits, little endian)
__device__ __forceinline__ wd_point wd_load(const u32 *__restrict__ src, const wd_lane &L) {
    u32 k = L.k < 10 ? L.k : 0;
    u32 off = (51 * k + 1) >> 1;                       // first bit of limb k: 0 26 51 77 102 128 153 179 204 230
    u32 wi = off >> 5; if (wi > 6) wi = 6;
    const u32 *w = src + 8 * L.row + wi;
    u64 pair = (u64)w[0] | ((u64)w[1] << 32);
    wd_point p; p.v = (u32)(pair >> (off - 32 * wi)) & L.mask;
    return p;
}
__device__ __forceinline__ wd_point wd_from_ge(const ge &g, const wd_lane &L) {
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        u32 limb = L.row == 0 ? g.X.v[i] : L.row == 1 ? g.Y.v[i] : L.row == 2 ? g.Z.v[i] : g.T.v[i];
        v = L.k == (u32)i ? limb : v;
    }
    wd_point p; p.v = v;
    return p;
}
// the whole point in every lane (wave-uniform)
__device__ __forceinline__ ge wd_gather(wd_point p) {
    ge g;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        g.X.v[i] = (u32)__builtin_amdgcn_readlane((int)p.v, i);      g.Y.v[i] = (u32)__builtin_amdgcn_readlane((int)p.v, 16 + i);
        g.Z.v[i] = (u32)__builtin_amdgcn_readlane((int)p.v, 32 + i); g.T.v[i] = (u32)__builtin_amdgcn_readlane((int)p.v, 48 + i);
    }
    return g;
}

}  // namespace xhe
