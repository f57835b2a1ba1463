// Device helpers shared by the verification kernels: unaligned loads from the record bytes, scalar
// load/store in both memory formats, the batch-factor PRF, Niels <-> extended conversions.
#pragma once
#include <hip/hip_runtime.h>
#include "xhe_ge.h"
#include "xhe_hash.h"
#include "xhe_verify_types.h"
#include "../../include/xhe.h"

namespace xhe {

__device__ __forceinline__ void load_words_unaligned(u32 w[8], const u8 *__restrict__ p) {
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = (u32)p[4 * i] | ((u32)p[4 * i + 1] << 8) | ((u32)p[4 * i + 2] << 16) | ((u32)p[4 * i + 3] << 24);
}
__device__ __forceinline__ sc load_scalar_bytes(const u8 *__restrict__ p) { u32 w[8]; load_words_unaligned(w, p); return sc_from_words(w); }
__device__ __forceinline__ void store_scalar_words(u32 *__restrict__ dst, const sc &x) {
    u32 w[8]; sc_to_words(w, x);
#pragma unroll
    for (int i = 0; i < 8; i++) dst[i] = w[i];
}
__device__ __forceinline__ sc load_scalar_words(const u32 *__restrict__ src) {
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = src[i];
    return sc_from_words(w);
}
// Montgomery-form limbs in memory (9 words)
__device__ __forceinline__ void store_sc_limbs(u32 *__restrict__ dst, const sc &x) {
#pragma unroll
    for (int i = 0; i < 9; i++) dst[i] = x.v[i];
}
__device__ __forceinline__ sc load_sc_limbs(const u32 *__restrict__ src) {
    sc x;
#pragma unroll
    for (int i = 0; i < 9; i++) x.v[i] = src[i];
    return x;
}
__device__ __forceinline__ void raise_err(u32 *tx_err, u32 tx, u32 key) { atomicMin(&tx_err[tx], key); }

__device__ __forceinline__ ge ge_from_niels_affine(const ge_niels &q) {
    // x = (ypx - ymx)/2, y = (ypx + ymx)/2
    fe inv2 = fe_zero();            // (p+1)/2 = 2^254 - 9
    inv2.v[0] = (1u << 26) - 9;
#pragma unroll
    for (int i = 1; i < 9; i++) inv2.v[i] = (i & 1) ? FE_MASK25 : FE_MASK26;
    inv2.v[9] = (1u << 24) - 1;
    ge p;
    p.X = fe_mul(fe_sub(q.ypx, q.ymx), inv2);
    p.Y = fe_mul(fe_add(q.ypx, q.ymx), inv2);
    p.Z = fe_one();
    p.T = fe_mul(p.X, p.Y);
    return p;
}
__device__ __forceinline__ ge_niels ge_to_niels_normalize(const ge &p) {
    fe zi = fe_invert(p.Z);
    ge a; a.X = fe_mul(p.X, zi); a.Y = fe_mul(p.Y, zi); a.Z = fe_one(); a.T = fe_mul(a.X, a.Y);
    return ge_to_niels_affine(a);
}

// Kernels of few, long-running wavefronts that sit on the critical path (hash chains, the Horner stage,
// the bucket reduction) raise their wave priority so the SIMD arbiter issues them ahead of the wide
// throughput kernels (point decoding) that run beside them on other streams.
__device__ __forceinline__ void wave_priority_high() { __builtin_amdgcn_s_setprio(3); }
// the point decoding and the static-generator scalars: the multiscalar check waits for them, the signatures (default
// priority) have slack until the verdict
__device__ __forceinline__ void wave_priority_mid() { __builtin_amdgcn_s_setprio(1); }
// the decodes that head the ledger scan and the signatures: ahead of the bulk, but behind the hash chains of the transcripts
// that start at the same moment (at equal priority the two share the SIMDs and the chain takes twice as long)
__device__ __forceinline__ void wave_priority_upper() { __builtin_amdgcn_s_setprio(2); }

// 64-bit rotate left by a constant on a (lo, hi) pair of 32-bit halves: two v_alignbit_b32
template <int N>
__device__ __forceinline__ void rotl64_halves(u32 &out_lo, u32 &out_hi, u32 lo, u32 hi) {
    static_assert(N > 0 && N < 64, "rotation count");
    if (N >= 32) { u32 t = lo; lo = hi; hi = t; }
    constexpr int R = N & 31;
    if (R == 0) { out_lo = lo; out_hi = hi; return; }
    out_hi = __builtin_amdgcn_alignbit(hi, lo, 32 - R);
    out_lo = __builtin_amdgcn_alignbit(lo, hi, 32 - R);
}
template <int N>
__device__ __forceinline__ u64 rotl64c(u64 x) {
    u32 lo, hi;
    rotl64_halves<N>(lo, hi, (u32)x, (u32)(x >> 32));
    return (u64)lo | ((u64)hi << 32);
}
// One round of Keccak-f[1600] on one 32-bit half-plane pair.  The state is kept as 2 x 25 32-bit words so that every
// three-input boolean expression (the five-way column parities, chi's a ^ (~b & c)) is one v_bitop3_b32; written on
// 64-bit words the same expressions compile to two-input operations.
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
#define XHE_KECCAK_RHO_PI(DST, SRC, N) rotl64_halves<N>(bl[DST], bh[DST], al[SRC], ah[SRC])
__device__ __forceinline__ void keccak_round_halves(u32 (&al)[25], u32 (&ah)[25], u32 rc_lo, u32 rc_hi) {
    u32 cl[5], ch[5], dl[5], dh[5];
#pragma unroll
    for (int x = 0; x < 5; x++) {
        cl[x] = xor3(xor3(al[x], al[x + 5], al[x + 10]), al[x + 15], al[x + 20]);
        ch[x] = xor3(xor3(ah[x], ah[x + 5], ah[x + 10]), ah[x + 15], ah[x + 20]);
    }
#pragma unroll
    for (int x = 0; x < 5; x++) {
        u32 rl, rh;
        rotl64_halves<1>(rl, rh, cl[(x + 1) % 5], ch[(x + 1) % 5]);
        dl[x] = cl[(x + 4) % 5] ^ rl;
        dh[x] = ch[(x + 4) % 5] ^ rh;
    }
#pragma unroll
    for (int i = 0; i < 25; i++) { al[i] ^= dl[i % 5]; ah[i] ^= dh[i % 5]; }
    u32 bl[25], bh[25];
    bl[0] = al[0]; bh[0] = ah[0];
    XHE_KECCAK_RHO_PI(10, 1, 1);   XHE_KECCAK_RHO_PI(20, 2, 62);  XHE_KECCAK_RHO_PI(5, 3, 28);   XHE_KECCAK_RHO_PI(15, 4, 27);
    XHE_KECCAK_RHO_PI(16, 5, 36);  XHE_KECCAK_RHO_PI(1, 6, 44);   XHE_KECCAK_RHO_PI(11, 7, 6);   XHE_KECCAK_RHO_PI(21, 8, 55);
    XHE_KECCAK_RHO_PI(6, 9, 20);   XHE_KECCAK_RHO_PI(7, 10, 3);   XHE_KECCAK_RHO_PI(17, 11, 10); XHE_KECCAK_RHO_PI(2, 12, 43);
    XHE_KECCAK_RHO_PI(12, 13, 25); XHE_KECCAK_RHO_PI(22, 14, 39); XHE_KECCAK_RHO_PI(23, 15, 41); XHE_KECCAK_RHO_PI(8, 16, 45);
    XHE_KECCAK_RHO_PI(18, 17, 15); XHE_KECCAK_RHO_PI(3, 18, 21);  XHE_KECCAK_RHO_PI(13, 19, 8);  XHE_KECCAK_RHO_PI(14, 20, 18);
    XHE_KECCAK_RHO_PI(24, 21, 2);  XHE_KECCAK_RHO_PI(9, 22, 61);  XHE_KECCAK_RHO_PI(19, 23, 56); XHE_KECCAK_RHO_PI(4, 24, 14);
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
#pragma unroll
        for (int x = 0; x < 5; x++) {
            al[y + x] = bl[y + x] ^ (~bl[y + (x + 1) % 5] & bl[y + (x + 2) % 5]);
            ah[y + x] = bh[y + x] ^ (~bh[y + (x + 1) % 5] & bh[y + (x + 2) % 5]);
        }
    }
    al[0] ^= rc_lo; ah[0] ^= rc_hi;
}
#undef XHE_KECCAK_RHO_PI
// Keccak-f[1600] on 25 words that stay in registers (every index is a compile-time constant)
__device__ __forceinline__ void keccak_f1600_regs(u64 (&a)[25]) {
    const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    u32 al[25], ah[25];
#pragma unroll
    for (int i = 0; i < 25; i++) { al[i] = (u32)a[i]; ah[i] = (u32)(a[i] >> 32); }
#pragma unroll 1
    for (int round = 0; round < 24; round++) keccak_round_halves(al, ah, (u32)RC[round], (u32)(RC[round] >> 32));
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = (u64)al[i] | ((u64)ah[i] << 32);
}

// ---- Keccak-f[1600] on a pair of lanes ----------------------------------------------------------------------------
// Bit-interleaved state: the even lane of a pair holds the even-numbered bits of the 25 words, the odd lane the odd-numbered
// ones (25 registers each).  theta, chi and iota are bitwise and stay inside a lane; a 64-bit rotation by 2k is a 32-bit
// rotation by k of both halves, one by 2k+1 takes the partner's half (one DPP move) and rotates it by k+1 (even lane) or k
// (odd lane).  A lane issues ~110 instructions per round instead of ~195 for the whole state, which is what counts for a
// hash chain that is bound by the issue rate of one wavefront.
// even bits of a 64-bit word -> low 32 bits, odd bits -> high 32 bits
__host__ __device__ constexpr u32 bit_unzip32(u32 x) {
    u32 t = (x ^ (x >> 1)) & 0x22222222u; x ^= t ^ (t << 1);
    t = (x ^ (x >> 2)) & 0x0c0c0c0cu; x ^= t ^ (t << 2);
    t = (x ^ (x >> 4)) & 0x00f000f0u; x ^= t ^ (t << 4);
    t = (x ^ (x >> 8)) & 0x0000ff00u; x ^= t ^ (t << 8);
    return x;                                          // even bits in the low half, odd bits in the high half
}
__host__ __device__ constexpr u32 bit_zip32(u32 x) {
    u32 t = (x ^ (x >> 8)) & 0x0000ff00u; x ^= t ^ (t << 8);
    t = (x ^ (x >> 4)) & 0x00f000f0u; x ^= t ^ (t << 4);
    t = (x ^ (x >> 2)) & 0x0c0c0c0cu; x ^= t ^ (t << 2);
    t = (x ^ (x >> 1)) & 0x22222222u; x ^= t ^ (t << 1);
    return x;
}
__host__ __device__ constexpr u64 bit_unzip64(u64 v) {
    u32 lo = bit_unzip32((u32)v), hi = bit_unzip32((u32)(v >> 32));
    u32 even = (lo & 0xffffu) | (hi << 16), odd = (lo >> 16) | (hi & 0xffff0000u);
    return (u64)even | ((u64)odd << 32);
}
__host__ __device__ constexpr u64 bit_zip64(u32 even, u32 odd) {
    u32 lo = bit_zip32((even & 0xffffu) | (odd << 16)), hi = bit_zip32((even >> 16) | (odd & 0xffff0000u));
    return (u64)lo | ((u64)hi << 32);
}
// the partner lane's value (lanes 2t and 2t+1 form a pair)
__device__ __forceinline__ u32 pair_swap(u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1 /* quad_perm 1,0,3,2 */, 0xf, 0xf, false); }
__device__ __forceinline__ u32 rotl32v(u32 x, u32 neg_count /* (32 - k) & 31 */) { return __builtin_amdgcn_alignbit(x, x, neg_count); }
// rotation counts of rho in the order of XHE_KECCAK2_RHO_PI below; odd counts 2k+1 carry their k in a per-lane register
struct keccak_pair_consts { u32 one; u32 k[12]; };     // (32 - (k + is_even_lane)) & 31 for theta's rotation by 1 and the 12 odd rho counts
__device__ __forceinline__ keccak_pair_consts keccak_pair_setup(u32 odd_lane) {
    constexpr int odd_counts[12] = {1, 27, 55, 3, 43, 25, 39, 41, 45, 15, 21, 61};
    keccak_pair_consts c;
    c.one = (32u - (0u + (odd_lane ^ 1u))) & 31u;
#pragma unroll
    for (int i = 0; i < 12; i++) c.k[i] = (32u - ((u32)(odd_counts[i] >> 1) + (odd_lane ^ 1u))) & 31u;
    return c;
}
#define XHE_K2_EVEN(DST, SRC, N) b[DST] = __builtin_amdgcn_alignbit(s[SRC], s[SRC], (32 - (N) / 2) & 31)
#define XHE_K2_ODD(DST, SRC, IDX) b[DST] = rotl32v(pair_swap(s[SRC]), kc.k[IDX])
__device__ __forceinline__ void keccak_round_pair(u32 (&s)[25], const keccak_pair_consts &kc, u32 rc_half) {
    u32 c[5], r[5];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = xor3(xor3(s[x], s[x + 5], s[x + 10]), s[x + 15], s[x + 20]);
#pragma unroll
    for (int x = 0; x < 5; x++) r[x] = rotl32v(pair_swap(c[x]), kc.one);          // this lane's half of rotl64(C[x], 1)
#pragma unroll
    for (int i = 0; i < 25; i++) s[i] = xor3(s[i], c[(i % 5 + 4) % 5], r[(i % 5 + 1) % 5]);
    u32 b[25];
    b[0] = s[0];
    XHE_K2_ODD(10, 1, 0);    XHE_K2_EVEN(20, 2, 62);  XHE_K2_EVEN(5, 3, 28);   XHE_K2_ODD(15, 4, 1);
    XHE_K2_EVEN(16, 5, 36);  XHE_K2_EVEN(1, 6, 44);   XHE_K2_EVEN(11, 7, 6);   XHE_K2_ODD(21, 8, 2);
    XHE_K2_EVEN(6, 9, 20);   XHE_K2_ODD(7, 10, 3);    XHE_K2_EVEN(17, 11, 10); XHE_K2_ODD(2, 12, 4);
    XHE_K2_ODD(12, 13, 5);   XHE_K2_ODD(22, 14, 6);   XHE_K2_ODD(23, 15, 7);   XHE_K2_ODD(8, 16, 8);
    XHE_K2_ODD(18, 17, 9);   XHE_K2_ODD(3, 18, 10);   XHE_K2_EVEN(13, 19, 8);  XHE_K2_EVEN(14, 20, 18);
    XHE_K2_EVEN(24, 21, 2);  XHE_K2_ODD(9, 22, 11);   XHE_K2_EVEN(19, 23, 56); XHE_K2_EVEN(4, 24, 14);
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
#pragma unroll
        for (int x = 0; x < 5; x++) s[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
    }
    s[0] ^= rc_half;
}
#undef XHE_K2_EVEN
#undef XHE_K2_ODD
struct keccak_rc_halves { u32 even[24], odd[24]; };
__host__ __device__ constexpr keccak_rc_halves keccak_rc_unzipped() {
    const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    keccak_rc_halves h{};
    for (int i = 0; i < 24; i++) { u64 z = bit_unzip64(RC[i]); h.even[i] = (u32)z; h.odd[i] = (u32)(z >> 32); }
    return h;
}
// Keccak-f[1600] on this lane's half of the bit-interleaved state
__device__ __forceinline__ void keccak_f1600_pair(u32 (&s)[25], const keccak_pair_consts &kc, u32 odd_lane) {
    constexpr keccak_rc_halves RC = keccak_rc_unzipped();
#pragma unroll 1
    for (int round = 0; round < 24; round++) keccak_round_pair(s, kc, odd_lane ? RC.odd[round] : RC.even[round]);
}

// Batch factors: scalar(s) = wide-reduce(Keccak-f[1600](seed || index || domain || SHAKE padding)) -- one permutation
// yields up to two scalars (128 of the 136 rate bytes).
__device__ __forceinline__ void prf_state(u64 (&st)[25], const u32 *__restrict__ seed8, u32 domain, u64 index) {
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) st[i] = (u64)seed8[2 * i] | ((u64)seed8[2 * i + 1] << 32);
    st[4] = index; st[5] = domain; st[6] = 0x1f; st[16] = 0x8000000000000000ULL;
    keccak_f1600_regs(st);
}
__device__ __forceinline__ sc prf_scalar(const u32 *__restrict__ seed8, u32 domain, u64 index) {
    u64 st[25];
    prf_state(st, seed8, domain, index);
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { w[2 * i] = (u32)st[i]; w[2 * i + 1] = (u32)(st[i] >> 32); }
    return sc_from_wide_words(w);
}
// A 128-bit factor (never zero): what batch verification needs for a 2^-128 error bound.  A term whose scalar is the bare
// factor or its negative (the proofs' first commitments Y_0) then has digits in the lower half of the windows only, and the
// bucket sort and the accumulation skip the rest.
__device__ __forceinline__ sc prf_scalar_128(const u32 *__restrict__ seed8, u32 domain, u64 index) {
    u64 st[25];
    prf_state(st, seed8, domain, index);
    u32 w[8] = {(u32)st[0] | 1u, (u32)(st[0] >> 32), (u32)st[1], (u32)(st[1] >> 32), 0, 0, 0, 0};
    return sc_from_words(w);
}
// three independent odd factors of `bits` bits (128..191) from one state
__device__ __forceinline__ void prf_scalar_x3(sc &a, sc &b, sc &c, const u32 *__restrict__ seed8, u32 domain, u64 index, u32 bits) {
    u64 st[25];
    prf_state(st, seed8, domain, index);
    const u64 top = (bits >= 192) ? ~0ull : ((1ull << (bits - 128)) - 1);
    u32 w[8] = {(u32)st[0] | 1u, (u32)(st[0] >> 32), (u32)st[1], (u32)(st[1] >> 32), (u32)(st[2] & top), (u32)((st[2] & top) >> 32), 0, 0};
    a = sc_from_words(w);
    w[0] = (u32)st[3] | 1u; w[1] = (u32)(st[3] >> 32); w[2] = (u32)st[4]; w[3] = (u32)(st[4] >> 32); w[4] = (u32)(st[5] & top); w[5] = (u32)((st[5] & top) >> 32);
    b = sc_from_words(w);
    w[0] = (u32)st[6] | 1u; w[1] = (u32)(st[6] >> 32); w[2] = (u32)st[7]; w[3] = (u32)(st[7] >> 32); w[4] = (u32)(st[8] & top); w[5] = (u32)((st[8] & top) >> 32);
    c = sc_from_words(w);
}
__device__ __forceinline__ void prf_scalar_pair(sc &first, sc &second, const u32 *__restrict__ seed8, u32 domain, u64 index) {
    u64 st[25];
    prf_state(st, seed8, domain, index);
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { w[2 * i] = (u32)st[i]; w[2 * i + 1] = (u32)(st[i] >> 32); }
    first = sc_from_wide_words(w);
#pragma unroll
    for (int i = 0; i < 8; i++) { w[2 * i] = (u32)st[8 + i]; w[2 * i + 1] = (u32)(st[8 + i] >> 32); }
    second = sc_from_wide_words(w);
}

// a point from the lane `delta` above (wavefront shuffle of its 40 limbs) and a branch-free choice between two points
__device__ __forceinline__ ge ge_shfl_down(const ge &p, u32 delta) {
    ge r;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        r.X.v[i] = __shfl_down(p.X.v[i], delta, 64); r.Y.v[i] = __shfl_down(p.Y.v[i], delta, 64);
        r.Z.v[i] = __shfl_down(p.Z.v[i], delta, 64); r.T.v[i] = __shfl_down(p.T.v[i], delta, 64);
    }
    return r;
}
__device__ __forceinline__ ge ge_select(const ge &a, const ge &b, bool flag) {
    ge r; r.X = fe_select(a.X, b.X, flag); r.Y = fe_select(a.Y, b.Y, flag); r.Z = fe_select(a.Z, b.Z, flag); r.T = fe_select(a.T, b.T, flag);
    return r;
}

static inline unsigned grid_for(u64 n, unsigned block) { return (unsigned)((n + block - 1) / block); }

}  // namespace xhe
