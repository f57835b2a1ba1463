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

static inline unsigned grid_for(u64 n, unsigned block) { return (unsigned)((n + block - 1) / block); }

}  // namespace xhe
