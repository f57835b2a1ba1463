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

// batch factor: Keccak-f over (seed, domain, index), 64 bytes reduced mod l
__device__ __forceinline__ sc prf_scalar(const u32 *__restrict__ seed8, u32 domain, u64 index) {
    u64 st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) st[i] = (u64)seed8[2 * i] | ((u64)seed8[2 * i + 1] << 32);
    st[4] = index; st[5] = domain; st[6] = 0x1f; st[16] = 0x8000000000000000ULL;
    keccak_f1600(st);
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { w[2 * i] = (u32)st[i]; w[2 * i + 1] = (u32)(st[i] >> 32); }
    return sc_from_wide_words(w);
}

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

// 64-bit rotate left by a constant as two v_alignbit_b32 (the generic shift/or form costs two 64-bit shifts and two ORs)
template <int N>
__device__ __forceinline__ u64 rotl64c(u64 x) {
    static_assert(N > 0 && N < 64, "rotation count");
    u32 lo = (u32)x, hi = (u32)(x >> 32);
    if (N >= 32) { u32 t = lo; lo = hi; hi = t; }
    constexpr int R = N & 31;
    if (R == 0) return (u64)lo | ((u64)hi << 32);
    u32 nh = __builtin_amdgcn_alignbit(hi, lo, 32 - R), nl = __builtin_amdgcn_alignbit(lo, hi, 32 - R);
    return (u64)nl | ((u64)nh << 32);
}
// Keccak-f[1600] on 25 words that stay in registers (every index is a compile-time constant)
__device__ __forceinline__ void keccak_f1600_regs(u64 (&a)[25]) {
    const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        u64 c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20];
        u64 c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21];
        u64 c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22];
        u64 c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23];
        u64 c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];
        u64 d0 = c4 ^ rotl64c<1>(c1), d1 = c0 ^ rotl64c<1>(c2), d2 = c1 ^ rotl64c<1>(c3), d3 = c2 ^ rotl64c<1>(c4), d4 = c3 ^ rotl64c<1>(c0);
#pragma unroll
        for (int y = 0; y < 25; y += 5) { a[y] ^= d0; a[y + 1] ^= d1; a[y + 2] ^= d2; a[y + 3] ^= d3; a[y + 4] ^= d4; }
        u64 b[25];
        b[0] = a[0];
        b[10] = rotl64c<1>(a[1]);   b[20] = rotl64c<62>(a[2]);  b[5] = rotl64c<28>(a[3]);   b[15] = rotl64c<27>(a[4]);
        b[16] = rotl64c<36>(a[5]);  b[1] = rotl64c<44>(a[6]);   b[11] = rotl64c<6>(a[7]);   b[21] = rotl64c<55>(a[8]);  b[6] = rotl64c<20>(a[9]);
        b[7] = rotl64c<3>(a[10]);   b[17] = rotl64c<10>(a[11]); b[2] = rotl64c<43>(a[12]);  b[12] = rotl64c<25>(a[13]); b[22] = rotl64c<39>(a[14]);
        b[23] = rotl64c<41>(a[15]); b[8] = rotl64c<45>(a[16]);  b[18] = rotl64c<15>(a[17]); b[3] = rotl64c<21>(a[18]);  b[13] = rotl64c<8>(a[19]);
        b[14] = rotl64c<18>(a[20]); b[24] = rotl64c<2>(a[21]);  b[9] = rotl64c<61>(a[22]);  b[19] = rotl64c<56>(a[23]); b[4] = rotl64c<14>(a[24]);
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            a[y] = b[y] ^ (~b[y + 1] & b[y + 2]);
            a[y + 1] = b[y + 1] ^ (~b[y + 2] & b[y + 3]);
            a[y + 2] = b[y + 2] ^ (~b[y + 3] & b[y + 4]);
            a[y + 3] = b[y + 3] ^ (~b[y + 4] & b[y]);
            a[y + 4] = b[y + 4] ^ (~b[y] & b[y + 1]);
        }
        a[0] ^= RC[round];
    }
}

static inline unsigned grid_for(u64 n, unsigned block) { return (unsigned)((n + block - 1) / block); }

}  // namespace xhe
