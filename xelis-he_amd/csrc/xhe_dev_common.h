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

static inline unsigned grid_for(u64 n, unsigned block) { return (unsigned)((n + block - 1) / block); }

}  // namespace xhe
