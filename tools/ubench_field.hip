// Micro-benchmark: complete field squarings and products in dependent chains (the shape of the point decode's 254
// back-to-back squarings), in the library's own form and in candidate forms with fewer or cheaper instructions.
// Every variant is a real GF(2^255-19) operation: the results of all variants are compared on the host (as values mod p).
//
//   sq_lib      fe_sq of xhe_fe.h: ten limbs of 25.5 bits, 55 products into ten 64-bit columns, carries added afterwards
//   sq_chain    the same products, every column's chain started from the carry of the column below it
//               (no separate 64-bit addition per column; the columns become one dependent chain)
//   sq_sat8     eight saturated 32-bit limbs, 36 products through the multiply-add's carry-out into 96-bit columns,
//               doubling per column, fold of the upper half by 38
//   mul_lib / mul_chain   fe_mul in the two forms (100 products)
//
// Grid: 256 workgroups x waves-per-SIMD of 256 lanes, every lane one chain of ITERS operations; reports the time, and the
// cycles per operation per wavefront per SIMD at the measured shader clock.
//
//   hipcc --offload-arch=gfx950 -O3 -I xelis-he_amd/csrc -o tools/ubench_field tools/ubench_field.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "xhe_fe.h"

using namespace xhe;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- chain form of the 25.5-bit arithmetic ----
// (the multiply-add is written as an instruction: left to itself the compiler re-associates the 64-bit additions and restores
// the library's form, carry additions included)
__device__ __forceinline__ u64 mad64_asm(u32 a, u32 b, u64 c) {
    u64 r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c) : "vcc");
    return r;
}
template <int A>
__device__ __forceinline__ fe fe_sq_chain(const fe_t<A> &f) {
    static_assert(A <= 3, "operand magnitude too large for squaring");
    u32 f2[10], f4[10], f19[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { f2[i] = 2 * f.v[i]; f4[i] = 4 * f.v[i]; f19[i] = 19 * f.v[i]; }
    fe r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            if (i > j) continue;
            const bool wrap = (i + j) >= 10;
            const bool both_odd = (i & 1) && (j & 1);
            if (i == j) acc = mad64_asm(both_odd ? f2[i] : f.v[i], wrap ? f19[j] : f.v[j], acc);
            else acc = mad64_asm(both_odd ? f4[i] : f2[i], wrap ? f19[j] : f.v[j], acc);
        }
        if (k & 1) { r.v[k] = (u32)acc & FE_MASK25; acc >>= 25; } else { r.v[k] = (u32)acc & FE_MASK26; acc >>= 26; }
    }
    u64 t = (u64)r.v[0] + 19 * acc;
    r.v[0] = (u32)t & FE_MASK26;
    r.v[1] += (u32)(t >> 26);
    return r;
}
template <int A, int B>
__device__ __forceinline__ fe fe_mul_chain(const fe_t<A> &f, const fe_t<B> &g) {
    static_assert(B <= 3 && A * B <= 24 && A <= 24, "operand magnitudes");
    u32 g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { g19[i] = 19 * g.v[i]; f2[i] = 2 * f.v[i]; }
    fe r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            const bool wrap = (i + j) >= 10;
            const bool both_odd = (i & 1) && (j & 1);
            acc = mad64_asm(both_odd ? f2[i] : f.v[i], wrap ? g19[j] : g.v[j], acc);
        }
        if (k & 1) { r.v[k] = (u32)acc & FE_MASK25; acc >>= 25; } else { r.v[k] = (u32)acc & FE_MASK26; acc >>= 26; }
    }
    u64 t = (u64)r.v[0] + 19 * acc;
    r.v[0] = (u32)t & FE_MASK26;
    r.v[1] += (u32)(t >> 26);
    return r;
}

// ---- eight saturated limbs: a value below 2^256, kept modulo 2^256 - 38 = 2p ----
struct sat8 { u32 w[8]; };
// column += a * b with the carry-out of the 64-bit multiply-add going into the third word
__device__ __forceinline__ void mac96(u64 &lo, u32 &hi, u32 a, u32 b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ sat8 sat8_sq(const sat8 &a) {
    u32 p[16];
    u64 lo = 0; u32 hi = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
        // the off-diagonal products of the column, doubled, plus the square on even columns, on top of the carried part
        u64 olo = 0; u32 ohi = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j <= i || j > 7) continue;
            mac96(olo, ohi, a.w[i], a.w[j]);
        }
        ohi = (ohi << 1) | (u32)(olo >> 63); olo <<= 1;
        if ((k & 1) == 0) mac96(olo, ohi, a.w[k / 2], a.w[k / 2]);
        // column total = carried + doubled products (fits 96 bits: below 2^70)
        const u64 s = lo + olo;
        hi = hi + ohi + (s < lo ? 1u : 0u);
        p[k] = (u32)s;
        lo = (s >> 32) | ((u64)hi << 32); hi = 0;
    }
    p[15] = (u32)lo;
    // 2^256 = 38 (mod 2p): fold the upper eight words
    sat8 r; u64 c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { c = mad64(p[8 + k], 38u, c + p[k]); r.w[k] = (u32)c; c >>= 32; }
    // c < 39: fold once more; the second carry-out can only occur with a tiny low word, where adding 38 cannot carry again
    u64 t = (u64)r.w[0] + 38 * c; r.w[0] = (u32)t; t >>= 32;
#pragma unroll
    for (int k = 1; k < 8; k++) { t += r.w[k]; r.w[k] = (u32)t; t >>= 32; }
    r.w[0] += 38 * (u32)t;
    return r;
}

template <int V>
__global__ void __launch_bounds__(256) k_field(u32 *io, int iters) {
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = io[8ull * t + i];
    if constexpr (V == 2) {
        sat8 a;
#pragma unroll
        for (int i = 0; i < 8; i++) a.w[i] = w[i];
        for (int i = 0; i < iters; i++) a = sat8_sq(a);
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = a.w[i];
    } else {
        fe a = fe_unpack(w), b = a;
        for (int i = 0; i < iters; i++) {
            if constexpr (V == 0) a = fe_sq(a);
            else if constexpr (V == 1) a = fe_sq_chain(a);
            else if constexpr (V == 3) a = fe_mul(a, b);
            else a = fe_mul_chain(a, b);
        }
        fe_pack(w, a);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) io[8ull * t + i] = w[i];
}

// value mod p of eight words below 2^256, as canonical words (host)
static void canonical(u32 w[8]) {
    // p = 2^255 - 19: subtract p while the value is >= p (at most twice for a value below 2^256)
    for (int round = 0; round < 3; round++) {
        const u32 P[8] = {0xffffffedu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x7fffffffu};
        bool ge = true;
        for (int i = 7; i >= 0; i--) { if (w[i] > P[i]) break; if (w[i] < P[i]) { ge = false; break; } }
        if (!ge) return;
        uint64_t borrow = 0;
        for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)w[i] - P[i] - borrow; w[i] = (u32)d; borrow = (d >> 63) & 1; }
    }
}

template <int V>
static double run(const char *name, int waves_per_simd, int iters, const std::vector<u32> &input, std::vector<u32> &result, double clock_hz, int products) {
    const int blocks = 256 * waves_per_simd;            // 4 wavefronts per workgroup, one per SIMD: waves_per_simd workgroups per CU
    const size_t words = (size_t)blocks * 256 * 8;
    u32 *d;
    CHECK(hipMalloc(&d, words * 4));
    CHECK(hipMemcpy(d, input.data(), words * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_field<V>, dim3(blocks), dim3(256), 0, 0, d, 8);        // warm-up on scratch values
    CHECK(hipMemcpy(d, input.data(), words * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_field<V>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    result.resize(words);
    CHECK(hipMemcpy(result.data(), d, words * 4, hipMemcpyDeviceToHost));
    CHECK(hipFree(d)); CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    const double cycles = ms * 1e-3 * clock_hz / iters / waves_per_simd;
    printf("%-10s waves/SIMD=%d  %8.3f ms  %7.1f cycles per operation per wavefront  (%d products: %.2f cycles per product)\n", name, waves_per_simd, ms, cycles, products, cycles / products);
    fflush(stdout);
    return cycles;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2032;          // eight chains of 254
    if (iters < 1 || iters > 100000) return 2;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const double clock_hz = prop.clockRate * 1e3;
    printf("device %s, %d CUs, nominal clock %.0f MHz, %d operations per chain\n", prop.gcnArchName, prop.multiProcessorCount, clock_hz / 1e6, iters);
    for (int wps : {4, 8, 2}) {
        const size_t words = (size_t)256 * wps * 256 * 8;
        std::vector<u32> in(words), r0, r1, r2, m0, m1;
        u64 s = 0x9e3779b97f4a7c15ull;
        for (size_t i = 0; i < words; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; in[i] = (u32)(s >> 11); if ((i & 7) == 7) in[i] &= 0x7fffffffu; }
        run<0>("sq_lib", wps, iters, in, r0, clock_hz, 55);
        run<1>("sq_chain", wps, iters, in, r1, clock_hz, 55);
        run<2>("sq_sat8", wps, iters, in, r2, clock_hz, 36);
        run<3>("mul_lib", wps, iters, in, m0, clock_hz, 100);
        run<4>("mul_chain", wps, iters, in, m1, clock_hz, 100);
        size_t bad1 = 0, bad2 = 0, badm = 0;
        for (size_t e = 0; e < words / 8; e++) {
            u32 x[8]; memcpy(x, &r2[8 * e], 32); canonical(x);
            if (memcmp(&r0[8 * e], &r1[8 * e], 32)) bad1++;
            if (memcmp(&r0[8 * e], x, 32)) bad2++;
            if (memcmp(&m0[8 * e], &m1[8 * e], 32)) badm++;
        }
        printf("agreement with the library form over %zu chains: sq_chain %s, sq_sat8 %s, mul_chain %s\n", words / 8, bad1 ? "DIFFERS" : "equal", bad2 ? "DIFFERS" : "equal", badm ? "DIFFERS" : "equal");
        if (bad1 || bad2 || badm) return 1;
    }
    return 0;
}
