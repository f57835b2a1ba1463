// Device self-test of the one-point-per-wavefront arithmetic (xelis-he_amd/csrc/xhe_wide.h) against the
// one-point-per-lane arithmetic (xhe_ge.h): field products at the magnitude limits, record loads, and a
// chain of doublings and additions as the Horner stage runs them.  Prints PASS or the first mismatch.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I xelis-he_amd/csrc -o tools/test_wide tools/test_wide.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "xhe_dev_common.h"
#include "xhe_wide.h"
using namespace xhe;

__device__ u32 rnd(u32 &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// out[0] = number of mismatches, out[1..] = first failing case
__global__ void __launch_bounds__(64) k_test(u32 *out, u32 *scratch /* 64 words */) {
    const wd_lane L = wd_lane_init();
    u32 bad = 0, first = 0;
    u32 seed = 12345u;
    // 1. products: every row gets its own pair of operands, limbs up to the documented magnitudes
    for (int it = 0; it < 64; it++) {
        fe_t<6> f[4]; fe_t<3> g[4];
        for (int r = 0; r < 4; r++)
            for (int i = 0; i < 10; i++) {
                u32 lim = (i & 1) ? (1u << 25) : (1u << 26);
                u32 a = rnd(seed), b = rnd(seed);
                f[r].v[i] = (it & 1) ? 5u * lim - 1 - (a & 0xff) : a % (5u * lim);
                g[r].v[i] = (it & 2) ? 3u * lim + (lim >> 2) - 1 - (b & 0xff) : b % (3u * lim);
            }
        u32 fv = 0, gv = 0;
        for (int i = 0; i < 10; i++) { fv = L.k == (u32)i ? f[L.row].v[i] : fv; gv = L.k == (u32)i ? g[L.row].v[i] : gv; }
        u32 hv = wd_mul(fv, gv, L);
        for (int r = 0; r < 4; r++) {
            fe_t<3> gg; for (int i = 0; i < 10; i++) gg.v[i] = g[r].v[i];
            fe ref = fe_mul(fe_reduce(f[r]), fe_reduce(gg));
            fe got; for (int i = 0; i < 10; i++) got.v[i] = (u32)__builtin_amdgcn_readlane((int)hv, 16 * r + i);
            bool ok = fe_eq(ref, got);
            for (int i = 0; i < 10; i++) ok = ok && got.v[i] <= (((i & 1) ? (1u << 25) : (1u << 26)) + (1u << 23));
            for (int i = 10; i < 16; i++) ok = ok && __builtin_amdgcn_readlane((int)hv, 16 * r + i) == 0;
            if (!ok) { if (!bad) first = 100 + it * 4 + r; bad++; }
        }
    }
    // 2. a chain as the Horner stage runs it, against ge_double / ge_add
    const u8 bb[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                       0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
    ge B; ristretto_decode(B, bb);
    ge ref = B, Q = ge_double(ge_double(B));
    wd_point W = wd_from_ge(B, L);
    for (int step = 0; step < 40; step++) {
        for (int k = 0; k < 7; k++) { ref = ge_double(ref); W = wd_double(W, L); }
        // the addend goes through memory like a window sum
        if (threadIdx.x == 0) ge_store(scratch, Q);
        __syncthreads();
        wd_cached qc = wd_to_cached(wd_load(scratch, L), L);
        __syncthreads();
        ref = ge_add(ref, Q); W = wd_add(W, qc, L);
        Q = ge_add(Q, B);
        ge got = wd_gather(W);
        bool ok = fe_eq(fe_mul(got.X, ref.Z), fe_mul(ref.X, got.Z)) && fe_eq(fe_mul(got.Y, ref.Z), fe_mul(ref.Y, got.Z)) &&
                  fe_eq(fe_mul(got.T, ref.Z), fe_mul(ref.T, got.Z)) && !fe_is_zero(got.Z);
        if (!ok) { if (!bad) first = 1000 + step; bad++; }
    }
    if (threadIdx.x == 0) { out[0] = bad; out[1] = first; }
}

int main() {
    u32 *d_out, *d_scr, h[2] = {99, 99};
    if (hipMalloc(&d_out, 8) != hipSuccess || hipMalloc(&d_scr, 256) != hipSuccess) { printf("no device\n"); return 2; }
    hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, d_out, d_scr);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
    if (hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("read-back failed\n"); return 2; }
    if (h[0]) { printf("FAIL: %u mismatches, first case %u\n", h[0], h[1]); return 1; }
    printf("PASS\n");
    return 0;
}
