// Occupancy experiment for the two arithmetic-bound kernels: ristretto decode and the mixed-addition chain of the bucket
// accumulation, compiled for 2, 3 and 4 wavefronts per SIMD (256, 168, 128 registers).  Prints ms per variant.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I xelis-he_amd/csrc -o tools/ubench_occupancy tools/ubench_occupancy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "xhe_dev_common.h"
using namespace xhe;

template <int OCC>
__global__ void __launch_bounds__(256, OCC) k_dec(const u32 *__restrict__ in, u32 n, u32 *__restrict__ out, u8 *__restrict__ ok) {
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        u32 w[8];
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = in[8ull * i + k];
        ge p; bool good = ristretto_decode_words(p, w);
        ge_niels_store(out + 32ull * i, ge_to_niels_affine(p));
        ok[i] = good ? 1 : 0;
    }
}
template <int OCC>
__global__ void __launch_bounds__(256, OCC) k_madd(const u32 *__restrict__ niels, u32 n_pts, u32 per_thread, u32 *__restrict__ out) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    ge acc = ge_identity();
    u32 idx = (t * 2654435761u) % n_pts;
    for (u32 k = 0; k < per_thread; k++) {
        ge_niels q = ge_niels_load(niels + 32ull * idx);
        acc = ge_madd_signed(acc, q, (k & 1) != 0);
        idx = (idx * 1664525u + 1013904223u) % n_pts;
    }
    ge_store(out + 32ull * t, acc);
}
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename F> float time_it(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 3; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 3;
}
int main() {
    const u32 n = 4u << 20;
    std::vector<u32> h(8ull * n);
    // valid encodings are not needed for timing: the work does not depend on the verdict (no early exit)
    for (size_t i = 0; i < h.size(); i++) h[i] = (u32)(i * 2654435761u) & 0x7ffffffeu;
    u32 *in, *out; u8 *ok;
    CHECK(hipMalloc(&in, 32ull * n)); CHECK(hipMalloc(&out, 128ull * n)); CHECK(hipMalloc(&ok, n));
    CHECK(hipMemcpy(in, h.data(), 32ull * n, hipMemcpyHostToDevice));
    for (int blocks : {512, 1024, 2048, 16384}) {
        printf("decode %u points, grid %d: ", n, blocks);
        printf("occ2 %.3f ms  ", time_it([&] { hipLaunchKernelGGL(k_dec<2>, dim3(blocks), dim3(256), 0, 0, in, n, out, ok); }));
        printf("occ3 %.3f ms  ", time_it([&] { hipLaunchKernelGGL(k_dec<3>, dim3(blocks), dim3(256), 0, 0, in, n, out, ok); }));
        printf("occ4 %.3f ms\n", time_it([&] { hipLaunchKernelGGL(k_dec<4>, dim3(blocks), dim3(256), 0, 0, in, n, out, ok); }));
    }
    const u32 threads = 1u << 20, per = 32;
    printf("madd chain %u threads x %u: ", threads, per);
    printf("occ2 %.3f ms  ", time_it([&] { hipLaunchKernelGGL(k_madd<2>, dim3(threads / 256), dim3(256), 0, 0, out, n, per, in); }));
    printf("occ3 %.3f ms  ", time_it([&] { hipLaunchKernelGGL(k_madd<3>, dim3(threads / 256), dim3(256), 0, 0, out, n, per, in); }));
    printf("occ4 %.3f ms\n", time_it([&] { hipLaunchKernelGGL(k_madd<4>, dim3(threads / 256), dim3(256), 0, 0, out, n, per, in); }));
    return 0;
}
