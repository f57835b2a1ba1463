// Gather micro-benchmark: what the memory-side counters report, and what rate the memory system sustains, for the access shape
// of the bucket accumulation (xhe_k_msm.hip, k_msm_accumulate): every lane walks its own run of entries; per entry it requests
// one point record with 16-byte LDS-DMA loads (no register destination), waits for it, reads it out of LDS and goes on.
//
// The table holds n_records records of `stride` bytes (> 256 MiB in total, beyond the 256 MiB on-die cache) and every record is
// named exactly once, in a pseudo-random order, so the bytes a pass must move are known:
//   useful = entries x pieces x 16      lines = entries x (128-byte lines touched per record)
// Kernels (one name each, so that a counter pass lists them apart):
//   gather<6,128>  the accumulation's record: 96 bytes used of a 128-byte record
//   gather<8,128>  the whole 128-byte record
//   gather<4,64>   a 64-byte record (two field elements), two records per 128-byte line
//   stream_lds     coalesced streaming read through the same LDS-DMA instruction, 16 bytes per lane
// Prints one JSON line per kernel: entries, bytes, average time over the timed launches, GB/s.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip
//   tools/ubench_gather [log2_records=22] [entries_per_lane=48] [launches=5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32;
typedef uint64_t u64;

// a bijection of [0, 2^bits): odd multiplications and xor-shifts are each one
__device__ __host__ inline u32 scramble(u32 x, u32 bits) {
    const u32 mask = (bits == 32) ? 0xffffffffu : ((1u << bits) - 1u);
    x = (x * 0x9e3779b1u) & mask;
    x ^= x >> (bits / 2);
    x = (x * 0x85ebca6bu) & mask;
    x ^= x >> (bits / 2 + 1);
    x = (x * 0xc2b2ae35u) & mask;
    return x;
}

__global__ void k_fill_order(u32 *order, u32 n, u32 bits) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = scramble(i, bits);
}
__global__ void k_fill_table(u32 *table, u64 words) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) table[i] = (u32)(i * 2654435761ull);
}

template <int PIECES, int STRIDE>
__global__ void __launch_bounds__(256) gather(const u32 *__restrict__ table, const u32 *__restrict__ order, u32 n_entries, u32 per_lane, u32 *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) u32 stage[4 * PIECES * 256];      // [wave][piece][lane][4 words]
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 *wbase = stage + wave * (PIECES * 256);
    const u32 *mine = wbase + 4 * lane;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u64 start64 = (u64)t * per_lane;
    if (start64 >= n_entries) return;
    const u32 start = (u32)start64, stop = (n_entries - start < per_lane) ? n_entries : start + per_lane;
    auto request = [&](u32 idx) {
        const u32 *src = table + (u64)(STRIDE / 4) * idx;
#pragma unroll
        for (int j = 0; j < PIECES; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * j), (__attribute__((address_space(3))) void *)(wbase + 256 * j), 16, 0, 0);
    };
    u32 acc = 0;
    request(order[start]);
    for (u32 k = start; k < stop; k++) {
        const bool more = k + 1 < stop;
        const u32 next = more ? order[k + 1] : 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint4 v[PIECES];
#pragma unroll
        for (int j = 0; j < PIECES; j++) v[j] = *(const uint4 *)(mine + 256 * j);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (more) request(next);
#pragma unroll
        for (int j = 0; j < PIECES; j++) acc ^= v[j].x + v[j].y + v[j].z + v[j].w;
    }
    out[t] = acc;
}

// coalesced: wave w of the grid reads 1024 consecutive bytes per step through the same instruction
__global__ void __launch_bounds__(256) stream_lds(const u32 *__restrict__ table, u64 n_vec16, u32 steps, u32 *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) u32 stage[4 * 256];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 *wbase = stage + wave * 256;
    const u64 gwave = (u64)blockIdx.x * 4 + wave;
    u32 acc = 0;
    for (u32 s = 0; s < steps; s++) {
        const u64 v = (gwave * steps + s) * 64 + lane;
        if (v >= n_vec16) break;                                   // whole waves leave together: n_vec16 is a multiple of 64
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(table + 4 * v), (__attribute__((address_space(3))) void *)wbase, 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4 x = *(const uint4 *)(wbase + 4 * lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc ^= x.x + x.y + x.z + x.w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename F>
static double timed(F launch, int launches) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch();                                                       // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, 0));
    for (int i = 0; i < launches; i++) launch();
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return ms / launches;
}

static void report(const char *name, u64 entries, u64 useful, u64 lines, double ms, int launches) {
    printf("{\"kernel\": \"%s\", \"entries\": %llu, \"useful_bytes\": %llu, \"line_bytes\": %llu, \"launches_timed\": %d, \"ms\": %.4f, \"useful_GBps\": %.1f, \"line_GBps\": %.1f}\n",
           name, (unsigned long long)entries, (unsigned long long)useful, (unsigned long long)lines, launches, ms, useful / ms / 1e6, lines / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const u32 bits = argc > 1 ? (u32)atoi(argv[1]) : 22;           // 2^22 records x 128 bytes = 512 MiB
    const u32 per_lane = argc > 2 ? (u32)atoi(argv[2]) : 48;
    const int launches = argc > 3 ? atoi(argv[3]) : 5;
    if (bits < 16 || bits > 24 || per_lane == 0 || per_lane > 4096 || launches < 1 || launches > 100) { fprintf(stderr, "arguments out of range\n"); return 2; }
    const u32 n = 1u << bits;
    const u64 table_bytes = (u64)n * 128;
    u32 *table, *order, *order64, *out;
    CHECK(hipMalloc(&table, table_bytes));
    CHECK(hipMalloc(&order, (u64)n * 4));
    CHECK(hipMalloc(&order64, (u64)n * 8));
    const u32 lanes = (2 * n + per_lane - 1) / per_lane;            // the 64-byte form names twice as many records
    const u32 blocks = (lanes + 255) / 256;
    const u64 n_vec16 = table_bytes / 16;
    const u32 stream_steps = 64;
    const u32 stream_blocks = (u32)((n_vec16 / 64 + 4ull * stream_steps - 1) / (4ull * stream_steps));
    const u64 out_words = (u64)256 * (blocks > stream_blocks ? blocks : stream_blocks);
    CHECK(hipMalloc(&out, out_words * 4));
    hipLaunchKernelGGL(k_fill_table, dim3((unsigned)((table_bytes / 4 + 255) / 256)), dim3(256), 0, 0, table, table_bytes / 4);
    hipLaunchKernelGGL(k_fill_order, dim3((n + 255) / 256), dim3(256), 0, 0, order, n, bits);
    hipLaunchKernelGGL(k_fill_order, dim3((2 * n + 255) / 256), dim3(256), 0, 0, order64, 2 * n, bits + 1);
    CHECK(hipDeviceSynchronize());

    // every index below 2^bits names a 128-byte record inside the table, every index below 2^(bits+1) a 64-byte record
    double ms;
    ms = timed([&] { hipLaunchKernelGGL((gather<6, 128>), dim3(blocks), dim3(256), 0, 0, table, order, n, per_lane, out); }, launches);
    report("gather<6,128>", n, (u64)n * 96, (u64)n * 128, ms, launches);
    ms = timed([&] { hipLaunchKernelGGL((gather<8, 128>), dim3(blocks), dim3(256), 0, 0, table, order, n, per_lane, out); }, launches);
    report("gather<8,128>", n, (u64)n * 128, (u64)n * 128, ms, launches);
    ms = timed([&] { hipLaunchKernelGGL((gather<4, 64>), dim3(blocks), dim3(256), 0, 0, table, order64, 2 * n, per_lane, out); }, launches);
    report("gather<4,64>", 2ull * n, (u64)n * 128, (u64)n * 128, ms, launches);
    ms = timed([&] { hipLaunchKernelGGL(stream_lds, dim3(stream_blocks), dim3(256), 0, 0, table, n_vec16, stream_steps, out); }, launches);
    report("stream_lds", n_vec16, table_bytes, table_bytes, ms, launches);
    CHECK(hipGetLastError());
    CHECK(hipFree(table)); CHECK(hipFree(order)); CHECK(hipFree(order64)); CHECK(hipFree(out));
    return 0;
}
