// libxhe.so: the C ABI of include/xhe.h over the gfx950 kernels.  Host side is orchestration only:
// it owns the HIP stream, the workspace and the launch sequence; all arithmetic on the verification
// path runs in the kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/xhe.h"
#include "xhe_msm.hip.h"

using namespace xhe;

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) { set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_)); return XHE_E_HIP; } \
    } while (0)

static std::string g_create_error;

struct dev_buf {
    void *p = nullptr; size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() { return (T *)p; }
};

struct stage_timer {
    std::vector<std::pair<const char *, hipEvent_t>> marks;
    std::vector<hipEvent_t> pool; size_t used = 0;
    std::vector<const char *> names; std::vector<float> ms;
    void begin() { used = 0; marks.clear(); }
    hipError_t mark(const char *name, hipStream_t s) {
        if (used == pool.size()) { hipEvent_t e; hipError_t r = hipEventCreate(&e); if (r != hipSuccess) return r; pool.push_back(e); }
        hipEvent_t e = pool[used++];
        marks.push_back({name, e});
        return hipEventRecord(e, s);
    }
    // call after the stream is synchronised: stage i spans mark i -> mark i+1 and carries mark i's name
    void finish() {
        names.clear(); ms.clear();
        for (size_t i = 0; i + 1 < marks.size(); i++) {
            float t = 0; if (hipEventElapsedTime(&t, marks[i].second, marks[i + 1].second) != hipSuccess) t = -1;
            names.push_back(marks[i].first); ms.push_back(t);
        }
    }
    void release() { for (auto e : pool) (void)hipEventDestroy(e); pool.clear(); }
};

struct xhe_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    stage_timer timer;
    // level-0 / MSM workspace
    dev_buf in_bytes, scalars, niels, ok, hist, offsets, cursor, sorted, buckets, segs, wins, result, xyzt;
};

static void set_error(xhe_ctx *ctx, const std::string &msg) { if (ctx) ctx->error = msg; else g_create_error = msg; }

// ---------------------------------------------------------------------------------------------------
// level-0 kernels
namespace xhe {

__global__ void __launch_bounds__(256, 2) k_decode_xyzt(const u32 *__restrict__ in, u64 n, u32 *__restrict__ out, u8 *__restrict__ ok) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = in[8 * i + k];
    ge p; bool good = ristretto_decode_words(p, w);
    ge_store(out + 32 * i, p);
    ok[i] = good ? 1 : 0;
}
__global__ void __launch_bounds__(256, 2) k_decode_niels(const u32 *__restrict__ in, u64 n, u32 *__restrict__ out, u8 *__restrict__ ok) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = in[8 * i + k];
    ge p; bool good = ristretto_decode_words(p, w);
    ge_niels_store(out + 32 * i, ge_to_niels_affine(p));
    ok[i] = good ? 1 : 0;
}
__global__ void __launch_bounds__(256, 2) k_encode_xyzt(const u32 *__restrict__ in, u64 n, u32 *__restrict__ out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ge p = ge_load(in + 32 * i);
    u32 w[8]; ristretto_encode_words(w, p);
#pragma unroll
    for (int k = 0; k < 8; k++) out[8 * i + k] = w[k];
}
// 32-byte little-endian integers -> canonical scalars (value mod l), in place
__global__ void __launch_bounds__(256) k_reduce_scalars(u32 *__restrict__ s, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = s[8 * i + k];
    sc x = sc_from_words(w);
    sc_to_words(w, x);
#pragma unroll
    for (int k = 0; k < 8; k++) s[8 * i + k] = w[k];
}
// ok-flag reduction: out[0] = number of zero flags
__global__ void k_count_bad(const u8 *__restrict__ ok, u64 n, u32 *__restrict__ out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !ok[i]) atomicAdd(out, 1u);
}

}  // namespace xhe

static inline unsigned grid_for(u64 n, unsigned block) { return (unsigned)((n + block - 1) / block); }

// MSM over device-resident canonical scalars and affine-Niels points; result (8 words encoding, flag)
// into d_result[0..8], d_result[8].  Stage marks are recorded when timing is on.
static int32_t msm_device(xhe_ctx *ctx, const u32 *d_scalars, const u32 *d_niels, u64 n, u32 *d_result, bool timing) {
    hipStream_t st = ctx->stream;
    msm_plan p = msm_make_plan(n);
    u64 n_buckets = (u64)p.W * p.B;
    u64 sorted_cap = n * p.W;
    if (sorted_cap == 0) sorted_cap = 1;
    HIPCHK(ctx, ctx->hist.reserve(n_buckets * 4));
    HIPCHK(ctx, ctx->offsets.reserve((n_buckets + 1) * 4));
    HIPCHK(ctx, ctx->cursor.reserve(n_buckets * 4));
    HIPCHK(ctx, ctx->sorted.reserve(sorted_cap * 4));
    HIPCHK(ctx, ctx->buckets.reserve(n_buckets * 128));
    HIPCHK(ctx, ctx->segs.reserve((u64)p.W * p.S * 2 * 128));
    HIPCHK(ctx, ctx->wins.reserve((u64)p.W * 128));
    if (timing) HIPCHK(ctx, ctx->timer.mark("msm_histogram", st));
    HIPCHK(ctx, hipMemsetAsync(ctx->hist.p, 0, n_buckets * 4, st));
    if (n) hipLaunchKernelGGL(k_msm_histogram, dim3(grid_for(n, 256)), dim3(256), 0, st, d_scalars, n, p, ctx->hist.as<u32>());
    if (timing) HIPCHK(ctx, ctx->timer.mark("msm_scan", st));
    hipLaunchKernelGGL(k_msm_scan, dim3(1), dim3(1024), 0, st, ctx->hist.as<u32>(), (u32)n_buckets, ctx->offsets.as<u32>(), ctx->cursor.as<u32>());
    if (timing) HIPCHK(ctx, ctx->timer.mark("msm_scatter", st));
    if (n) hipLaunchKernelGGL(k_msm_scatter, dim3(grid_for(n, 256)), dim3(256), 0, st, d_scalars, n, p, ctx->cursor.as<u32>(), ctx->sorted.as<u32>(), sorted_cap);
    if (timing) HIPCHK(ctx, ctx->timer.mark("msm_accumulate", st));
    hipLaunchKernelGGL(k_msm_accumulate, dim3(grid_for(n_buckets, 256)), dim3(256), 0, st, d_niels, n, ctx->offsets.as<u32>(), ctx->sorted.as<u32>(),
                       (u32)n_buckets, ctx->buckets.as<u32>());
    if (timing) HIPCHK(ctx, ctx->timer.mark("msm_reduce", st));
    hipLaunchKernelGGL(k_msm_reduce_segments, dim3(p.W * p.S), dim3(64), 0, st, ctx->buckets.as<u32>(), p, ctx->segs.as<u32>());
    hipLaunchKernelGGL(k_msm_reduce_windows, dim3(p.W), dim3(64), 0, st, ctx->segs.as<u32>(), p, ctx->wins.as<u32>());
    if (timing) HIPCHK(ctx, ctx->timer.mark("msm_horner", st));
    hipLaunchKernelGGL(k_msm_horner, dim3(1), dim3(64), 0, st, ctx->wins.as<u32>(), p, d_result, d_result + 8, (u32 *)nullptr);
    if (timing) HIPCHK(ctx, ctx->timer.mark("end", st));
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

extern "C" {

xhe_ctx *xhe_create(int device_id) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) { g_create_error = "no HIP device available"; return nullptr; }
    if (device_id < 0 || device_id >= count) { g_create_error = "device ordinal out of range"; return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { g_create_error = "hipGetDeviceProperties failed"; return nullptr; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { g_create_error = std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only"; return nullptr; }
    if (hipSetDevice(device_id) != hipSuccess) { g_create_error = "hipSetDevice failed"; return nullptr; }
    xhe_ctx *ctx = new xhe_ctx();
    ctx->device = device_id;
    if (hipStreamCreate(&ctx->stream) != hipSuccess) { g_create_error = "hipStreamCreate failed"; delete ctx; return nullptr; }
    return ctx;
}

void xhe_destroy(xhe_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (dev_buf *b : {&ctx->in_bytes, &ctx->scalars, &ctx->niels, &ctx->ok, &ctx->hist, &ctx->offsets, &ctx->cursor, &ctx->sorted, &ctx->buckets,
                       &ctx->segs, &ctx->wins, &ctx->result, &ctx->xyzt})
        b->release();
    ctx->timer.release();
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *xhe_last_error(const xhe_ctx *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int32_t xhe_last_stage_times(const xhe_ctx *ctx, const char **names, float *ms, int32_t cap) {
    if (!ctx) return XHE_E_ARG;
    int32_t n = (int32_t)ctx->timer.names.size();
    for (int32_t i = 0; i < n && i < cap; i++) { if (names) names[i] = ctx->timer.names[i]; if (ms) ms[i] = ctx->timer.ms[i]; }
    return n;
}

int32_t xhe_ristretto_decompress_batch(xhe_ctx *ctx, uint64_t n, const uint8_t *in32, uint8_t *out_xyzt128, uint8_t *ok) {
    if (!ctx || (n && (!in32 || !out_xyzt128 || !ok))) return XHE_E_ARG;
    if (n == 0) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, ctx->in_bytes.reserve(n * 32));
    HIPCHK(ctx, ctx->xyzt.reserve(n * 128));
    HIPCHK(ctx, ctx->ok.reserve(n));
    HIPCHK(ctx, hipMemcpyAsync(ctx->in_bytes.p, in32, n * 32, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_decode_xyzt, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, ctx->in_bytes.as<u32>(), n, ctx->xyzt.as<u32>(), ctx->ok.as<u8>());
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out_xyzt128, ctx->xyzt.p, n * 128, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ok, ctx->ok.p, n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int32_t xhe_ristretto_compress_batch(xhe_ctx *ctx, uint64_t n, const uint8_t *in_xyzt128, uint8_t *out32) {
    if (!ctx || (n && (!in_xyzt128 || !out32))) return XHE_E_ARG;
    if (n == 0) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, ctx->xyzt.reserve(n * 128));
    HIPCHK(ctx, ctx->in_bytes.reserve(n * 32));
    HIPCHK(ctx, hipMemcpyAsync(ctx->xyzt.p, in_xyzt128, n * 128, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_encode_xyzt, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, ctx->xyzt.as<u32>(), n, ctx->in_bytes.as<u32>());
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out32, ctx->in_bytes.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int32_t xhe_msm_vartime(xhe_ctx *ctx, uint64_t n, const uint8_t *scalars32, const uint8_t *points32, uint8_t out32[32], uint8_t *is_identity) {
    if (!ctx || !out32 || (n && (!scalars32 || !points32))) return XHE_E_ARG;
    if (n > 0x7fffffffull / 40) return XHE_E_ARG;       // term index carries a sign bit; n*W must fit 32 bits
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    u64 nn = n ? n : 1;
    HIPCHK(ctx, ctx->in_bytes.reserve(nn * 32));
    HIPCHK(ctx, ctx->scalars.reserve(nn * 32));
    HIPCHK(ctx, ctx->niels.reserve(nn * 128));
    HIPCHK(ctx, ctx->ok.reserve(nn));
    HIPCHK(ctx, ctx->result.reserve(64));
    ctx->timer.begin();
    HIPCHK(ctx, ctx->timer.mark("decode", st));
    HIPCHK(ctx, hipMemsetAsync(ctx->result.p, 0, 64, st));
    if (n) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->in_bytes.p, points32, n * 32, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(ctx->scalars.p, scalars32, n * 32, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_decode_niels, dim3(grid_for(n, 256)), dim3(256), 0, st, ctx->in_bytes.as<u32>(), n, ctx->niels.as<u32>(), ctx->ok.as<u8>());
        hipLaunchKernelGGL(k_count_bad, dim3(grid_for(n, 256)), dim3(256), 0, st, ctx->ok.as<u8>(), n, ctx->result.as<u32>() + 12);
        hipLaunchKernelGGL(k_reduce_scalars, dim3(grid_for(n, 256)), dim3(256), 0, st, ctx->scalars.as<u32>(), n);
    }
    int32_t rc = msm_device(ctx, ctx->scalars.as<u32>(), ctx->niels.as<u32>(), n, ctx->result.as<u32>(), true);
    if (rc) return rc;
    u32 host[16];
    HIPCHK(ctx, hipMemcpyAsync(host, ctx->result.p, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    ctx->timer.finish();
    if (host[12] != 0) { set_error(ctx, "a compressed point does not decode"); return XHE_E_DECODE; }
    memcpy(out32, host, 32);
    if (is_identity) *is_identity = (uint8_t)host[8];
    return 0;
}

}  // extern "C"
