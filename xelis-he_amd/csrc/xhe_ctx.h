// The context behind the C ABI of include/xhe.h, shared by the translation units that hold host code (xhe_lib.hip: verification,
// xhe_prover.hip: batched range-proof generation): device-memory helpers, stage timing, the per-GPU state.
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>
#include "../../include/xhe.h"
#include "xhe_kernels.h"
#include "xhe_ingest.h"

namespace xhe {
void set_error(xhe_ctx *ctx, const std::string &msg);
}

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) { set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_)); return XHE_E_HIP; } \
    } while (0)


using xhe::u8; using xhe::u32; using xhe::u64;

// Device blocks released by one batch are kept for the next: a batch takes two blocks (its uploaded tables, its
// workspace), and in a steady stream of batches of similar size neither costs a hipMalloc / hipFree pair (each of which
// is tens of microseconds and, for hipFree, a device-wide synchronisation).
struct dev_pool {
    std::mutex mu;
    std::vector<std::pair<size_t, void *>> blocks;       // free blocks: capacity, pointer
    size_t held = 0;
    static constexpr size_t HOLD_MAX = 32ull << 30;      // beyond this the blocks go back to the driver
    hipError_t take(size_t bytes, void **p, size_t *cap) {
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = blocks.size();
            for (size_t i = 0; i < blocks.size(); i++)
                if (blocks[i].first >= bytes && blocks[i].first <= 2 * bytes + (1u << 20) && (best == blocks.size() || blocks[i].first < blocks[best].first)) best = i;
            if (best != blocks.size()) {
                *cap = blocks[best].first; *p = blocks[best].second; held -= *cap;
                blocks.erase(blocks.begin() + (long)best);
                return hipSuccess;
            }
        }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(p, want);
        if (e != hipSuccess) {                            // memory held for later may be what is missing now
            drain();
            e = hipMalloc(p, want);
        }
        if (e == hipSuccess) *cap = want;
        return e;
    }
    void give(void *p, size_t cap) {
        std::lock_guard<std::mutex> lk(mu);
        if (held + cap > HOLD_MAX) { (void)hipFree(p); return; }
        blocks.push_back({cap, p}); held += cap;
    }
    void drain() {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &b : blocks) (void)hipFree(b.second);
        blocks.clear(); held = 0;
    }
};

struct dev_buf {
    void *p = nullptr; size_t cap = 0;
    dev_pool *pool = nullptr;                            // where the block comes from and goes back to (null: the driver)
    bool view = false;                                   // a range inside another buffer: nothing to free
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (view) return hipErrorInvalidValue;
        release();
        if (pool) return pool->take(bytes, &p, &cap);
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void point_at(void *base, size_t off, size_t bytes) { p = (char *)base + off; cap = bytes; view = true; }
    void release() {
        if (p && !view) { if (pool) pool->give(p, cap); else (void)hipFree(p); }
        p = nullptr; cap = 0; view = false;
    }
    template <typename T> T *as() const { return (T *)p; }
};

// Stage times from HIP events.  mark() opens a stage on the main stream that lasts until the next mark();
// open()/close() bracket a stage on any stream (stages on different streams overlap in time).
struct stage_timer {
    struct span { const char *name; hipEvent_t a, b; };
    std::vector<std::pair<const char *, hipEvent_t>> marks;
    std::vector<span> spans;
    std::vector<hipEvent_t> pool; size_t used = 0;
    std::vector<const char *> names; std::vector<float> ms;
    int level = 2;                         // 0: nothing, 1: whole verification + the bucket accumulation, 2: every stage
    void begin() { used = 0; marks.clear(); spans.clear(); }
    static bool coarse(const char *name) { return !strcmp(name, "setup") || !strcmp(name, "msm_accumulate") || !strcmp(name, "msm_merge") || !strcmp(name, "end") || !strcmp(name, "decode"); }
    hipError_t take(hipEvent_t *out) {
        if (used == pool.size()) { hipEvent_t e; hipError_t r = hipEventCreate(&e); if (r != hipSuccess) return r; pool.push_back(e); }
        *out = pool[used++];
        return hipSuccess;
    }
    hipError_t mark(const char *name, hipStream_t s) {
        if (level == 0 || (level == 1 && !coarse(name))) return hipSuccess;
        if (level == 1) name = !strcmp(name, "setup") || !strcmp(name, "decode") ? "before_accumulate" : !strcmp(name, "msm_merge") ? "after_accumulate" : name;
        hipEvent_t e; hipError_t r = take(&e); if (r != hipSuccess) return r;
        marks.push_back({name, e});
        return hipEventRecord(e, s);
    }
    hipError_t open(const char *name, hipStream_t s) {
        if (level < 2) return hipSuccess;
        span sp{name, nullptr, nullptr};
        hipError_t r = take(&sp.a); if (r != hipSuccess) return r;
        r = take(&sp.b); if (r != hipSuccess) return r;
        spans.push_back(sp);
        return hipEventRecord(sp.a, s);
    }
    // closes the span of that name (spans on different streams are open at the same time)
    hipError_t close(const char *name, hipStream_t s) {
        if (level < 2) return hipSuccess;
        for (auto &sp : spans) if (!strcmp(sp.name, name)) return hipEventRecord(sp.b, s);
        return hipErrorInvalidValue;
    }
    // after every stream is synchronised
    void finish() {
        names.clear(); ms.clear();
        for (size_t i = 0; i + 1 < marks.size(); i++) {
            float t = 0; if (hipEventElapsedTime(&t, marks[i].second, marks[i + 1].second) != hipSuccess) t = -1;
            names.push_back(marks[i].first); ms.push_back(t);
        }
        for (auto &sp : spans) {
            float t = 0; if (hipEventElapsedTime(&t, sp.a, sp.b) != hipSuccess) t = -1;
            names.push_back(sp.name); ms.push_back(t);
        }
        if (marks.size() >= 2) {
            float t = 0; if (hipEventElapsedTime(&t, marks.front().second, marks.back().second) != hipSuccess) t = -1;
            names.push_back("total"); ms.push_back(t);
        }
    }
    void release() { for (auto e : pool) (void)hipEventDestroy(e); pool.clear(); }
};

struct msm_workspace { dev_buf hist, offsets, tile_hist, sorted, buckets, digits, chunk_sums, big_list, segs, wins, tmp, coarse; };

struct xhe_ctx {
    int device = 0;
    hipStream_t stream = nullptr;                 // main stream: transcript, proof scalars, multiscalar check, verdict
    hipStream_t side[3] = {nullptr, nullptr, nullptr};            // [0] ledger scan + static-generator scalars, [1] point decoding (+ signatures), [2] signatures of
                                                                  // large batches only; all at the default priority
    hipEvent_t ev_keys = nullptr, ev_static_done = nullptr, ev_setup = nullptr, ev_scalars = nullptr, ev_points_checked = nullptr, ev_early = nullptr, ev_side_done[3] = {nullptr, nullptr, nullptr};
    hipStream_t copy_stream = nullptr;            // host-to-device copies of the uploads only (no kernels): an upload may run beside a verification
    hipEvent_t ev_stage_free = nullptr;           // fires when the last copy out of the pinned staging memory has finished
    std::string error; std::mutex err_mu;
    // held shared by everything that reads the constant tables on the device (verification, apply, proving), exclusively while
    // ensure_tables rebuilds them for a larger capacity
    std::shared_mutex tables_mu;
    stage_timer timer;
    dev_pool pool;                 // blocks of freed batches
    xhe::host_batch spare_host; std::mutex spare_mu;   // host tables of a freed batch, kept for their memory
    xhe::ingest_workers workers;   // host threads of the uploads' parsing (started by the first upload)
    xhe::ingest_arena arena;       // memory of an upload's temporary host tables, kept between uploads
    void *stage = nullptr; size_t stage_cap = 0;     // pinned host memory the uploads are assembled in
    u32 *verdict_host = nullptr;                     // pinned memory the verdict words are read back into: a ring of 64-byte slots
    u32 verdict_next = 0;                            // the next slot of that ring
    u8 *readback[2] = {nullptr, nullptr}; size_t readback_cap[2] = {0, 0};   // pinned memory two batches' resulting ciphertexts are read back into (xhe_verify_batches)
    // level-0 workspace
    dev_buf in_bytes, scalars, niels, ok, result, xyzt;
    msm_workspace msm;
    msm_workspace msm_odd;      // the bucket workspace of every second part of a batch verified in parts (see batch_verify_enqueue, `lane`)
    // constant tables
    dev_buf base_pts;          // B, H as affine Niels
    dev_buf gtab, htab;        // fixed-base tables: 8 x 256 multiples of G, 32 x 256 multiples of H
    dev_buf static_pts;        // B, H, G_vec[cap], H_vec[cap]
    u32 static_cap = 0;        // generators per side currently held
    dev_buf gen_tab;           // 32 x 256 fixed-base table of every static point (static_cap <= GEN_TABLE_CAP_MAX), else empty
    u32 gen_tab_cap = 0;       // the capacity gen_tab was built for, 0 = none
    bool tables_ready = false;
    unsigned char *static_scratch = nullptr;   // this device's copy of `scratch`
    bool stats_on = false; xhe_msm_stats stats{0, 0, 0, 0.0};      // xhe_msm_stats_enable / xhe_msm_stats_read
};


namespace xhe {
// B/H points, fixed-base tables, generators up to `cap` per side (a multiple of 64); see xhe_lib.hip
int32_t ensure_tables(xhe_ctx *ctx, u32 cap);
}
