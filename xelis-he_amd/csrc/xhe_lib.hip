// libxhe.so: the C ABI of include/xhe.h over the gfx950 kernels.  The host side is orchestration: it owns
// the HIP stream, the device buffers and the launch sequence, parses records and resolves ledger keys
// (xhe_ingest.h); all field, group and hash arithmetic of the verification path runs in the kernels.
#include <hip/hip_runtime.h>
#include "xhe_settings.h"
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/xhe.h"
#include <algorithm>
#include "xhe_ingest.h"
#include "xhe_kernels.h"
#include "xhe_ctx.h"
#include "xhe_ge.h"

using namespace xhe;

static std::string g_create_error;

// Memory of xhe_msm_vartime_static: the staging buffers and the whole workspace of one multiscalar multiplication of at
// most XHE_MSM_STATIC_MAX_TERMS terms.  Every GPU the library runs on holds its own copy until the process ends; host
// code reaches it only through the address hipGetSymbolAddress returns for the current device (xhe_create).
static constexpr size_t MSM_STATIC_BYTES = 12ull << 20;
static __device__ __align__(256) unsigned char scratch[MSM_STATIC_BYTES];

static constexpr size_t STAGE_MAX = 1ull << 30;      // pinned staging memory per context, at most
// xhe_verify_batch cuts a batch into parts (verify_batch_split below) by the size of its records and by the number of its
// transactions, whichever asks for more, at most SPLIT_PARTS_MAX: the host's ordered walk costs per transaction, the copies and the
// device's work per byte.  Every part adds fixed host work to the one uploader thread and a latency-bound head and tail on the
// device (0.6-0.8 ms).  Measured, one call from host buffers: 100 000 transactions of 8 transfers (390 MB) in one piece 39-41 ms,
// 4 parts 31 ms, 6 parts 25.5-26 ms, 8 parts 28 ms; 100 000 mixed transactions (100 MB) in 3 parts 27.0 ms, 5-6 parts 24.0 ms;
// 125 000 one-transfer transactions (170 MB) in 4 parts 17.2 ms, 5-6 parts 16.2 ms; 40 000 of those in one piece 7.7 ms, 3 parts
// 6.6 ms; 20 000 transactions of 8 transfers (78 MB) in one piece 8.0 ms, 2-3 parts 7.5 ms; 20 000 and fewer one-transfer
// transactions are fastest in one piece; 32 768 of them in one piece 6.0 ms, 2 parts 5.6 ms, 4 parts 6.6 ms; 65 536 in 2 parts
// 9.9 ms, 4 parts 9.1 ms, 5 parts 9.7 ms, 6 parts 10.3 ms.
static constexpr u64 SPLIT_PART_BYTES = 32ull << 20;
static constexpr u64 SPLIT_PART_TXS = 16000;
static constexpr u64 SPLIT_PARTS_MAX = 6;
static constexpr u32 VERDICT_SLOTS = 4;              // verifications whose verdict words may be on their way to the host at once
// From this batch size on the signatures get a stream of their own and start beside the transcripts (measured on the
// 10 000-transaction batch: 1.53 ms against 1.59 ms with the signatures behind the bulk decode).  Smaller batches are pure
// chains of short kernels and keep to three streams.  XHE_SIG_STREAM_MIN_TXS overrides it (the tests run either schedule on
// small batches).
static u32 sig_stream_min_txs() { const char *e = setting("XHE_SIG_STREAM_MIN_TXS"); return e ? (u32)strtoul(e, nullptr, 10) : 2048u; }

struct xhe_batch {
    host_batch hb;             // host tables (kept for reading the ledger back)
    u64 n_balances = 0, n_accounts = 0;
    std::vector<u8> in_cts; std::vector<u64> in_nonces;
    dev_buf ex_txs, ex_xfs, ex_cms, ex_mss, sh_decs, sh_early, sh_key, ex_slot_pos, ex_slot_base;     // inputs of the descriptor expansion
    dev_buf merge_pairs, key_terms, key_group_off, bytes, txs, cms, xfers, decs, early_decs, key_decs, bulk_decs, sig_keys, sig_tab, ops, sigs, tx_err0, apply_err0, slot_last_op, emit_slot, ms_hash_tx, group_txs, tmpl, segs, serial_txs, tr_list, tr_tiles, tr_stream, sig_state, sig_r;
    u64 rp_gemm_min_rows = ~0ull;
    dev_buf pts, ok, chal, scalars, term_point, gh, gh_sums, rp_ctx, rp_acc, rp_mat, rp_dig, rp_bins, rp_rho, tx_err, rp_err, ms_hash;
    dev_buf thread_excl, block_tot, slot_base, slot_final, emit_incl, final_cts, verdict, seed;
    dev_buf static_partials, static_sum;       // the static generators' contribution to the multiscalar check (from the fixed-base tables)
    dev_buf slot_of_used;                      // used slot -> index of its balance in the ledger view
    dev_buf noncanonical;                      // index of the first transaction with a scalar that is not below the group order (the upload's device check), else 0xffffffff
    dev_buf arena_up, arena_ws, arena_bytes;   // the blocks every buffer above is a range of: uploaded tables, workspace, the byte buffer
    std::vector<u32> group_off;   // per lg: offset into group_txs
    hipEvent_t ev_uploaded = nullptr;   // fires on the copy stream when every table of the batch has arrived
    hipEvent_t ev_done = nullptr;       // fires on the main stream when a verification's verdict words have been read back
    hipEvent_t ev_carried = nullptr;    // fires on side stream 0 when the batch's resulting ciphertexts have been handed on (batches verified in parts)
    hipEvent_t ev_acc[2] = {nullptr, nullptr};   // around the bucket accumulation kernel of the verification in flight (xhe_msm_stats)
    bool acc_timed = false;
    u32 *verdict_words = nullptr;       // where in the context's pinned memory the verdict words of the verification in flight arrive
    bool in_flight = false;             // a verification has been enqueued and not yet finished
    u32 lane = 0;                       // ... on which pair of streams (batch_verify_enqueue)
    hipStream_t main_stream = nullptr, ledger_stream = nullptr;      // that verification's main stream, and the one its ledger scan ran on
    bool trivial = false;               // ... of an empty batch
    std::shared_lock<std::shared_mutex> tables_lock;     // held from a verification's first launch to its finish: its kernels read the context's tables
    bool verified_ok = false;
    void release() {
        for (dev_buf *b : {&static_partials, &static_sum, &slot_of_used}) b->release();
        for (hipEvent_t *e : {&ev_done, &ev_carried, &ev_acc[0], &ev_acc[1]}) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
        for (dev_buf *b : {&ex_txs, &ex_xfs, &ex_cms, &ex_mss, &sh_decs, &sh_early, &sh_key, &ex_slot_pos, &ex_slot_base, &merge_pairs, &key_terms, &key_group_off, &bytes, &txs, &cms, &xfers, &decs, &early_decs, &key_decs, &bulk_decs, &sig_keys, &sig_tab, &ops, &sigs, &tx_err0, &apply_err0, &slot_last_op, &emit_slot, &ms_hash_tx, &group_txs, &tmpl, &segs, &serial_txs, &tr_list, &tr_tiles, &tr_stream, &sig_state, &sig_r, &pts, &ok, &chal,
                           &scalars, &term_point, &gh, &gh_sums, &rp_ctx, &rp_acc, &rp_mat, &rp_dig, &rp_bins, &rp_rho, &tx_err, &rp_err, &ms_hash, &thread_excl, &block_tot, &slot_base,
                           &slot_final, &emit_incl, &final_cts, &verdict, &seed, &noncanonical, &arena_up, &arena_ws, &arena_bytes})
            b->release();
        if (ev_uploaded) { (void)hipEventDestroy(ev_uploaded); ev_uploaded = nullptr; }
    }
};

void xhe::set_error(xhe_ctx *ctx, const std::string &msg) { if (ctx) { std::lock_guard<std::mutex> lk(ctx->err_mu); ctx->error = msg; } else g_create_error = msg; }

// MSM over device-resident canonical scalars and affine-Niels records; result (8 words encoding, flag)
// into d_result[0..8], d_result[8].
struct msm_mark_target { xhe_ctx *ctx; hipStream_t stream; bool timing; hipEvent_t *acc_events; };
static hipError_t timer_mark(void *user, const char *name) {
    msm_mark_target *t = (msm_mark_target *)user;
    // the pair of events around the bucket accumulation kernel (xhe_msm_stats): "msm_merge" is the stage behind it
    if (t->acc_events) {
        hipError_t e = !strcmp(name, "msm_accumulate") ? hipEventRecord(t->acc_events[0], t->stream) : !strcmp(name, "msm_merge") ? hipEventRecord(t->acc_events[1], t->stream) : hipSuccess;
        if (e != hipSuccess) return e;
    }
    return t->timing ? t->ctx->timer.mark(name, t->stream) : hipSuccess;
}

// lane 1: on the stream and in the workspace of every second part of a batch verified in parts (batch_verify_enqueue)
static int32_t msm_device(xhe_ctx *ctx, const u32 *d_scalars, const u32 *d_niels, const u32 *d_term_point, u64 n_points, u64 n, u32 *d_result, bool want_encoding,
                          bool timing, const msm_deps &deps = msm_deps(), hipStream_t stream = nullptr, u32 lane = 0, hipEvent_t *acc_events = nullptr) {
    msm_workspace &ws = lane ? ctx->msm_odd : ctx->msm;
    if (!stream) stream = ctx->stream;
    msm_mark_target target{ctx, stream, timing, acc_events};
    msm_plan p = msm_make_plan(n);
    if (n * p.W > 0xfffffff0ull) { set_error(ctx, "multiscalar multiplication too large for 32-bit positions"); return XHE_E_ARG; }
    msm_ws_bytes need = msm_workspace_bytes(n, p);
    HIPCHK(ctx, ws.hist.reserve(need.hist));
    HIPCHK(ctx, ws.offsets.reserve(need.offsets));
    HIPCHK(ctx, ws.tile_hist.reserve(need.tile_hist));
    HIPCHK(ctx, ws.digits.reserve(need.digits));
    HIPCHK(ctx, ws.sorted.reserve(need.sorted));
    HIPCHK(ctx, ws.buckets.reserve(need.buckets));
    HIPCHK(ctx, ws.chunk_sums.reserve(need.chunk_sums));
    HIPCHK(ctx, ws.big_list.reserve(need.big_list));
    HIPCHK(ctx, ws.segs.reserve(need.segs));
    HIPCHK(ctx, ws.wins.reserve(need.wins));
    HIPCHK(ctx, ws.tmp.reserve(need.tmp));
    HIPCHK(ctx, ws.coarse.reserve(need.coarse));
    msm_ws w{ws.hist.as<u32>(), ws.offsets.as<u32>(), ws.tile_hist.as<u32>(), ws.sorted.as<u32>(), ws.buckets.as<u32>(), ws.digits.as<u32>(), ws.chunk_sums.as<u32>(),
             ws.big_list.as<u32>(), ws.segs.as<u32>(), ws.wins.as<u32>(), ws.tmp.as<u32>(), ws.coarse.as<u32>()};
    HIPCHK(ctx, msm_launch(stream, d_scalars, d_niels, d_term_point, n_points, n, p, w, d_result, want_encoding, deps, (timing || acc_events) ? timer_mark : nullptr, &target, acc_events != nullptr));
    return 0;
}

// B/H points, fixed-base tables, generators up to `cap` per side
int32_t xhe::ensure_tables(xhe_ctx *ctx, u32 cap) {
    {
        std::shared_lock<std::shared_mutex> rd(ctx->tables_mu);
        if (ctx->tables_ready && cap <= ctx->static_cap && std::min<u32>(ctx->static_cap, GEN_TABLE_CAP_MAX) <= ctx->gen_tab_cap) return 0;
    }
    std::unique_lock<std::shared_mutex> wr(ctx->tables_mu);      // waits for the verifications in flight: they read the tables
    hipStream_t st = ctx->stream;
    if (!ctx->tables_ready) {
        HIPCHK(ctx, ctx->base_pts.reserve(2 * 128));
        HIPCHK(ctx, ctx->gtab.reserve(8 * 256 * 128));
        HIPCHK(ctx, ctx->htab.reserve(32 * 256 * 128));
        launch_base_points(st, ctx->base_pts.as<u32>());
        launch_build_fixed_table(st, ctx->base_pts.as<u32>(), 1u, 8u, ctx->gtab.as<u32>());
        launch_build_fixed_table(st, ctx->base_pts.as<u32>() + 32, 1u, 32u, ctx->htab.as<u32>());
        HIPCHK(ctx, hipGetLastError());
        ctx->tables_ready = true;
    }
    if (cap > ctx->static_cap) {
        dev_buf fresh;
        HIPCHK(ctx, fresh.reserve((2ull + 2ull * cap) * 128));
        HIPCHK(ctx, hipMemcpyAsync(fresh.p, ctx->base_pts.p, 256, hipMemcpyDeviceToDevice, st));
        // generators are regenerated for the new layout (one-off per capacity growth)
        launch_bp_generators(st, cap / 64, cap, fresh.as<u32>());
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));
        ctx->static_pts.release();
        ctx->static_pts = fresh;
        ctx->static_cap = cap;
    }
    // fixed-base tables of the static points (up to GEN_TABLE_CAP_MAX generators per side): the static part
    // of the multiscalar check becomes look-ups
    u32 tcap = std::min<u32>(ctx->static_cap, GEN_TABLE_CAP_MAX);
    if (tcap > ctx->gen_tab_cap) {
        ctx->gen_tab.release(); ctx->gen_tab_cap = 0;
        u64 n_bases = 2ull + 2ull * tcap;
        HIPCHK(ctx, ctx->gen_tab.reserve(n_bases * 32 * 256 * 128));
        launch_build_fixed_table(st, ctx->static_pts.as<u32>(), 2 + tcap, 32u, ctx->gen_tab.as<u32>());                      // B, H, G_vec[0..tcap)
        launch_build_fixed_table(st, ctx->static_pts.as<u32>() + 32ull * (2 + ctx->static_cap), tcap, 32u,
                                 ctx->gen_tab.as<u32>() + 32ull * (2ull + tcap) * 32 * 256);                                    // H_vec[0..tcap)
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));
        ctx->gen_tab_cap = tcap;
    }
    return 0;
}

static int32_t batch_verify_impl(xhe_ctx *ctx, xhe_batch *B, xhe_verdict *verdict, const uint8_t seed[32]);
static constexpr int32_t ENQUEUE_TABLES_BUSY = 2;
// The ledger a part of a batch verified in parts starts from: the call's ledger array on the device, valid once `after` has fired
// (null: valid already).  The part copies it into its byte buffer on its ledger stream, in front of the decode of the ledger's
// slots - the only work of the part that has to wait for the part in front.
struct ledger_handover { const void *src; size_t bytes; hipEvent_t after; };
static int32_t batch_verify_enqueue(xhe_ctx *ctx, xhe_batch *B, const uint8_t seed[32], bool setup_on_side, bool may_block = true, int lane_or_none = -1,
                                    u32 *carry_ledger = nullptr, const ledger_handover *handover = nullptr);
static int32_t batch_verify_finish(xhe_ctx *ctx, xhe_batch *B, xhe_verdict *verdict);

extern "C" {

xhe_ctx *xhe_create(int device_id) {
    // the XHE_* switches are read here, by the first context of the process (xhe_settings.h)
    static std::once_flag settings_once;
    std::call_once(settings_once, [] { settings_load(); });
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
    bool ok = hipStreamCreate(&ctx->stream) == hipSuccess && hipStreamCreate(&ctx->side[0]) == hipSuccess && hipStreamCreate(&ctx->side[1]) == hipSuccess && hipStreamCreate(&ctx->side[2]) == hipSuccess &&
              hipStreamCreate(&ctx->copy_stream) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->ev_stage_free, hipEventDisableTiming) == hipSuccess && hipEventRecord(ctx->ev_stage_free, ctx->copy_stream) == hipSuccess;
    for (hipEvent_t *e : {&ctx->ev_keys, &ctx->ev_static_done, &ctx->ev_setup, &ctx->ev_scalars, &ctx->ev_points_checked, &ctx->ev_early, &ctx->ev_side_done[0], &ctx->ev_side_done[1], &ctx->ev_side_done[2]})
        ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&ctx->verdict_host, 64 * VERDICT_SLOTS, hipHostMallocDefault) == hipSuccess;
    // the address of this device's copy of the static scratch array (the array's name itself is a host-side stand-in)
    if (ok && hipGetSymbolAddress((void **)&ctx->static_scratch, HIP_SYMBOL(scratch)) != hipSuccess) { ctx->static_scratch = nullptr; ok = false; }
    if (!ok) { g_create_error = "creating the HIP streams, events and read-back buffer failed"; xhe_destroy(ctx); return nullptr; }
    return ctx;
}

void xhe_destroy(xhe_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (hipStream_t s : {ctx->stream, ctx->side[0], ctx->side[1], ctx->side[2], ctx->copy_stream}) if (s) (void)hipStreamSynchronize(s);
    for (dev_buf *b : {&ctx->in_bytes, &ctx->scalars, &ctx->niels, &ctx->ok, &ctx->result, &ctx->xyzt, &ctx->msm.hist, &ctx->msm.offsets, &ctx->msm.tile_hist, &ctx->msm.digits,
                       &ctx->msm.sorted, &ctx->msm.buckets, &ctx->msm.chunk_sums, &ctx->msm.big_list, &ctx->msm.segs, &ctx->msm.wins, &ctx->msm.tmp, &ctx->msm.coarse,
                       &ctx->msm_odd.hist, &ctx->msm_odd.offsets, &ctx->msm_odd.tile_hist, &ctx->msm_odd.digits, &ctx->msm_odd.sorted, &ctx->msm_odd.buckets, &ctx->msm_odd.chunk_sums, &ctx->msm_odd.big_list,
                       &ctx->msm_odd.segs, &ctx->msm_odd.wins, &ctx->msm_odd.tmp, &ctx->msm_odd.coarse, &ctx->base_pts, &ctx->gtab, &ctx->htab, &ctx->static_pts, &ctx->gen_tab})
        b->release();
    ctx->pool.drain();
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    if (ctx->verdict_host) (void)hipHostFree(ctx->verdict_host);
    for (u8 *p : ctx->readback) if (p) (void)hipHostFree(p);
    ctx->timer.release();
    for (hipEvent_t e : {ctx->ev_keys, ctx->ev_static_done, ctx->ev_setup, ctx->ev_scalars, ctx->ev_points_checked, ctx->ev_early, ctx->ev_side_done[0], ctx->ev_side_done[1], ctx->ev_side_done[2]}) if (e) (void)hipEventDestroy(e);
    if (ctx->ev_stage_free) (void)hipEventDestroy(ctx->ev_stage_free);
    for (hipStream_t s : {ctx->stream, ctx->side[0], ctx->side[1], ctx->side[2], ctx->copy_stream}) if (s) (void)hipStreamDestroy(s);
    delete ctx;
}

const char *xhe_last_error(const xhe_ctx *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int32_t xhe_set_stage_timing(xhe_ctx *ctx, int32_t level) {
    if (!ctx || level < 0 || level > 2) return XHE_E_ARG;
    ctx->timer.level = level;
    return 0;
}

int32_t xhe_msm_stats_enable(xhe_ctx *ctx, int32_t on) {
    if (!ctx) return XHE_E_ARG;
    ctx->stats_on = on != 0;
    return 0;
}
int32_t xhe_msm_stats_read(xhe_ctx *ctx, xhe_msm_stats *out, int32_t reset) {
    if (!ctx || !out) return XHE_E_ARG;
    *out = ctx->stats;
    if (reset) ctx->stats = xhe_msm_stats{0, 0, 0, 0.0};
    return 0;
}

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
    launch_decode_xyzt(ctx->stream, ctx->in_bytes.as<u32>(), n, ctx->xyzt.as<u32>(), ctx->ok.as<u8>());
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
    launch_encode_xyzt(ctx->stream, ctx->xyzt.as<u32>(), n, ctx->in_bytes.as<u32>());
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
        launch_decode_niels(st, ctx->in_bytes.as<u32>(), n, ctx->niels.as<u32>(), ctx->ok.as<u8>());
        launch_count_bad(st, ctx->ok.as<u8>(), n, ctx->result.as<u32>() + 12);
        launch_reduce_scalars(st, ctx->scalars.as<u32>(), n);
    }
    int32_t rc = msm_device(ctx, ctx->scalars.as<u32>(), ctx->niels.as<u32>(), nullptr, n, n, ctx->result.as<u32>(), true, true);
    if (rc) return rc;
    HIPCHK(ctx, ctx->timer.mark("end", st));
    u32 host[16];
    HIPCHK(ctx, hipMemcpyAsync(host, ctx->result.p, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    ctx->timer.finish();
    if (host[12] != 0) { set_error(ctx, "a compressed point does not decode"); return XHE_E_DECODE; }
    memcpy(out32, host, 32);
    if (is_identity) *is_identity = (uint8_t)host[8];
    return 0;
}

// xhe_msm_vartime without any allocation: staging buffers and workspace are ranges of this device's static scratch array.
int32_t xhe_msm_vartime_static(xhe_ctx *ctx, uint64_t n, const uint8_t *scalars32, const uint8_t *points32, uint8_t out32[32], uint8_t *is_identity) {
    if (!ctx || !out32 || (n && (!scalars32 || !points32))) return XHE_E_ARG;
    if (n > XHE_MSM_STATIC_MAX_TERMS) { set_error(ctx, "xhe_msm_vartime_static: more than XHE_MSM_STATIC_MAX_TERMS terms"); return XHE_E_ARG; }
    if (!ctx->static_scratch) { set_error(ctx, "xhe_msm_vartime_static: the static scratch array has no device address"); return XHE_E_HIP; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const u64 nn = n ? n : 1;
    const msm_plan p = msm_make_plan(n);
    const msm_ws_bytes need = msm_workspace_bytes(n, p);
    // carve 256-byte aligned ranges: result, inputs, decoded points, validity marks, then the workspace
    size_t off = 0;
    auto carve = [&](u64 bytes) { unsigned char *q = ctx->static_scratch + off; off += (size_t)((bytes + 255) & ~255ull); return q; };
    u32 *d_result = (u32 *)carve(64), *d_in = (u32 *)carve(nn * 32), *d_sc = (u32 *)carve(nn * 32), *d_niels = (u32 *)carve(nn * 128);
    u8 *d_ok = (u8 *)carve(nn);
    msm_ws w;
    w.hist = (u32 *)carve(need.hist); w.offsets = (u32 *)carve(need.offsets); w.tile_hist = (u32 *)carve(need.tile_hist); w.sorted = (u32 *)carve(need.sorted);
    w.buckets = (u32 *)carve(need.buckets); w.digits = (u32 *)carve(need.digits); w.chunk_sums = (u32 *)carve(need.chunk_sums); w.big_list = (u32 *)carve(need.big_list);
    w.segs = (u32 *)carve(need.segs); w.wins = (u32 *)carve(need.wins); w.tmp = (u32 *)carve(need.tmp); w.coarse = (u32 *)carve(need.coarse);
    if (off > MSM_STATIC_BYTES) { set_error(ctx, "xhe_msm_vartime_static: the workspace of this size does not fit the static scratch array"); return XHE_E_ARG; }
    // the array keeps the previous call's contents: the result words (identity flag, count of undecodable points) start at zero,
    // every other range is written before it is read (the digit kernel resets the list of oversized buckets)
    HIPCHK(ctx, hipMemsetAsync(d_result, 0, 64, st));
    if (n) {
        HIPCHK(ctx, hipMemcpyAsync(d_in, points32, n * 32, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(d_sc, scalars32, n * 32, hipMemcpyHostToDevice, st));
        launch_decode_niels(st, d_in, n, d_niels, d_ok);
        launch_count_bad(st, d_ok, n, d_result + 12);
        launch_reduce_scalars(st, d_sc, n);
    }
    HIPCHK(ctx, msm_launch(st, d_sc, d_niels, nullptr, n, n, p, w, d_result, true, msm_deps(), nullptr, nullptr));
    u32 *host = ctx->verdict_host;                       // pinned at context creation
    HIPCHK(ctx, hipMemcpyAsync(host, d_result, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (host[12] != 0) { set_error(ctx, "a compressed point does not decode"); return XHE_E_DECODE; }
    memcpy(out32, host, 32);
    if (is_identity) *is_identity = (uint8_t)host[8];
    return 0;
}

// ---- level 1 ----------------------------------------------------------------------------------------
void xhe_batch_free(xhe_ctx *ctx, xhe_batch *b) {
    if (!b) return;
    if (ctx) {
        // nothing may still read or write the batch's blocks when they go back to the pool: the copies of its upload, and the
        // kernels of a verification that returned early on an error
        (void)hipSetDevice(ctx->device);
        if (b->ev_uploaded) (void)hipEventSynchronize(b->ev_uploaded);
        for (hipStream_t s : {ctx->stream, ctx->side[0], ctx->side[1], ctx->side[2]}) (void)hipStreamSynchronize(s);
    }
    if (b->tables_lock.owns_lock()) b->tables_lock.unlock();
    b->release();
    if (ctx) { std::lock_guard<std::mutex> lk(ctx->spare_mu); ctx->spare_host.trade_storage(b->hb); }
    delete b;
}

// Copies a list of (destination, source, bytes) on several host threads, in chunks of a few megabytes.
struct copy_task { char *dst; const char *src; size_t bytes; };
static void parallel_copy(const std::vector<copy_task> &tasks_in, unsigned threads, ingest_workers *workers = nullptr) {
    const bool pooled = workers && workers->size() && threads > 1;      // the context's idle ingest threads: worth it from a megabyte on
    const size_t CHUNK = pooled ? (1u << 20) : (4u << 20);
    std::vector<copy_task> tasks;
    size_t total = 0;
    for (auto &t : tasks_in)
        for (size_t o = 0; o < t.bytes; o += CHUNK) { tasks.push_back({t.dst + o, t.src + o, std::min(CHUNK, t.bytes - o)}); total += tasks.back().bytes; }
    if (threads <= 1 || total < (pooled ? (1u << 20) : (8u << 20))) { for (auto &t : tasks) memcpy(t.dst, t.src, t.bytes); return; }
    std::atomic<size_t> next(0);
    auto work = [&]() { for (;;) { size_t i = next.fetch_add(1); if (i >= tasks.size()) break; memcpy(tasks[i].dst, tasks[i].src, tasks[i].bytes); } };
    if (pooled) {
        const std::function<void(unsigned)> part = [&](unsigned) { work(); };
        workers->run(std::min<unsigned>(threads, (unsigned)tasks.size()), part);
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; t++) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
}

// wait_for_copies = false leaves the table copies in flight on the main stream: the caller must synchronise that stream before
// the next upload on this context (which reuses the pinned staging memory) - xhe_verify_batch does, by verifying.
//
// The device's byte buffer (records | account keys | ledger ciphertexts | multisig signer keys) depends on nothing the ingest
// computes, so it is staged and sent by a helper thread while the ingest runs; the descriptor tables follow.
// Range-proof column sums: groups of at least this many transactions go through the matrix cores (xhe_k_rpgemm.hip), smaller
// ones through the element-by-element kernel, whose cost has no fixed part.  XHE_RP_GEMM_MIN_ROWS overrides the threshold
// (1: always the matrix path; a huge value: never); it is read when a batch is uploaded and kept with the batch.
static u64 rp_gemm_min_rows() { const char *e = setting("XHE_RP_GEMM_MIN_ROWS"); return e ? strtoull(e, nullptr, 10) : 512ull; }
static int32_t batch_upload_impl(xhe_ctx *ctx, const uint8_t *records, const uint64_t *offsets, uint64_t n_txs, const xhe_ledger_view *in, xhe_batch **batch_out,
                                 xhe_verdict *structural, bool wait_for_copies, const ledger_index *shared_index = nullptr) {
    if (!ctx || !offsets || !in || !batch_out || (n_txs && !records)) return XHE_E_ARG;
    if (n_txs > 0x7ffffff0ull) return XHE_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    xhe_batch *B = new xhe_batch();
    host_batch &hb = B->hb;
    hipStream_t st = ctx->copy_stream;
    auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!setting("XHE_UPLOAD_TIMES")) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[xhe upload] %s %.3f ms\n", what, std::chrono::duration<double>(now - t_start).count() * 1e3);
        t_start = now;
    };
    auto grow_stage = [&](size_t want) -> hipError_t {
        if (ctx->stage_cap >= want) return hipSuccess;
        if (ctx->stage) (void)hipHostFree(ctx->stage);
        ctx->stage = nullptr; ctx->stage_cap = 0;
        size_t cap = want + want / 4 + 4096;
        hipError_t e = hipHostMalloc(&ctx->stage, cap, hipHostMallocDefault);
        if (e == hipSuccess) ctx->stage_cap = cap;
        return e;
    };
    const unsigned host_threads = ingest_thread_count();
    { std::lock_guard<std::mutex> lk(ctx->spare_mu); hb.trade_storage(ctx->spare_host); }
    // ---- the byte buffer: its own device block, staged and sent while the ingest runs ----
    const u64 rec_bytes = offsets[n_txs], keys_base = (rec_bytes + 63) & ~63ull, cts_base = keys_base + 32 * in->n_accounts, signers_base = cts_base + 64 * in->n_balances;
    const u64 n_signer_keys = in->n_multisig ? in->multisig_offsets[in->n_multisig] : 0;
    const u64 bytes_len = signers_base + 32 * n_signer_keys + 64, bytes_region = (256 + bytes_len + 255) & ~255ull;      // the transcript's word gather may read a few bytes in front of a record
    const bool early_bytes = bytes_region <= STAGE_MAX / 2;
    if (hipEventCreateWithFlags(&B->ev_uploaded, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&B->ev_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&B->ev_carried, hipEventDisableTiming) != hipSuccess) { set_error(ctx, "creating the batch's events failed"); B->release(); delete B; return XHE_E_HIP; }
    std::thread copier;
    hipError_t copier_err = hipSuccess;
    B->arena_bytes.pool = &ctx->pool;
    if (early_bytes) {
        hipError_t he = B->arena_bytes.reserve(bytes_region);
        if (he == hipSuccess && ctx->stage_cap < bytes_region) { he = hipEventSynchronize(ctx->ev_stage_free); if (he == hipSuccess) he = grow_stage(bytes_region); }
        if (he != hipSuccess) { set_error(ctx, std::string("allocating the batch's byte buffer: ") + hipGetErrorString(he)); B->release(); delete B; return XHE_E_HIP; }
        B->bytes.point_at(B->arena_bytes.p, 256, bytes_len);
        copier = std::thread([&, rec_bytes, keys_base, cts_base, signers_base, n_signer_keys, bytes_len]() {
            (void)hipSetDevice(ctx->device);
            (void)hipEventSynchronize(ctx->ev_stage_free);        // the previous upload's copies have left the staging memory
            char *dst = (char *)ctx->stage + 256;
            memset(ctx->stage, 0, 256);
            memset(dst + rec_bytes, 0, keys_base - rec_bytes);
            memset(dst + signers_base + 32 * n_signer_keys, 0, 64);
            std::vector<copy_task> tasks = {{dst, (const char *)records, (size_t)rec_bytes}, {dst + keys_base, (const char *)in->account_keys, (size_t)(32 * in->n_accounts)},
                                            {dst + cts_base, (const char *)in->balance_cts, (size_t)(64 * in->n_balances)},
                                            {dst + signers_base, (const char *)in->multisig_signers, (size_t)(32 * n_signer_keys)}};
            parallel_copy(tasks, host_threads >= 8 ? 4 : 1);
            copier_err = hipMemcpyAsync(B->arena_bytes.p, ctx->stage, 256 + bytes_len, hipMemcpyHostToDevice, st);
        });
    }
    auto join_copier = [&]() { if (copier.joinable()) copier.join(); };
    if (!ctx->workers.size() && host_threads > 1) ctx->workers.start(host_threads - 1);
    bool parsed = ingest_batch(hb, records, offsets, n_txs, *in, &ctx->workers, &ctx->arena, shared_index);
    lap("ingest");
    join_copier();
    lap("byte buffer staged and enqueued");
    if (hb.bad_ledger) {
        set_error(ctx, "the ledger view is inconsistent (an account index out of range or multisig offsets that do not ascend from 0)");
        (void)hipStreamSynchronize(st); B->release(); delete B; return XHE_E_ARG;
    }
    if (copier_err != hipSuccess) { set_error(ctx, std::string("uploading the batch's byte buffer: ") + hipGetErrorString(copier_err)); (void)hipStreamSynchronize(st); B->release(); delete B; return XHE_E_HIP; }
    if (structural) *structural = hb.structural;
    *batch_out = B;
    B->n_balances = in->n_balances; B->n_accounts = in->n_accounts;
    B->in_cts.assign(in->balance_cts, in->balance_cts + 64 * in->n_balances);
    B->in_nonces.assign(in->account_nonces, in->account_nonces + in->n_accounts);
    if (!parsed) { (void)hipStreamSynchronize(st); return 0; }
    int32_t rc = ensure_tables(ctx, hb.nm_cap);
    if (rc) { xhe_batch_free(ctx, B); *batch_out = nullptr; return rc; }
    const std::vector<u32> &emit_slot = hb.emit_slot;
    std::vector<u32> group_txs; B->group_off.assign(17, 0);
    for (u32 lg = 0; lg < 16; lg++) { B->group_off[lg] = (u32)group_txs.size(); group_txs.insert(group_txs.end(), hb.lg_groups[lg].begin(), hb.lg_groups[lg].end()); }
    B->group_off[16] = (u32)group_txs.size();
    // One device block for the uploaded tables and one for the workspace; every buffer of the batch is a 256-byte aligned
    // range of one of them.  The tables are assembled in the context's pinned staging memory (behind the byte buffer's region)
    // and cross in one copy (in slices when they exceed it).
    B->arena_up.pool = &ctx->pool; B->arena_ws.pool = &ctx->pool;
    struct piece { dev_buf *buf; const void *src; size_t bytes, off; };
    std::vector<piece> ups, wss;
    size_t up_total = 0, ws_total = 0;
    auto add = [](std::vector<piece> &list, size_t &total, dev_buf &b, const void *src, size_t bytes) {
        size_t sz = std::max<size_t>(bytes, 64);
        list.push_back({&b, src, bytes, total});
        total += (sz + 255) & ~(size_t)255;
    };
#define UP(buf, vec) add(ups, up_total, B->buf, (vec).data(), (vec).size() * sizeof((vec)[0]))
    if (!early_bytes) {
        // too large for the early path: the byte buffer travels with the tables, straight from the caller's arrays; the gaps between its parts are zero
        static const u8 zeros[128] = {0};
        up_total = 256;
        const size_t base = up_total;
        add(ups, up_total, B->bytes, nullptr, hb.bytes_len);
        auto part = [&](const void *src, size_t off, size_t bytes) { if (bytes) ups.push_back({nullptr, src, bytes, base + off}); };
        part(records, 0, hb.rec_bytes);
        part(zeros, hb.rec_bytes, hb.keys_base - hb.rec_bytes);
        part(in->account_keys, hb.keys_base, 32 * in->n_accounts);
        part(in->balance_cts, hb.cts_base, 64 * in->n_balances);
        part(in->multisig_signers, hb.signers_base, 32 * hb.n_signer_keys);
        part(zeros, hb.signers_base + 32 * hb.n_signer_keys, 64);
    }
    // what the device expands (launch_expand below) into descriptors that live in the workspace block
    UP(ex_txs, hb.ex_txs); UP(ex_xfs, hb.ex_xfs); UP(ex_cms, hb.ex_cms); UP(ex_mss, hb.ex_mss); UP(sh_decs, hb.shared_decs); UP(sh_early, hb.shared_early); UP(sh_key, hb.shared_key);
    UP(ex_slot_pos, hb.slot_pos); UP(ex_slot_base, hb.slot_base);
    UP(sig_keys, hb.sig_keys);
    UP(tmpl, hb.tmpl); UP(serial_txs, hb.serial_txs); UP(tr_list, hb.tr_list); UP(tr_tiles, hb.tr_tiles);
    UP(key_group_off, hb.key_group_off);
    UP(emit_slot, emit_slot); UP(ms_hash_tx, hb.ms_hash_tx); UP(group_txs, group_txs); UP(slot_of_used, hb.slot_of_used);
#undef UP
    u64 n_slots = hb.slot_of_used.size();
    u64 n_ops = hb.n_ops;
    u64 scan_threads = (n_ops + SCAN_K - 1) / SCAN_K, scan_blocks = (scan_threads + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (scan_blocks == 0) scan_blocks = 1;
    u64 n_parts = (u64)hb.n_cm + hb.n_xf + hb.n_txs;
#define RS(buf, bytes_) add(wss, ws_total, B->buf, nullptr, (size_t)(bytes_))
    RS(txs, sizeof(dev_tx) * hb.n_txs); RS(cms, sizeof(dev_cm) * (u64)hb.n_cm); RS(xfers, sizeof(dev_xfer) * (u64)hb.n_xf); RS(decs, sizeof(dev_dec) * (u64)hb.n_decs);
    RS(early_decs, 4ull * hb.n_early); RS(key_decs, 4ull * hb.n_key); RS(bulk_decs, 4ull * hb.n_bulk); RS(ops, sizeof(dev_op) * (u64)hb.n_ops); RS(sigs, sizeof(dev_sig) * (u64)hb.n_sigs);
    RS(segs, 4ull * hb.n_segs); RS(merge_pairs, 8ull * ((u64)hb.n_cm + hb.n_xf)); RS(key_terms, 4ull * hb.n_key_terms); RS(tx_err0, 4ull * hb.n_txs); RS(apply_err0, 4ull * hb.n_txs);
    RS(slot_last_op, 4ull * hb.slot_pos.size());
    RS(pts, 128ull * hb.n_points); RS(ok, hb.n_points); RS(chal, 32ull * hb.n_chal); RS(scalars, 32ull * hb.n_terms); RS(term_point, 4ull * hb.n_terms);
    RS(gh, 72ull * n_parts); RS(gh_sums, 72ull * 256); RS(rp_ctx, 36ull * RPC_SCALARS * hb.n_txs); RS(rp_acc, 72ull * hb.nm_cap);
    RS(tx_err, 4ull * hb.n_txs); RS(rp_err, 4ull * hb.n_txs); RS(ms_hash, 32ull * hb.ms_hash_tx.size());
    RS(thread_excl, 256ull * scan_blocks * SCAN_BLOCK); RS(block_tot, 256ull * scan_blocks); RS(slot_base, 256ull * (n_slots + 1)); RS(slot_final, 256ull * (n_slots + 1));
    RS(emit_incl, 256ull * ((u64)hb.n_cm + 1)); RS(final_cts, 64ull * (n_slots + 1)); RS(verdict, 64); RS(seed, 32); RS(tr_stream, 8ull * hb.tr_plan.stream_words); RS(sig_state, 200ull * hb.n_sigs); RS(sig_r, 128ull * hb.n_sigs); RS(sig_tab, 4ull * SIG_KEY_TABLE_WORDS * hb.sig_keys.size());
    // partial column sums of the range-proof scalar matrix: one row per tile of transactions
    // (groups with many transactions take the matrix-core path instead: digit tables of one chunk, partial diagonal sums)
    {
        u64 worst = 0, dig = 0, bins = 0;
        B->rp_gemm_min_rows = rp_gemm_min_rows();
        for (u32 lg = 6; lg < 16; lg++) {
            u64 rows = hb.lg_groups[lg].size(); if (!rows) continue;
            if (rows >= B->rp_gemm_min_rows) { const rp_gemm_plan p = rp_gemm_plan_for(rows, lg); dig = std::max(dig, p.dig_bytes); bins = std::max(bins, p.bins_bytes); }
            else worst = std::max<u64>(worst, rp_matrix_part_rows(rows, lg) * (72ull << lg));
        }
        RS(rp_mat, worst); RS(rp_dig, dig); RS(rp_bins, bins); RS(rp_rho, 36ull * RP_GEMM_RHO_PARTS);
        RS(static_partials, ((2ull + 2ull * GEN_TABLE_CAP_MAX) * 32 / 256 + 2) * 128); RS(static_sum, 128); RS(noncanonical, 64);
    }
#undef RS
    auto fail = [&](const char *what, hipError_t e) { set_error(ctx, std::string(what) + ": " + hipGetErrorString(e)); xhe_batch_free(ctx, B); *batch_out = nullptr; return XHE_E_HIP; };
    hipError_t he;
    if ((he = B->arena_up.reserve(up_total + 256)) != hipSuccess) return fail("allocating the batch tables", he);
    if ((he = B->arena_ws.reserve(ws_total + 256)) != hipSuccess) return fail("allocating the batch workspace", he);
    for (auto &pc : ups) if (pc.buf) pc.buf->point_at(B->arena_up.p, pc.off, std::max<size_t>(pc.bytes, 64));
    for (auto &pc : wss) pc.buf->point_at(B->arena_ws.p, pc.off, std::max<size_t>(pc.bytes, 64));
    lap("device blocks");
    {
        if (!early_bytes && (he = hipEventSynchronize(ctx->ev_stage_free)) != hipSuccess) return fail("waiting for the staging memory", he);
        // the tables' staging region lies behind the byte buffer's (whose copy may still be in flight)
        size_t stage_off = early_bytes ? bytes_region : 0;
        if (early_bytes && stage_off + std::min<size_t>(up_total, STAGE_MAX / 2) > ctx->stage_cap) {
            // grow the staging memory: only once the byte buffer has left it
            if ((he = hipStreamSynchronize(st)) != hipSuccess) return fail("uploading the batch's byte buffer", he);
            stage_off = 0;
            if ((he = grow_stage(bytes_region + std::min<size_t>(up_total, STAGE_MAX / 2))) != hipSuccess) return fail("allocating pinned staging memory", he);
        } else if (!early_bytes) {
            if ((he = grow_stage(std::min<size_t>(up_total, STAGE_MAX))) != hipSuccess) return fail("allocating pinned staging memory", he);
        }
        const size_t room = ctx->stage_cap - stage_off;
        // slices of the concatenated layout: [lo, hi) in arena offsets
        size_t lo = 0;
        while (lo < up_total) {
            size_t hi = std::min(up_total, lo + room);
            std::vector<copy_task> tasks;
            for (auto &pc : ups) {
                size_t a = std::max(pc.off, lo), b = std::min(pc.off + pc.bytes, hi);
                if (a < b && pc.src) tasks.push_back({(char *)ctx->stage + stage_off + (a - lo), (const char *)pc.src + (a - pc.off), b - a});
            }
            parallel_copy(tasks, host_threads, &ctx->workers);      // (the ingest is over: its threads are idle)
            if ((he = hipMemcpyAsync((char *)B->arena_up.p + lo, (char *)ctx->stage + stage_off, hi - lo, hipMemcpyHostToDevice, st)) != hipSuccess) return fail("uploading the batch tables", he);
            if (hi < up_total && (he = hipStreamSynchronize(st)) != hipSuccess) return fail("uploading the batch tables", he);      // the staging memory is reused
            lo = hi;
        }
        if (up_total > room) wait_for_copies = true;
    }
    lap("tables staged and enqueued");
    {
        // the descriptors, decode lists, ledger operations and signature descriptors are written on the device, behind the copies
        expand_args ea;
        ea.txs = B->ex_txs.as<expand_tx>(); ea.xfs = B->ex_xfs.as<expand_xf>(); ea.cms = B->ex_cms.as<expand_cm>(); ea.mss = B->ex_mss.as<expand_ms>(); ea.n_txs = (u32)hb.n_txs;
        ea.n_static = hb.n_static; ea.emit_base = hb.emit_base;
        ea.out_txs = B->txs.as<dev_tx>(); ea.out_cms = B->cms.as<dev_cm>(); ea.out_xfers = B->xfers.as<dev_xfer>(); ea.decs = B->decs.as<dev_dec>();
        ea.early_decs = B->early_decs.as<u32>(); ea.key_decs = B->key_decs.as<u32>(); ea.bulk_decs = B->bulk_decs.as<u32>(); ea.ops = B->ops.as<dev_op>(); ea.sigs = B->sigs.as<dev_sig>();
        ea.segs = B->segs.as<u32>(); ea.merge_pairs = B->merge_pairs.as<u32>(); ea.key_terms = B->key_terms.as<u32>(); ea.tx_err = B->tx_err0.as<u32>(); ea.apply_err = B->apply_err0.as<u32>();
        ea.shared_decs = B->sh_decs.as<dev_dec>(); ea.shared_early = B->sh_early.as<u32>(); ea.shared_key = B->sh_key.as<u32>(); ea.slot_pos = B->ex_slot_pos.as<u32>(); ea.slot_base = B->ex_slot_base.as<u32>();
        ea.n_shared = (u32)hb.shared_decs.size(); ea.n_shared_early = (u32)hb.shared_early.size(); ea.n_shared_key = (u32)hb.shared_key.size(); ea.n_slots = (u32)hb.slot_pos.size();
        ea.blk_end = hb.blk_end; ea.early_end = hb.early_end; ea.key_end = hb.key_end;
        ea.slot_last_op = B->slot_last_op.as<u32>();
        ea.bytes = B->bytes.as<u8>(); ea.noncanonical_first = B->noncanonical.as<u32>();
        HIPCHK(ctx, hipMemsetAsync(B->noncanonical.p, 0xff, 64, st));
        launch_expand(st, ea);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_stage_free, st));
    HIPCHK(ctx, hipEventRecord(B->ev_uploaded, st));
    if (wait_for_copies) HIPCHK(ctx, hipStreamSynchronize(st));
    lap("synchronize");
    return 0;
}
int32_t xhe_batch_upload(xhe_ctx *ctx, const uint8_t *records, const uint64_t *offsets, uint64_t n_txs, const xhe_ledger_view *in, xhe_batch **batch_out,
                         xhe_verdict *structural) {
    int32_t rc = batch_upload_impl(ctx, records, offsets, n_txs, in, batch_out, structural, true);
    if (rc != 0 || !*batch_out) return rc;
    // the device's range check of the scalars has run by now (the upload waited for its copies and kernels): a batch with a
    // scalar that is not canonical is malformed like one whose lengths are wrong
    xhe_batch *B = *batch_out;
    if (B->hb.structural.accept && B->noncanonical.p) {
        u32 first = 0xffffffffu;
        HIPCHK(ctx, hipMemcpy(&first, B->noncanonical.p, 4, hipMemcpyDeviceToHost));
        if (first != 0xffffffffu) { B->hb.structural = {0, XHE_ERR_MALFORMED, (int64_t)first}; if (structural) *structural = B->hb.structural; }
    }
    return rc;
}

int32_t xhe_batch_verify(xhe_ctx *ctx, xhe_batch *B, xhe_verdict *verdict, const uint8_t rng_seed[32]) {
    if (!ctx || !B || !verdict || !rng_seed) return XHE_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int32_t rc = batch_verify_impl(ctx, B, verdict, rng_seed);
    // an error return leaves in the middle of the launch sequence: nothing of it may still run when the caller goes on
    if (rc != 0) for (hipStream_t s : {ctx->stream, ctx->side[0], ctx->side[1], ctx->side[2]}) (void)hipStreamSynchronize(s);
    return rc;
}

// Transaction::apply_without_verify for every transaction of the batch in order: the ledger arithmetic alone
// (decode of the ledger and transfer ciphertexts, the ledger scan, the resulting encodings) on the main stream.
int32_t xhe_batch_apply(xhe_ctx *ctx, xhe_batch *B, xhe_verdict *verdict) {
    if (!ctx || !B || !verdict) return XHE_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const host_batch &hb = B->hb;
    hipStream_t st = ctx->stream;
    B->verified_ok = false;
    if (!hb.structural.accept) { *verdict = hb.structural; return 0; }
    const u32 n_txs = (u32)hb.n_txs, n_cm = hb.n_cm, n_ops = hb.n_ops, n_slots = (u32)hb.slot_of_used.size();
    stage_timer &tm = ctx->timer;
    tm.begin();
    std::shared_lock<std::shared_mutex> tables_lock(ctx->tables_mu);
    HIPCHK(ctx, hipStreamWaitEvent(st, B->ev_uploaded, 0));
    HIPCHK(ctx, tm.mark("setup", st));
    setup_jobs jobs;
    memset(jobs.seed, 0, 32); jobs.seed_dst = nullptr;
    // the static generators of this batch's point array: B, H, G_vec[nm_cap], H_vec[nm_cap]
    jobs.copy(B->pts.p, ctx->static_pts.p, 64 + 32ull * hb.nm_cap);
    jobs.copy(B->pts.as<u32>() + 64 + 32ull * hb.nm_cap, ctx->static_pts.as<u32>() + 64 + 32ull * ctx->static_cap, 32ull * hb.nm_cap);
    jobs.fill(B->ok.p, 0x01010101u, (hb.n_points + 3) / 4);
    jobs.copy(B->tx_err.p, B->apply_err0.p, std::max<u32>(n_txs, 1));
    jobs.fill(B->rp_err.p, 0, std::max<u32>(n_txs, 1));
    jobs.fill(B->emit_incl.p, 0, 64ull * (n_cm + 1));
    jobs.fill(B->slot_base.p, 0, 64ull * (n_slots + 1));
    jobs.fill(B->slot_final.p, 0, 64ull * (n_slots + 1));
    jobs.fill(B->pts.as<u32>() + 32ull * hb.emit_base, 0, 32ull * hb.n_emit);
    jobs.fill(B->verdict.p, 0xffffffffu, 4);
    jobs.fill(B->verdict.as<u32>() + 4, 0, 12);
    launch_setup(st, jobs);
    if (n_txs) {
        u32 *pts = B->pts.as<u32>(); u8 *ok = B->ok.as<u8>();
        HIPCHK(ctx, tm.mark("decode_first", st));
        launch_decode_list(st, B->bytes.as<u8>(), B->decs.as<dev_dec>(), B->early_decs.as<u32>(), hb.n_early, hb.n_static, pts, ok);
        HIPCHK(ctx, tm.mark("ledger_scan", st));
        launch_ledger_scan(st, B->ops.as<dev_op>(), n_ops, pts, ctx->gtab.as<u32>(), B->thread_excl.as<u32>(), B->block_tot.as<u32>(), B->slot_last_op.as<u32>(),
                           B->slot_base.as<u32>(), B->slot_final.as<u32>(), B->emit_incl.as<u32>(), B->emit_slot.as<u32>(), n_cm, hb.emit_base, pts, ok);
        if (n_ops) launch_ledger_final(st, B->slot_final.as<u32>(), B->slot_base.as<u32>(), n_slots, B->final_cts.as<u32>());
        HIPCHK(ctx, tm.mark("verdict", st));
        // only the ciphertexts were decoded: every other point still carries its initial "valid" mark
        launch_point_errors(st, B->txs.as<dev_tx>(), B->cms.as<dev_cm>(), B->xfers.as<dev_xfer>(), n_txs, ok, B->tx_err.as<u32>(), B->rp_err.as<u32>());
        launch_verdict(st, B->tx_err.as<u32>(), B->rp_err.as<u32>(), n_txs, B->verdict.as<unsigned long long>(), B->verdict.as<u32>() + 2, B->noncanonical.as<u32>());
    }
    HIPCHK(ctx, tm.mark("end", st));
    HIPCHK(ctx, hipGetLastError());
    u32 host[16];
    HIPCHK(ctx, hipMemcpyAsync(host, B->verdict.p, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    tm.finish();
    u64 first_err = (u64)host[0] | ((u64)host[1] << 32);
    if (n_txs && host[3] != 0xffffffffu) { *verdict = {0, XHE_ERR_MALFORMED, (int64_t)host[3]}; return 0; }      // a scalar that is not canonical: the record would not have deserialised
    if (first_err != ~0ull) { *verdict = {0, (int32_t)(first_err & 0xff), (int64_t)(first_err >> 32)}; return 0; }
    *verdict = {1, XHE_OK, -1};
    B->verified_ok = true;
    return 0;
}

int32_t xhe_batch_read_ledger(xhe_ctx *ctx, xhe_batch *B, const xhe_ledger_out *out) {
    if (!ctx || !B || !out) return XHE_E_ARG;
    if (!B->verified_ok) { set_error(ctx, "the batch has not been verified (or applied) and accepted"); return XHE_E_ARG; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const host_batch &hb = B->hb;
    u64 n_slots = hb.slot_of_used.size();
    std::vector<u8> fin(64 * n_slots + 64);
    if (n_slots) HIPCHK(ctx, hipMemcpyAsync(fin.data(), B->final_cts.p, 64 * n_slots, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (out->balance_cts) {
        if (out->balance_cts != B->in_cts.data()) memmove(out->balance_cts, B->in_cts.data(), 64 * B->n_balances);
        for (u64 u = 0; u < n_slots; u++) memcpy(out->balance_cts + 64ull * hb.slot_of_used[u], fin.data() + 64 * u, 64);
    }
    if (out->account_nonces) memcpy(out->account_nonces, B->in_nonces.data(), 8 * B->n_accounts);
    if (out->multisig_set_by) memcpy(out->multisig_set_by, hb.multisig_set_by.data(), 8 * B->n_accounts);
    return 0;
}

// Replaces the ledger ciphertexts an uploaded batch starts from (its byte buffer holds them behind the records and the account keys).
int32_t xhe_batch_set_ledger(xhe_ctx *ctx, xhe_batch *B, const uint8_t *balance_cts) {
    if (!ctx || !B || (B->n_balances && !balance_cts)) return XHE_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    B->verified_ok = false;
    if (!B->n_balances) return 0;
    B->in_cts.assign(balance_cts, balance_cts + 64 * B->n_balances);
    if (!B->hb.structural.accept || !B->bytes.p) return 0;              // nothing of the batch is on the device
    HIPCHK(ctx, hipMemcpyAsync(B->bytes.as<u8>() + B->hb.cts_base, B->in_cts.data(), 64 * B->n_balances, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// Diagnostics for parity tests: internal values of an uploaded batch, copied out after xhe_batch_verify.
int64_t xhe_batch_debug_read(xhe_ctx *ctx, xhe_batch *B, int32_t what, uint8_t *out, uint64_t cap_bytes) {
    if (!ctx || !B || !out) return XHE_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const host_batch &hb = B->hb;
    if (!hb.structural.accept || !B->pts.p) { set_error(ctx, "xhe_batch_debug_read: the batch was not uploaded to the device"); return XHE_E_ARG; }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (what == XHE_DEBUG_CHALLENGES) {
        const u64 n = std::min<u64>(hb.n_chal, cap_bytes / 32);
        if (n) HIPCHK(ctx, hipMemcpy(out, B->chal.p, 32 * n, hipMemcpyDeviceToHost));
        return (int64_t)n;
    }
    if (what == XHE_DEBUG_GENERATORS) {
        // B, H, then G_vec[0..k) and H_vec[0..k) of the batch's point array (affine Niels records), re-encoded on the host
        const u64 k = std::min<u64>(hb.nm_cap, cap_bytes / 32 >= 2 ? (cap_bytes / 32 - 2) / 2 : 0);
        if (cap_bytes < 64) return XHE_E_ARG;
        std::vector<u32> rec(32 * (2 + 2 * k));
        HIPCHK(ctx, hipMemcpy(rec.data(), B->pts.p, 128 * (2 + k), hipMemcpyDeviceToHost));
        if (k) HIPCHK(ctx, hipMemcpy(rec.data() + 32 * (2 + k), B->pts.as<u8>() + 128ull * (2 + hb.nm_cap), 128 * k, hipMemcpyDeviceToHost));
        fe inv2 = fe_zero();                                // (p + 1) / 2
        inv2.v[0] = (1u << 26) - 9;
        for (int i = 1; i < 9; i++) inv2.v[i] = (i & 1) ? FE_MASK25 : FE_MASK26;
        inv2.v[9] = (1u << 24) - 1;
        for (u64 i = 0; i < 2 + 2 * k; i++) {
            ge_niels q = ge_niels_load(rec.data() + 32 * i);
            ge p; p.X = fe_mul(fe_sub(q.ypx, q.ymx), inv2); p.Y = fe_mul(fe_add(q.ypx, q.ymx), inv2); p.Z = fe_one(); p.T = fe_mul(p.X, p.Y);
            ristretto_encode(out + 32 * i, p);
        }
        return (int64_t)(2 + 2 * k);
    }
    if (what == XHE_DEBUG_STATIC_SCALARS) {
        // scalars of B, H, G_vec[nm_cap], H_vec[nm_cap] as the last xhe_batch_verify of this batch left them (canonical, 32 bytes each)
        const u64 n = std::min<u64>(2 + 2ull * hb.nm_cap, cap_bytes / 32);
        if (n) HIPCHK(ctx, hipMemcpy(out, B->scalars.as<u8>() + 32ull * hb.static_term_off, 32 * n, hipMemcpyDeviceToHost));
        return (int64_t)n;
    }
    return XHE_E_ARG;
}

// ---- one large batch from host buffers, cut into consecutive parts ---------------------------------------------------------
// The host's share of a call (parse, ledger look-ups, the ordered walk: 16 ms for 100 000 transactions of 8 transfers) runs
// before the device has anything to do.  A large batch is therefore cut into consecutive parts that go through the
// upload-beside-verification pipeline of xhe_verify_batches, each part verified against the ledger the part before it left:
//  * the ciphertexts a part starts from are copied into its byte buffer on the device, behind its upload (the host tables of a
//    part depend on the ledger's keys and slots only, never on ciphertext values, so its upload need not wait);
//  * the verdict is the reference's (verify.rs:492-514): the per-transaction error with the lowest index, else GenericProof,
//    else RangeProof, a MIN over one key per part; a malformed record anywhere comes before all of them, as in the uncut path
//    (every part is still parsed after an error);
//  * a transaction that changes an account's multisig configuration decides which signer keys later transactions are checked
//    against: the parts behind it are uploaded with a ledger view whose multisig configurations are the ones the parts so far
//    leave (the uploads run in order on one thread).
// Parts: XHE_SPLIT_PARTS (0 or 1: never; k: k parts whenever the batch has k transactions), else by the size of the records.
static u32 split_parts_for(u64 n_txs, const u64 *offsets, const xhe_ledger_view *in) {
    if (n_txs < 2) return 1;
    for (u64 i = 0; i < n_txs; i++) if (offsets[i + 1] < offsets[i]) return 1;      // the uncut path names the malformed record
    if (offsets[0] != 0) return 1;
    if (const char *e = setting("XHE_SPLIT_PARTS")) {
        const u64 v = strtoull(e, nullptr, 10);
        return v <= 1 ? 1u : (u32)std::min<u64>(std::min<u64>(v, n_txs), 64);
    }
    const u64 rec_bytes = offsets[n_txs];
    const u64 n_signers = in->n_multisig ? in->multisig_offsets[in->n_multisig] : 0;
    const u64 ledger_bytes = 32 * in->n_accounts + 64 * in->n_balances + 32 * n_signers;
    // every part carries the whole ledger view: not worth it for a view that is large beside the records
    if (ledger_bytes > rec_bytes / 16) return 1;
    const u64 parts = std::max(rec_bytes / SPLIT_PART_BYTES, n_txs / SPLIT_PART_TXS);
    return parts < 2 ? 1u : (u32)std::min<u64>(std::min<u64>(parts, SPLIT_PARTS_MAX), n_txs);
}

static u64 split_verdict_key(const xhe_verdict &v, u64 base) {
    if (v.accept) return ~0ull >> 1;
    const u64 stage = v.err_code == XHE_ERR_GENERIC_PROOF ? 1 : v.err_code == XHE_ERR_RANGE_PROOF ? 2 : 0;
    const u64 index = v.first_bad_tx >= 0 ? base + (u64)v.first_bad_tx : (1ull << 52) - 1;
    return (stage << 60) | (index << 8) | ((u64)v.err_code & 0xff);
}

// a batch whose verification has finished (its events have fired) goes back to the pool without draining the streams: the next
// part's verification is running on them
static void batch_free_finished(xhe_ctx *ctx, xhe_batch *b) {
    if (b->tables_lock.owns_lock()) b->tables_lock.unlock();
    b->release();
    { std::lock_guard<std::mutex> lk(ctx->spare_mu); ctx->spare_host.trade_storage(b->hb); }
    delete b;
}

// returns 1 when the uncut path has to take the call (an inconsistent ledger view: it reports that), else what xhe_verify_batch returns
static int32_t verify_batch_split(xhe_ctx *ctx, const uint8_t *records, const uint64_t *offsets, uint64_t n_txs, const xhe_ledger_view *in, const xhe_ledger_out *out,
                                  xhe_verdict *verdict, const uint8_t rng_seed[32], u32 n_parts) {
    struct part { u64 lo = 0, hi = 0; std::vector<u64> offs; xhe_batch *B = nullptr; int32_t rc = 0; xhe_verdict structural{1, XHE_OK, -1}; };
    std::vector<part> parts(n_parts);
    {
        // Shares of the record bytes, at least one transaction each.  The first part is smaller than the others (half their size;
        // XHE_SPLIT_FIRST = its share in per cent of theirs): the device idles while the host parses it.
        const u64 rec_bytes = offsets[n_txs];
        u64 first_pct = 50, second_pct = 100;
        if (const char *e = setting("XHE_SPLIT_FIRST")) { const u64 v = strtoull(e, nullptr, 10); if (v >= 1 && v <= 100) first_pct = v; }
        if (const char *e = setting("XHE_SPLIT_SECOND")) { const u64 v = strtoull(e, nullptr, 10); if (v >= 1 && v <= 100) second_pct = v; }
        std::vector<u64> upto(n_parts + 1, 0);                      // weight of the parts in front of part k
        u64 last_pct = 100;
        if (const char *e = setting("XHE_SPLIT_LAST")) { const u64 v = strtoull(e, nullptr, 10); if (v >= 1 && v <= 100) last_pct = v; }
        for (u32 k = 0; k < n_parts; k++) upto[k + 1] = upto[k] + (k == 0 ? first_pct : (k == 1 ? second_pct : (k + 1 == n_parts && n_parts > 2 ? last_pct : 100)));
        u64 lo = 0;
        for (u32 k = 0; k < n_parts; k++) {
            const u64 cut = (u64)((unsigned __int128)rec_bytes * upto[k + 1] / upto[n_parts]);
            u64 hi = k + 1 == n_parts ? n_txs : (u64)(std::lower_bound(offsets, offsets + n_txs, cut) - offsets);
            hi = std::max(hi, lo + 1); hi = std::min(hi, n_txs - (n_parts - 1 - k));
            parts[k].lo = lo; parts[k].hi = hi;
            parts[k].offs.resize(hi - lo + 1);
            for (u64 i = lo; i <= hi; i++) parts[k].offs[i - lo] = offsets[i] - offsets[lo];
            lo = hi;
        }
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // The ledger's ciphertexts as the parts so far left them, on the device: every part writes the slots it used into it
    // (side stream 0, behind its ledger scan), the next part's byte buffer takes a copy in front of its own ledger decode on the
    // same stream.  A part's verification is enqueued as soon as its upload is in, behind the part in front, which is finished
    // (its verdict read) afterwards: the device goes from one part to the next without waiting for the host.
    // XHE_SPLIT_OVERLAP=0: a part is enqueued only when the part in front has finished (the ledger passes through the host).
    const char *overlap_env = setting("XHE_SPLIT_OVERLAP");
    const bool overlap = !(overlap_env && overlap_env[0] == '0');
    // consecutive parts alternate between two pairs of streams and two bucket workspaces (XHE_SPLIT_LANES=0: all on one pair)
    const char *lanes_env = setting("XHE_SPLIT_LANES");
    const bool lanes = overlap && !(lanes_env && lanes_env[0] == '0') && !setting("XHE_ONE_STREAM");
    u32 n_enqueued = 0;
    // (XHE_ONE_STREAM, diagnostics: the verification's ledger work runs on the main stream, so the hand-over does too)
    hipStream_t s0 = setting("XHE_ONE_STREAM") ? ctx->stream : ctx->side[0];
    dev_buf ledger; ledger.pool = &ctx->pool;
    struct release_ledger { xhe_ctx *c; dev_buf &b; ~release_ledger() { for (hipStream_t s : {c->stream, c->side[0]}) (void)hipStreamSynchronize(s); b.release(); } } ledger_guard{ctx, ledger};
    if (overlap && in->n_balances) {
        HIPCHK(ctx, ledger.reserve(64 * in->n_balances));
        HIPCHK(ctx, hipMemcpyAsync(ledger.p, in->balance_cts, 64 * in->n_balances, hipMemcpyHostToDevice, s0));
    }
    const int timing_level = ctx->timer.level;
    if (overlap) ctx->timer.level = 0;              // stage times are per verification; two are in flight here
    struct restore_level { stage_timer &t; int level; ~restore_level() { t.level = level; } } level_guard{ctx->timer, timing_level};

    // the ledger's key maps are the same for every part: built once (an inconsistent view is the uncut path's error to report)
    if (!ledger_index::view_consistent(*in)) return 1;
    ledger_index index;
    index.build(*in);
    std::mutex mu; std::condition_variable cv;
    u32 uploaded = 0, verified = 0; bool stop = false;
    std::thread uploader([&]() {
        // The multisig configurations as the parts so far leave them.  A MultiSig transaction decides which signer keys the later
        // transactions of its account are checked against (verify.rs:259-292, 401-428); for the parts behind it those keys are
        // handed on as part of the ledger view they are uploaded with (the accounts and balances of the view stay the caller's).
        struct ms_entry { bool present = false; u8 threshold = 0; std::vector<u8> signers; };
        std::vector<ms_entry> ms;
        std::vector<u32> ms_account, ms_offsets; std::vector<u8> ms_threshold, ms_signers;
        xhe_ledger_view carried = *in;
        bool ms_changed = false;
        for (u32 k = 0; k < n_parts; k++) {
            // at most two parts ahead of the last finished one (one ahead of the one the device is working on)
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return stop || k <= verified + (overlap ? 2 : 1); }); if (stop) return; }
            part &p = parts[k];
            p.rc = batch_upload_impl(ctx, records + offsets[p.lo], p.offs.data(), p.hi - p.lo, ms_changed ? &carried : in, &p.B, &p.structural, false, &index);
            if (p.rc == 0 && p.B && p.B->hb.multisig_updated && k + 1 < n_parts) {
                if (ms.empty()) {
                    ms.resize(in->n_accounts);
                    for (u64 i = 0; i < in->n_multisig; i++) {         // (the last entry of an account counts, as in the upload's walk)
                        ms_entry &e = ms[in->multisig_account[i]];
                        e.present = true; e.threshold = in->multisig_threshold[i];
                        e.signers.assign(in->multisig_signers + 32ull * in->multisig_offsets[i], in->multisig_signers + 32ull * in->multisig_offsets[i + 1]);
                    }
                }
                const std::vector<int64_t> &set_by = p.B->hb.multisig_set_by;
                for (u64 a = 0; a < in->n_accounts; a++) {
                    if (set_by[a] < 0) continue;
                    // the record of a MultiSig transaction: 16-byte header (n_items = signers), pre-image = version, source, fee, nonce, threshold, signer keys
                    const u8 *rec = records + offsets[p.lo + (u64)set_by[a]];
                    const u32 n_signers = (u32)rec[2] | ((u32)rec[3] << 8);
                    ms_entry &e = ms[a];
                    e.present = n_signers != 0; e.threshold = rec[16 + 49];
                    e.signers.assign(rec + 16 + 50, rec + 16 + 50 + 32ull * n_signers);
                }
                ms_account.clear(); ms_offsets.assign(1, 0u); ms_threshold.clear(); ms_signers.clear();
                for (u64 a = 0; a < in->n_accounts; a++) {
                    if (!ms[a].present) continue;
                    ms_account.push_back((u32)a); ms_threshold.push_back(ms[a].threshold);
                    ms_signers.insert(ms_signers.end(), ms[a].signers.begin(), ms[a].signers.end());
                    ms_offsets.push_back((u32)(ms_signers.size() / 32));
                }
                carried.n_multisig = ms_account.size(); carried.multisig_account = ms_account.data(); carried.multisig_threshold = ms_threshold.data();
                carried.multisig_offsets = ms_offsets.data(); carried.multisig_signers = ms_signers.data();
                ms_changed = true;
            }
            { std::lock_guard<std::mutex> lk(mu); uploaded = k + 1; }
            cv.notify_all();
        }
    });
    std::vector<u8> cur;                               // (without overlap) the ledger's ciphertexts as the parts so far left them
    if (!overlap) cur.assign(in->balance_cts, in->balance_cts + 64 * in->n_balances);
    std::vector<int64_t> ms_set_by;
    u64 key = ~0ull >> 1, malformed_key = ~0ull;      // a malformed record comes before every other error, whichever part holds it
    bool skip_verify = false;
    int32_t rc = 0;
    const bool show_times = setting("XHE_UPLOAD_TIMES") != nullptr;      // phase times on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto at = [&](u32 k, const char *what) {
        if (show_times) fprintf(stderr, "[xhe split] part %u %-20s at %.3f ms\n", k, what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() * 1e3);
    };
    auto mark_verified = [&](u32 k) { { std::lock_guard<std::mutex> lk(mu); verified = k + 1; } cv.notify_all(); };
    // the verdict of an enqueued part; what it left in the ledger (host copy, without overlap); its batch back to the pool
    auto finish_part = [&](u32 k) -> int32_t {
        part &p = parts[k];
        xhe_batch *B = p.B;
        xhe_verdict v{1, XHE_OK, -1};
        int32_t r = batch_verify_finish(ctx, B, &v);
        at(k, "verified");
        if (r == 0 && !v.accept && v.err_code == XHE_ERR_MALFORMED) {
            // a scalar that is not canonical, found by the device's check of this part: malformed records come before every other
            // error, the one with the lowest index first; nothing behind this part can change the verdict
            malformed_key = std::min(malformed_key, split_verdict_key(v, p.lo));
            skip_verify = true;
        } else if (r == 0) {
            const u64 vk = split_verdict_key(v, p.lo);
            key = std::min(key, vk);
            if (!v.accept && (vk >> 60) == 0) skip_verify = true;      // a per-transaction error: nothing behind it can come first but a malformed record
            const host_batch &hb = B->hb;
            if (!overlap && !skip_verify) {
                // what this part left in the slots it used (also after a failed aggregate check: the reference has applied
                // every transaction by then, verify.rs:492-498)
                const u64 n_slots = hb.slot_of_used.size();
                std::vector<u8> fin(64 * n_slots + 64);
                hipError_t e = hipSuccess;
                if (n_slots) { e = hipMemcpyAsync(fin.data(), B->final_cts.p, 64 * n_slots, hipMemcpyDeviceToHost, ctx->stream); if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream); }
                if (e != hipSuccess) { set_error(ctx, std::string("reading a part's ledger back: ") + hipGetErrorString(e)); r = XHE_E_HIP; }
                else for (u64 u = 0; u < n_slots; u++) memcpy(cur.data() + 64ull * hb.slot_of_used[u], fin.data() + 64 * u, 64);
            }
            if (hb.multisig_updated) {            // (the parts finish in order: a later part's set-up of an account replaces an earlier one's)
                if (ms_set_by.empty()) ms_set_by.assign(in->n_accounts, -1);
                for (u64 a = 0; a < in->n_accounts; a++) if (hb.multisig_set_by[a] >= 0) ms_set_by[a] = hb.multisig_set_by[a] + (int64_t)p.lo;
            }
        }
        if (r == 0 && overlap) { hipError_t e = hipEventSynchronize(B->ev_carried); if (e != hipSuccess) { set_error(ctx, std::string("waiting for a part's ledger: ") + hipGetErrorString(e)); r = XHE_E_HIP; } }
        if (r == 0) { batch_free_finished(ctx, B); p.B = nullptr; }     // (after an error the batch is freed below, behind a drain of the streams)
        at(k, "freed");
        mark_verified(k);
        return r;
    };
    int64_t in_flight = -1;                            // the part whose verification is enqueued and not finished
    for (u32 k = 0; k < n_parts && rc == 0; k++) {
        // Whichever comes first: the part's upload (it is then enqueued behind the part in flight), or the end of the part in
        // flight (it is then finished while the upload goes on -- an upload that has to rebuild the context's tables for larger
        // proofs even waits for that).
        while (in_flight >= 0) {
            { std::unique_lock<std::mutex> lk(mu); if (cv.wait_for(lk, std::chrono::microseconds(200), [&]() { return uploaded > k; })) break; }
            if (hipEventQuery(parts[(size_t)in_flight].B->ev_done) != hipErrorNotReady) { rc = finish_part((u32)in_flight); in_flight = -1; }
        }
        if (rc != 0) break;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return uploaded > k; }); }
        at(k, "uploaded");
        part &p = parts[k];
        rc = p.rc;
        if (rc == 0 && !p.structural.accept) { malformed_key = std::min(malformed_key, split_verdict_key(p.structural, p.lo)); break; }
        if (rc == 0 && !overlap && in_flight >= 0) { rc = finish_part((u32)in_flight); in_flight = -1; }
        if (rc == 0 && !skip_verify) {
            xhe_batch *B = p.B;
            const host_batch &hb = B->hb;
            // the stream this part's ledger inputs are decoded on: side stream 1 with lanes (batch_verify_enqueue), else side stream 0
            hipStream_t s_in = lanes ? ctx->side[1] : s0;
            ledger_handover ho{ledger.p, 64 * in->n_balances, (overlap && in_flight >= 0) ? parts[(size_t)in_flight].B->ev_carried : nullptr};
            const bool hand_inside = lanes && k > 0 && in->n_balances;      // with lanes the copy is part of the part's own launch sequence
            if (k > 0 && in->n_balances && !hand_inside) {
                // the ciphertexts this part starts from, behind its upload, behind the hand-over of the part in front (a part that
                // has finished has handed over) and in front of its ledger decode
                hipError_t e = hipStreamWaitEvent(s_in, B->ev_uploaded, 0);
                if (e == hipSuccess && overlap && in_flight >= 0) e = hipStreamWaitEvent(s_in, parts[(size_t)in_flight].B->ev_carried, 0);
                if (e == hipSuccess) e = overlap ? hipMemcpyAsync(B->bytes.as<u8>() + hb.cts_base, ledger.p, 64 * in->n_balances, hipMemcpyDeviceToDevice, s_in)
                                                 : hipMemcpyAsync(B->bytes.as<u8>() + hb.cts_base, cur.data(), 64 * in->n_balances, hipMemcpyHostToDevice, s_in);
                if (e == hipSuccess && !overlap) e = hipStreamSynchronize(s_in);
                if (e != hipSuccess) { set_error(ctx, std::string("handing the ledger to the next part of the batch: ") + hipGetErrorString(e)); rc = XHE_E_HIP; }
            }
            u8 seed[32]; memcpy(seed, rng_seed, 32); seed[31] ^= (u8)k; seed[30] ^= (u8)(k >> 8);      // independent batch factors per part
            const int lane = lanes ? (int)(n_enqueued & 1) : -1;
            u32 *carry = overlap ? ledger.as<u32>() : nullptr;
            if (rc == 0) rc = batch_verify_enqueue(ctx, B, seed, overlap, in_flight < 0, lane, carry, hand_inside ? &ho : nullptr);
            if (rc == ENQUEUE_TABLES_BUSY) {
                rc = finish_part((u32)in_flight); in_flight = -1;
                ho.after = nullptr;
                if (rc == 0) rc = batch_verify_enqueue(ctx, B, seed, overlap, true, lane, carry, hand_inside ? &ho : nullptr);
            }
            if (rc == 0) n_enqueued++;
            at(k, "enqueued");
            if (rc == 0 && in_flight >= 0) { rc = finish_part((u32)in_flight); in_flight = -1; }
            if (rc == 0) in_flight = k;
        } else if (rc == 0) {
            // nothing to verify behind a per-transaction error: the part was parsed for malformed records only (the scalars' range
            // check ran on the device with its upload)
            if (in_flight >= 0) { rc = finish_part((u32)in_flight); in_flight = -1; }
            if (rc == 0 && p.B && p.B->noncanonical.p && malformed_key == ~0ull) {
                u32 first = 0xffffffffu;
                hipError_t e = hipEventSynchronize(p.B->ev_uploaded);
                if (e == hipSuccess) e = hipMemcpy(&first, p.B->noncanonical.p, 4, hipMemcpyDeviceToHost);
                if (e != hipSuccess) { set_error(ctx, std::string("reading a part's scalar check back: ") + hipGetErrorString(e)); rc = XHE_E_HIP; }
                else if (first != 0xffffffffu) malformed_key = split_verdict_key(xhe_verdict{0, XHE_ERR_MALFORMED, (int64_t)first}, p.lo);
            }
            if (p.B) { xhe_batch_free(ctx, p.B); p.B = nullptr; }
            mark_verified(k);
        }
    }
    if (rc == 0 && in_flight >= 0) { rc = finish_part((u32)in_flight); in_flight = -1; }
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    uploader.join();
    // whatever is left (an early exit, an error): drain the streams once, then free
    bool any_left = false;
    for (auto &p : parts) if (p.B) any_left = true;
    if (any_left) for (hipStream_t s : {ctx->stream, ctx->side[0], ctx->side[1], ctx->side[2]}) (void)hipStreamSynchronize(s);
    for (auto &p : parts) if (p.B) { p.B->in_flight = false; xhe_batch_free(ctx, p.B); p.B = nullptr; }
    if (rc != 0) return rc;
    if (malformed_key != ~0ull) key = malformed_key;
    if (key == ~0ull >> 1) *verdict = {1, XHE_OK, -1};
    else {
        const u64 index = (key >> 8) & ((1ull << 52) - 1);
        *verdict = {0, (int32_t)(key & 0xff), index == (1ull << 52) - 1 ? (int64_t)-1 : (int64_t)index};
    }
    if (!verdict->accept || !out) return 0;
    if (out->balance_cts && in->n_balances) {
        if (overlap) {
            std::vector<u8> fin(64 * in->n_balances);
            HIPCHK(ctx, hipMemcpyAsync(fin.data(), ledger.p, 64 * in->n_balances, hipMemcpyDeviceToHost, s0));
            HIPCHK(ctx, hipStreamSynchronize(s0));
            memcpy(out->balance_cts, fin.data(), 64 * in->n_balances);
        } else memcpy(out->balance_cts, cur.data(), 64 * in->n_balances);
    }
    if (out->account_nonces) memmove(out->account_nonces, in->account_nonces, 8 * in->n_accounts);
    if (out->multisig_set_by) {
        if (ms_set_by.size() == in->n_accounts) memcpy(out->multisig_set_by, ms_set_by.data(), 8 * in->n_accounts);
        else for (u64 a = 0; a < in->n_accounts; a++) out->multisig_set_by[a] = -1;
    }
    return 0;
}

int32_t xhe_verify_batch(xhe_ctx *ctx, const uint8_t *records, const uint64_t *offsets, uint64_t n_txs, const xhe_ledger_view *in, const xhe_ledger_out *out,
                         xhe_verdict *verdict, const uint8_t rng_seed[32]) {
    if (!ctx || !verdict || !rng_seed) return XHE_E_ARG;
    if (offsets && in && records) {
        const u32 n_parts = split_parts_for(n_txs, offsets, in);
        if (n_parts > 1) {
            const int32_t rc = verify_batch_split(ctx, records, offsets, n_txs, in, out, verdict, rng_seed, n_parts);
            if (rc != 1) return rc;
        }
    }
    xhe_batch *B = nullptr;
    int32_t rc = batch_upload_impl(ctx, records, offsets, n_txs, in, &B, verdict, false);
    if (rc) return rc;
    if (!verdict->accept) { xhe_batch_free(ctx, B); return 0; }
    rc = xhe_batch_verify(ctx, B, verdict, rng_seed);
    if (rc == 0 && verdict->accept && out) rc = xhe_batch_read_ledger(ctx, B, out);
    xhe_batch_free(ctx, B);
    return rc;
}

// Transaction::verify_batch over several independent batches: a helper thread parses and uploads the batches ahead (host work
// and copies on the copy stream) while this thread verifies.  Batch k + 1 is enqueued on the device as soon as its upload is in,
// behind batch k, whose verdict is read afterwards (batch_verify_enqueue / batch_verify_finish): the latency-bound head of one
// verification runs beside the tail of the one before.  The resulting ledger ciphertexts come back through pinned memory of
// the context, copied on side stream 0 right behind the ledger scan, so that reading them does not wait for the batch behind.
// XHE_BATCHES_OVERLAP=0: a batch is enqueued only when the batch before it has finished.
int32_t xhe_verify_batches(xhe_ctx *ctx, uint64_t n_batches, const xhe_batch_args *batches, xhe_verdict *verdicts) {
    if (!ctx || (n_batches && (!batches || !verdicts))) return XHE_E_ARG;
    for (uint64_t k = 0; k < n_batches; k++) if (!batches[k].offsets || !batches[k].in || !batches[k].rng_seed || (batches[k].n_txs && !batches[k].records)) return XHE_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const char *overlap_env = setting("XHE_BATCHES_OVERLAP");
    const bool overlap = !(overlap_env && overlap_env[0] == '0');
    const int timing_level = ctx->timer.level;
    if (overlap) ctx->timer.level = 0;              // stage times are per verification; two are in flight here
    struct restore_level { stage_timer &t; int level; ~restore_level() { t.level = level; } } level_guard{ctx->timer, timing_level};
    std::vector<xhe_batch *> up(n_batches, nullptr);
    std::vector<int32_t> up_rc(n_batches, 0);
    // A verdict counts only once its batch has been verified to the end (or found malformed by the upload): decided[k].  The
    // uploads' structural verdicts (accept = 1 for a well-formed batch) land in a side array, never in the caller's, so that an
    // error that ends the call early cannot leave an unverified batch reading "accept".
    std::vector<xhe_verdict> structural(n_batches, xhe_verdict{0, XHE_ERR_STATE, -1});
    std::vector<char> decided(n_batches, 0);
    for (uint64_t k = 0; k < n_batches; k++) verdicts[k] = {0, XHE_ERR_STATE, -1};
    int64_t failed_batch = -1;
    std::mutex mu; std::condition_variable cv;
    uint64_t uploaded = 0, verified = 0; bool stop = false;
    std::thread uploader([&]() {
        for (uint64_t k = 0; k < n_batches; k++) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return stop || k <= verified + (overlap ? 2 : 1); }); if (stop) return; }
            const xhe_batch_args &a = batches[k];
            up_rc[k] = batch_upload_impl(ctx, a.records, a.offsets, a.n_txs, a.in, &up[k], &structural[k], false);
            { std::lock_guard<std::mutex> lk(mu); uploaded = k + 1; }
            cv.notify_all();
        }
    });
    auto mark_verified = [&](uint64_t k) { { std::lock_guard<std::mutex> lk(mu); verified = k + 1; } cv.notify_all(); };
    const bool show_times = setting("XHE_UPLOAD_TIMES") != nullptr;      // phase times on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto at = [&](uint64_t k, const char *what) {
        if (show_times) fprintf(stderr, "[xhe batches] batch %llu %-12s at %.3f ms\n", (unsigned long long)k, what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() * 1e3);
    };
    hipStream_t s0 = setting("XHE_ONE_STREAM") ? ctx->stream : ctx->side[0];      // (diagnostics: the ledger work of a verification is on the main stream then)
    int32_t rc = 0;
    // the verdict of an enqueued batch, its ledger, its memory back to the pool
    auto finish_batch = [&](uint64_t k) -> int32_t {
        xhe_batch *B = up[k];
        xhe_verdict v{0, XHE_ERR_STATE, -1};
        int32_t r = batch_verify_finish(ctx, B, &v);
        at(k, "verified");
        if (r == 0) { hipError_t e = hipEventSynchronize(B->ev_carried); if (e != hipSuccess) { set_error(ctx, std::string("reading a batch's ledger back: ") + hipGetErrorString(e)); r = XHE_E_HIP; } }
        const xhe_ledger_out *out = batches[k].out;
        if (r == 0 && v.accept && out) {
            const host_batch &hb = B->hb;
            const u8 *fin = ctx->readback[k & 1];
            if (out->balance_cts) {
                if (out->balance_cts != B->in_cts.data()) memmove(out->balance_cts, B->in_cts.data(), 64 * B->n_balances);
                for (u64 u = 0; u < hb.slot_of_used.size(); u++) memcpy(out->balance_cts + 64ull * hb.slot_of_used[u], fin + 64 * u, 64);
            }
            if (out->account_nonces) memcpy(out->account_nonces, B->in_nonces.data(), 8 * B->n_accounts);
            if (out->multisig_set_by) memcpy(out->multisig_set_by, hb.multisig_set_by.data(), 8 * B->n_accounts);
        }
        if (r == 0) { verdicts[k] = v; decided[k] = 1; batch_free_finished(ctx, B); up[k] = nullptr; }      // (after an error the batch is freed below, behind a drain of the streams)
        else if (failed_batch < 0) failed_batch = (int64_t)k;
        mark_verified(k);
        return r;
    };
    int64_t in_flight = -1;
    for (uint64_t k = 0; k < n_batches && rc == 0; k++) {
        // whichever comes first: the batch's upload, or the end of the batch in flight (an upload that has to rebuild the
        // context's tables waits for the verifications in flight)
        while (in_flight >= 0) {
            { std::unique_lock<std::mutex> lk(mu); if (cv.wait_for(lk, std::chrono::microseconds(200), [&]() { return uploaded > k; })) break; }
            if (hipEventQuery(up[(size_t)in_flight]->ev_done) != hipErrorNotReady) { rc = finish_batch((uint64_t)in_flight); in_flight = -1; }
        }
        if (rc != 0) break;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return uploaded > k; }); }
        at(k, "uploaded");
        rc = up_rc[k];
        if (rc != 0) {
            // the batch in flight was enqueued before this upload failed: its verdict is still worth having
            failed_batch = (int64_t)k;
            if (in_flight >= 0) { (void)finish_batch((uint64_t)in_flight); in_flight = -1; }
            break;
        }
        if (!overlap && in_flight >= 0) { rc = finish_batch((uint64_t)in_flight); in_flight = -1; if (rc != 0) break; }
        xhe_batch *B = up[k];
        if (!structural[k].accept) {
            // malformed: the upload has said so; nothing to verify
            if (in_flight >= 0) { rc = finish_batch((uint64_t)in_flight); in_flight = -1; }
            if (B) { xhe_batch_free(ctx, B); up[k] = nullptr; }
            verdicts[k] = structural[k]; decided[k] = 1;
            mark_verified(k);
            continue;
        }
        rc = batch_verify_enqueue(ctx, B, batches[k].rng_seed, overlap, in_flight < 0);
        if (rc == ENQUEUE_TABLES_BUSY) {
            rc = finish_batch((uint64_t)in_flight); in_flight = -1;
            if (rc == 0) rc = batch_verify_enqueue(ctx, B, batches[k].rng_seed, overlap, true);
        }
        if (rc != 0 && failed_batch < 0) failed_batch = (int64_t)k;
        if (rc == 0) {
            // the resulting ciphertexts of the slots the batch used, into pinned memory (the slot of batch k - 2, long read)
            const u64 n_slots = B->hb.slot_of_used.size();
            hipError_t e = hipSuccess;
            if (batches[k].out && n_slots) {
                if (ctx->readback_cap[k & 1] < 64 * n_slots) {
                    if (ctx->readback[k & 1]) (void)hipHostFree(ctx->readback[k & 1]);
                    ctx->readback[k & 1] = nullptr; ctx->readback_cap[k & 1] = 0;
                    e = hipHostMalloc((void **)&ctx->readback[k & 1], 64 * n_slots + 4096, hipHostMallocDefault);
                    if (e == hipSuccess) ctx->readback_cap[k & 1] = 64 * n_slots + 4096;
                }
                if (e == hipSuccess) e = hipMemcpyAsync(ctx->readback[k & 1], B->final_cts.p, 64 * n_slots, hipMemcpyDeviceToHost, s0);
            }
            if (e == hipSuccess) e = hipEventRecord(B->ev_carried, s0);
            if (e != hipSuccess) { set_error(ctx, std::string("reading a batch's ledger back: ") + hipGetErrorString(e)); rc = XHE_E_HIP; if (failed_batch < 0) failed_batch = (int64_t)k; }
        }
        at(k, "enqueued");
        if (rc == 0 && in_flight >= 0) { rc = finish_batch((uint64_t)in_flight); in_flight = -1; }
        if (rc == 0) in_flight = (int64_t)k;
    }
    if (rc == 0 && in_flight >= 0) { rc = finish_batch((uint64_t)in_flight); in_flight = -1; }
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    uploader.join();
    bool any_left = false;
    for (auto *b : up) if (b) any_left = true;
    if (any_left) for (hipStream_t s : {ctx->stream, ctx->side[0], ctx->side[1], ctx->side[2]}) (void)hipStreamSynchronize(s);
    for (auto *&b : up) if (b) { b->in_flight = false; xhe_batch_free(ctx, b); b = nullptr; }
    if (rc != 0) {
        // every batch without a finished verification reads reject {0, XHE_ERR_STATE, -1} (set above); name the batch the call stopped at
        for (uint64_t k = 0; k < n_batches; k++) if (!decided[k]) verdicts[k] = {0, XHE_ERR_STATE, -1};
        std::string msg;
        { std::lock_guard<std::mutex> lk(ctx->err_mu); msg = ctx->error; }
        set_error(ctx, "xhe_verify_batches stopped at batch " + std::to_string((long long)failed_batch) + ": " + msg);
    }
    return rc;
}

}  // extern "C"

// The launch sequence of one verification, without waiting for it: batch_verify_finish does.  A second batch's sequence may be
// enqueued behind it before it has finished (every stream is a queue; what the two share beyond the streams -- the bucket
// workspace -- is used on the main stream only).
// may_block = false (a caller that holds another verification in flight): when the context's tables are being rebuilt, or a
// rebuild is waiting for the verifications in flight, the shared lock is not waited for -- ENQUEUE_TABLES_BUSY tells the caller to
// finish what it has in flight (which lets the rebuild through) and to call again.  Waiting here instead would rely on the
// lock letting new readers pass a waiting writer.
//
// lane (parts of a batch verified in parts): consecutive parts alternate between two lanes.  Lane 1 swaps the roles of the main
// stream and side stream 0 and uses a bucket workspace of its own, so the main chain of part k + 1 (transcripts, proof scalars, bucket
// sort -- none of which needs a decoded point) queues behind the short ledger and static-generator work of part k instead of behind
// part k's whole multiscalar check, and runs beside it; part k + 1's ledger scan in turn queues behind part k's main chain, which
// it could not overtake anyway (its inputs are the ciphertexts part k leaves).  For that the part's early decode (the ledger's
// and the transfers' ciphertexts) moves to side stream 1, in front of the bulk decode.  Parts two apart share streams and
// workspace in stream order.
// lane_or_none = -1: no lanes (a single verification, or a run of independent batches); carry_ledger: the device array of the whole
// ledger's ciphertexts that a part writes the slots it used into, right behind its ledger scan (B->ev_carried fires then).
static int32_t batch_verify_enqueue(xhe_ctx *ctx, xhe_batch *B, const uint8_t seed[32], bool setup_on_side /* one of the parts of a batch verified in parts */, bool may_block,
                                    int lane_or_none, u32 *carry_ledger, const ledger_handover *handover) {
    const host_batch &hb = B->hb;
    // XHE_ONE_STREAM=1 (diagnostics): everything on the main stream, so that a kernel trace shows every kernel's stand-alone time
    const bool one_stream = setting("XHE_ONE_STREAM") != nullptr;
    const bool lanes = lane_or_none >= 0 && !one_stream;
    const u32 lane = lanes ? (u32)lane_or_none & 1u : 0u;
    hipStream_t st = lane ? ctx->side[0] : ctx->stream;
    B->verified_ok = false; B->in_flight = false; B->trivial = false;
    B->lane = lane; B->main_stream = st; B->ledger_stream = one_stream ? st : (lane ? ctx->stream : ctx->side[0]);
    if (!hb.structural.accept) return 0;
    const u32 n_txs = (u32)hb.n_txs, n_cm = hb.n_cm, n_xf = hb.n_xf, n_dec = hb.n_decs;
    const u32 n_ops = hb.n_ops, n_slots = (u32)hb.slot_of_used.size(), n_sigs = hb.n_sigs;
    const u32 n_parts = n_cm + n_xf + n_txs;
    stage_timer &tm = ctx->timer;
    tm.begin();
    // XHE_ENQUEUE_TIMES=1: host clock at the points of the launch sequence (how far ahead of the device the enqueueing runs)
    const bool host_times = setting("XHE_ENQUEUE_TIMES") != nullptr;
    const auto h0 = std::chrono::steady_clock::now();
    auto host_mark = [&](const char *what) {
        if (host_times) fprintf(stderr, "[xhe enqueue] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count());
    };
    if (may_block) B->tables_lock = std::shared_lock<std::shared_mutex>(ctx->tables_mu);
    else {
        B->tables_lock = std::shared_lock<std::shared_mutex>(ctx->tables_mu, std::try_to_lock);
        if (!B->tables_lock.owns_lock()) return ENQUEUE_TABLES_BUSY;
    }
    B->in_flight = true;
    // The fills that open a verification run on the main stream, or (setup_on_side: a batch verified in parts) on side stream 1,
    // which is free long before the main stream has finished the part in front: the part's decoding and signatures then start
    // beside the multiscalar check of the part before.
    hipStream_t s_setup = (setup_on_side && !one_stream) ? ctx->side[1] : st;
    HIPCHK(ctx, hipStreamWaitEvent(st, B->ev_uploaded, 0));
    if (s_setup != st) HIPCHK(ctx, hipStreamWaitEvent(s_setup, B->ev_uploaded, 0));
    HIPCHK(ctx, tm.mark("setup", st));
    {
        setup_jobs jobs;
        memcpy(jobs.seed, seed, 32); jobs.seed_dst = B->seed.as<u32>();
        // the static generators of this batch's point array: B, H, G_vec[nm_cap], H_vec[nm_cap]
        jobs.copy(B->pts.p, ctx->static_pts.p, 64 + 32ull * hb.nm_cap);
        jobs.copy(B->pts.as<u32>() + 64 + 32ull * hb.nm_cap, ctx->static_pts.as<u32>() + 64 + 32ull * ctx->static_cap, 32ull * hb.nm_cap);
        jobs.fill(B->ok.p, 0x01010101u, (hb.n_points + 3) / 4);
        jobs.copy(B->tx_err.p, B->tx_err0.p, std::max<u32>(n_txs, 1));
        jobs.fill(B->rp_err.p, 0, std::max<u32>(n_txs, 1));
        jobs.fill(B->rp_acc.p, 0, 18ull * hb.nm_cap);
        jobs.fill(B->emit_incl.p, 0, 64ull * (n_cm + 1));
        jobs.fill(B->slot_base.p, 0, 64ull * (n_slots + 1));
        jobs.fill(B->slot_final.p, 0, 64ull * (n_slots + 1));
        jobs.fill(B->pts.as<u32>() + 32ull * hb.emit_base, 0, 32ull * hb.n_emit);
        jobs.fill(B->verdict.p, 0xffffffffu, 4);
        jobs.fill(B->verdict.as<u32>() + 4, 0, 12);
        launch_setup(s_setup, jobs);
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_setup, s_setup));
    if (s_setup != st) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_setup, 0));
    if (n_txs == 0) {
        B->trivial = true;
        HIPCHK(ctx, tm.mark("end", st));
        HIPCHK(ctx, hipEventRecord(B->ev_done, st));
        return 0;
    }
    const u8 *bytes = B->bytes.as<u8>();
    const dev_tx *txs = B->txs.as<dev_tx>(); const dev_cm *cms = B->cms.as<dev_cm>(); const dev_xfer *xfers = B->xfers.as<dev_xfer>();
    u32 *pts = B->pts.as<u32>(); u8 *ok = B->ok.as<u8>(); u32 *chal = B->chal.as<u32>(); u32 *scalars = B->scalars.as<u32>();
    u32 *term_point = B->term_point.as<u32>(); u32 *gh = B->gh.as<u32>(); u32 *tx_err = B->tx_err.as<u32>(); u32 *rp_err = B->rp_err.as<u32>();
    const u32 *sd = B->seed.as<u32>();

    // Three streams (four for large batches, see below): the device exposes four hardware queues to a process, streams that
    // share a queue run one after the other, and the caller (its own stream, RCCL) needs one too.  Enqueue order = who gets
    // wave slots first.
    //   main     transcripts -> proof scalars -> multiscalar check -> verdict
    //   side 0   ledger inputs -> ledger scan (its points feed the multiscalar check) -> static-generator scalars -> ledger encodings, point errors
    //   side 1   every other point of the batch (public keys included) -> signature points + hashing -> end of the signature hashes
    hipStream_t s0 = B->ledger_stream, s1 = one_stream ? st : ctx->side[1];
    // a part of a batch verified in parts decodes the ledger scan's inputs on side stream 1 (see `lane` above)
    const bool early_on_s1 = lanes;
    HIPCHK(ctx, tm.mark("transcript", st));
    launch_transcript(st, bytes, txs, cms, xfers, B->tmpl.as<u32>(), B->segs.as<u32>(), hb.tr_plan, B->tr_list.as<u32>(), B->tr_tiles.as<transcript_tile>(),
                      B->tr_stream.as<u64>(), B->serial_txs.as<u32>(), (u32)hb.serial_txs.size(), chal, tx_err, rp_err);
    // side 0: ledger and transfer ciphertexts, then the ledger as a prefix sum
    // side 1: the bulk of the points, then the signatures; from sig_stream_min_txs() transactions on the signatures start at
    // once on a stream of their own (side 2, with the public keys decoded there).
    // The host needs several microseconds per launch, so the order below is also the order in which the work can start: the
    // decodes first (the bulk decode is the longest side job), the ledger scan (which waits for its inputs anyway) after them.
    const bool own_sig_stream = n_txs >= sig_stream_min_txs();
    hipStream_t ssig = one_stream ? st : (own_sig_stream ? ctx->side[2] : s1);
    HIPCHK(ctx, hipStreamWaitEvent(s0, ctx->ev_setup, 0));
    if (early_on_s1) {
        // the transfers' ciphertexts on side stream 1; the ledger's slots (the tail of the list) on the ledger stream, behind the
        // hand-over from the part in front: nothing else of this part waits for that part's ledger
        HIPCHK(ctx, hipStreamWaitEvent(s1, ctx->ev_setup, 0));
        launch_decode_list(s1, bytes, B->decs.as<dev_dec>(), B->early_decs.as<u32>(), hb.early_end, hb.n_static, pts, ok, false, !lanes);
        HIPCHK(ctx, hipEventRecord(ctx->ev_early, s1));
        if (handover && handover->bytes) {
            if (handover->after) HIPCHK(ctx, hipStreamWaitEvent(s0, handover->after, 0));
            HIPCHK(ctx, hipMemcpyAsync(B->bytes.as<u8>() + hb.cts_base, handover->src, handover->bytes, hipMemcpyDeviceToDevice, s0));
        }
        launch_decode_list(s0, bytes, B->decs.as<dev_dec>(), B->early_decs.as<u32>() + hb.early_end, hb.n_early - hb.early_end, hb.n_static, pts, ok, false, !lanes);
        HIPCHK(ctx, hipStreamWaitEvent(s0, ctx->ev_early, 0));
    } else {
        HIPCHK(ctx, tm.open("decode_first", s0));
        launch_decode_list(s0, bytes, B->decs.as<dev_dec>(), B->early_decs.as<u32>(), hb.n_early, hb.n_static, pts, ok, false, !lanes);
        HIPCHK(ctx, tm.close("decode_first", s0));
    }
    auto enqueue_scan = [&]() -> int32_t {
        HIPCHK(ctx, tm.open("ledger_scan", s0));
        launch_ledger_scan(s0, B->ops.as<dev_op>(), n_ops, pts, ctx->gtab.as<u32>(), B->thread_excl.as<u32>(), B->block_tot.as<u32>(), B->slot_last_op.as<u32>(),
                           B->slot_base.as<u32>(), B->slot_final.as<u32>(), B->emit_incl.as<u32>(), B->emit_slot.as<u32>(), n_cm, hb.emit_base, pts, ok);
        HIPCHK(ctx, tm.close("ledger_scan", s0));
        HIPCHK(ctx, hipEventRecord(ctx->ev_side_done[0], s0));         // ledger inputs and the senders' new ciphertexts are in place
        // a batch verified in parts: the next part waits for the resulting ledger ciphertexts (otherwise they are written at the
        // end of this stream's work, off the path to the verdict)
        if (n_ops && setup_on_side) launch_ledger_final(s0, B->slot_final.as<u32>(), B->slot_base.as<u32>(), n_slots, B->final_cts.as<u32>());
        if (carry_ledger) {
            launch_ledger_carry(s0, B->final_cts.as<u32>(), B->slot_of_used.as<u32>(), n_slots, carry_ledger);
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipEventRecord(B->ev_carried, s0));
        }
        return 0;
    };
    HIPCHK(ctx, hipStreamWaitEvent(s1, ctx->ev_setup, 0));
    if (own_sig_stream) HIPCHK(ctx, hipStreamWaitEvent(ssig, ctx->ev_setup, 0));
    HIPCHK(ctx, tm.open("signatures", ssig));
    launch_decode_list(ssig, bytes, B->decs.as<dev_dec>(), B->key_decs.as<u32>(), hb.n_key, hb.n_static, pts, ok, false, !lanes);
    if (own_sig_stream) HIPCHK(ctx, hipEventRecord(ctx->ev_keys, ssig));
    HIPCHK(ctx, tm.open("decode_rest", s1));
    launch_decode_list(s1, bytes, B->decs.as<dev_dec>(), B->bulk_decs.as<u32>(), hb.n_bulk, hb.n_static, pts, ok, true, !lanes);
    HIPCHK(ctx, tm.close("decode_rest", s1));
    if (own_sig_stream) HIPCHK(ctx, hipStreamWaitEvent(s1, ctx->ev_keys, 0));
    HIPCHK(ctx, hipEventRecord(ctx->ev_side_done[2], s1));             // every point that is not the ledger scan's
    launch_blake3_preimages(ssig, bytes, txs, B->ms_hash_tx.as<u32>(), (u32)hb.ms_hash_tx.size(), B->ms_hash.as<u8>());
    // A table of multiples per signing key costs 192 doublings and 36 additions per key and saves 192 doublings per signature,
    // but the signatures wait for it: it pays when the keys sign many transactions each (measured on one device: 9.8 signatures
    // per key 1.45 -> 1.37 ms; 4 per key with long messages and 2.5 per key slower by 3-4 %).  XHE_SIG_KEY_TABLES = 0 / 1 forces the choice.
    bool key_tables = (u64)n_sigs >= 6ull * hb.sig_keys.size();
    if (const char *e = setting("XHE_SIG_KEY_TABLES")) key_tables = e[0] == '1';
    if (key_tables) launch_sig_key_tables(ssig, B->sig_keys.as<u32>(), (u32)hb.sig_keys.size(), pts, B->sig_tab.as<u32>());
    launch_sig_point_absorb(ssig, bytes, B->sigs.as<dev_sig>(), n_sigs, key_tables ? B->sig_tab.as<u32>() : nullptr, pts, ctx->htab.as<u32>(), B->sig_r.as<u32>(),
                            B->sig_state.as<u64>());
    launch_sig_final(ssig, bytes, B->sigs.as<dev_sig>(), n_sigs, B->sig_state.as<u64>(), B->sig_r.as<u32>(), ok, B->ms_hash.as<u8>(), tx_err);
    HIPCHK(ctx, tm.close("signatures", ssig));
    HIPCHK(ctx, hipEventRecord(ctx->ev_side_done[1], ssig));
    if (int32_t rc = enqueue_scan()) return rc;
    host_mark("side streams enqueued");
    // main: proof scalars (sigma terms, range-proof terms, per-transaction range-proof context)
    HIPCHK(ctx, tm.mark("proof_scalars", st));
    {
        const bool tables = ctx->gen_tab_cap != 0 && hb.nm_cap <= ctx->gen_tab_cap;
        proof_scalar_args pa{bytes, txs, cms, xfers, n_cm, n_xf, n_txs, chal, sd, scalars, term_point, B->rp_ctx.as<u32>(), gh, rp_err,
                             sigma_weight_bits(msm_make_plan(tables ? hb.static_term_off : hb.n_terms).c)};
        launch_proof_scalars(st, pa);
        // terms on one point become one term: a value commitment's sigma and range-proof terms, every term on an account's key
        const char *merge_env = setting("XHE_MERGE_TERMS");
        const bool merge_terms = !(merge_env && merge_env[0] == '0');
        if (merge_terms) launch_merge_terms(st, B->merge_pairs.as<u32>(), hb.n_cm + hb.n_xf, B->key_group_off.as<u32>(), (u32)hb.key_group_off.size() - 1, B->key_terms.as<u32>(), scalars);
    }
    // The scalars of the static generators (column sums over all range proofs, sums of the per-proof B and H
    // parts).  With fixed-base tables their whole contribution is evaluated on side 0 while the main stream
    // sorts and accumulates the dynamic terms; without, they join the multiscalar check as terms.
    const bool use_tables = ctx->gen_tab_cap != 0 && hb.nm_cap <= ctx->gen_tab_cap;
    hipStream_t ss = use_tables ? s0 : st;
    if (use_tables) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_scalars, st));
        HIPCHK(ctx, hipStreamWaitEvent(ss, ctx->ev_scalars, 0));
        HIPCHK(ctx, tm.open("static_scalars", ss));
    }
    for (u32 lg = 6; lg < 16; lg++) {
        u32 rows = (u32)hb.lg_groups[lg].size();
        if (!rows) continue;
        if (rows >= B->rp_gemm_min_rows) {
            launch_rp_gemm(ss, B->group_txs.as<u32>() + B->group_off[lg], rows, lg, B->rp_ctx.as<u32>(), rp_err, B->rp_dig.as<u8>(), B->rp_bins.as<int>(), B->rp_rho.as<u32>(), hb.nm_cap,
                           B->rp_acc.as<u32>());
            continue;
        }
        launch_rp_matrix(ss, B->group_txs.as<u32>() + B->group_off[lg], rows, rp_rows_per_thread(rows, lg), lg, B->rp_ctx.as<u32>(), rp_err, B->rp_mat.as<u32>());
        launch_rp_colsum(ss, B->rp_mat.as<u32>(), (u32)rp_matrix_part_rows(rows, lg), lg, hb.nm_cap, B->rp_acc.as<u32>());
    }
    u32 sum_blocks = std::min<u32>(128, (n_parts + 255) / 256);
    launch_sc_sum_partial(ss, sum_blocks, gh, n_parts, 2u, B->gh_sums.as<u32>());
    launch_static_terms(ss, B->gh_sums.as<u32>(), sum_blocks, B->rp_acc.as<u32>(), hb.nm_cap, hb.static_term_off, scalars, term_point);
    if (use_tables) {
        launch_static_eval(ss, ctx->gen_tab.as<u32>(), scalars + 8ull * hb.static_term_off, hb.nm_cap, ctx->gen_tab_cap, B->static_partials.as<u32>(),
                           B->static_sum.as<u32>());
        HIPCHK(ctx, tm.close("static_scalars", ss));
        HIPCHK(ctx, hipEventRecord(ctx->ev_static_done, ss));
    }
    // side 0 again: resulting ledger ciphertexts, point-validity errors (need every point)
    if (n_ops && !setup_on_side) launch_ledger_final(s0, B->slot_final.as<u32>(), B->slot_base.as<u32>(), n_slots, B->final_cts.as<u32>());
    HIPCHK(ctx, hipStreamWaitEvent(s0, ctx->ev_side_done[2], 0));
    launch_point_errors(s0, txs, cms, xfers, n_txs, ok, tx_err, rp_err);
    HIPCHK(ctx, hipEventRecord(ctx->ev_points_checked, s0));
    HIPCHK(ctx, hipGetLastError());
    host_mark("static chain enqueued");
    // merged multiscalar check: sigma terms + range-proof terms (+ static generators, unless their sum comes from the tables)
    msm_deps deps;
    deps.points_ready[0] = ctx->ev_side_done[0]; deps.points_ready[1] = ctx->ev_side_done[2];
    if (use_tables) { deps.extra_point = B->static_sum.as<u32>(); deps.extra_ready = ctx->ev_static_done; }
    B->acc_timed = false;
    if (ctx->stats_on) {
        for (hipEvent_t &e : B->ev_acc) if (!e) HIPCHK(ctx, hipEventCreate(&e));
        B->acc_timed = true;
    }
    int32_t rc = msm_device(ctx, scalars, pts, term_point, hb.n_points, use_tables ? hb.static_term_off : hb.n_terms, B->verdict.as<u32>() + 4, false, true, deps, st, lane,
                            B->acc_timed ? B->ev_acc : nullptr);
    if (rc) return rc;
    host_mark("multiscalar check enqueued");
    HIPCHK(ctx, tm.mark("join_signatures", st));
    HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_side_done[1], 0));
    HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_points_checked, 0));
    HIPCHK(ctx, tm.mark("verdict", st));
    launch_verdict(st, tx_err, rp_err, n_txs, B->verdict.as<unsigned long long>(), B->verdict.as<u32>() + 2, B->noncanonical.as<u32>());
    HIPCHK(ctx, tm.mark("end", st));
    B->verdict_words = ctx->verdict_host + 16 * (ctx->verdict_next++ % VERDICT_SLOTS);
    HIPCHK(ctx, hipMemcpyAsync(B->verdict_words, B->verdict.p, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipEventRecord(B->ev_done, st));
    host_mark("all enqueued");
    return 0;
}

// Waits for the verification batch_verify_enqueue started and reads its verdict; when a proof is wrong, decides which error the
// reference reports with a second, sigma-only check.
static int32_t batch_verify_finish(xhe_ctx *ctx, xhe_batch *B, xhe_verdict *verdict) {
    const host_batch &hb = B->hb;
    hipStream_t st = B->main_stream ? B->main_stream : ctx->stream;
    if (!hb.structural.accept) { *verdict = hb.structural; return 0; }
    struct unlock_on_exit { xhe_batch *b; ~unlock_on_exit() { b->in_flight = false; if (b->tables_lock.owns_lock()) b->tables_lock.unlock(); } } unlock{B};
    if (!B->in_flight) { set_error(ctx, "no verification of this batch is in flight"); return XHE_E_ARG; }
    HIPCHK(ctx, hipEventSynchronize(B->ev_done));
    ctx->timer.finish();
    if (B->trivial) { *verdict = {1, XHE_OK, -1}; B->verified_ok = true; return 0; }
    const u32 n_txs = (u32)hb.n_txs, n_cm = hb.n_cm, n_xf = hb.n_xf;
    const u8 *bytes = B->bytes.as<u8>();
    const dev_tx *txs = B->txs.as<dev_tx>(); const dev_cm *cms = B->cms.as<dev_cm>(); const dev_xfer *xfers = B->xfers.as<dev_xfer>();
    u32 *pts = B->pts.as<u32>(); u32 *chal = B->chal.as<u32>(); u32 *scalars = B->scalars.as<u32>();
    u32 *term_point = B->term_point.as<u32>(); u32 *gh = B->gh.as<u32>(); u32 *rp_err = B->rp_err.as<u32>();
    const u32 *sd = B->seed.as<u32>();
    int32_t rc = 0;
    const u32 *host = B->verdict_words;
    u64 first_err = (u64)host[0] | ((u64)host[1] << 32);
    u32 first_rp = host[2], merged_identity = host[12];
    if (B->acc_timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, B->ev_acc[0], B->ev_acc[1]) == hipSuccess) { ctx->stats.launches++; ctx->stats.accumulate_ms += ms; ctx->stats.entries += host[13]; ctx->stats.terms_nonzero += host[14]; }
        B->acc_timed = false;
    }
    // a scalar that is not canonical: the reference would not have deserialised the record, whatever else is wrong with the batch
    if (host[3] != 0xffffffffu) { *verdict = {0, XHE_ERR_MALFORMED, (int64_t)host[3]}; return 0; }
    if (first_err != ~0ull) { *verdict = {0, (int32_t)(first_err & 0xff), (int64_t)(first_err >> 32)}; return 0; }
    if (merged_identity && first_rp == 0xffffffffu) { *verdict = {1, XHE_OK, -1}; B->verified_ok = true; return 0; }
    // Slow path (a proof is wrong): decide between GenericProof and RangeProof as the reference orders them
    // (verify.rs:500-514) with the sigma-only multiscalar check.
    {
        // the merged scalars mix sigma-proof and range-proof terms: the sigma-only check runs on the scalars as the proofs give them
        proof_scalar_args pa{bytes, txs, cms, xfers, n_cm, n_xf, n_txs, chal, sd, scalars, term_point, B->rp_ctx.as<u32>(), gh, rp_err,
                             sigma_weight_bits(msm_make_plan((u64)hb.sigma_terms + 2).c)};
        launch_proof_scalars(st, pa);
        u32 n_sig_terms = hb.sigma_terms;
        dev_buf s2, tp2, r2;
        struct release_on_exit { dev_buf &a, &b, &c; ~release_on_exit() { a.release(); b.release(); c.release(); } } temporaries{s2, tp2, r2};   // also on an early error return
        HIPCHK(ctx, s2.reserve(32ull * (n_sig_terms + 2))); HIPCHK(ctx, tp2.reserve(4ull * (n_sig_terms + 2))); HIPCHK(ctx, r2.reserve(64));
        HIPCHK(ctx, hipMemcpyAsync(s2.p, scalars, 32ull * n_sig_terms, hipMemcpyDeviceToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(tp2.p, term_point, 4ull * n_sig_terms, hipMemcpyDeviceToDevice, st));
        u32 sb = std::min<u32>(128, std::max<u32>(1, (n_cm + n_xf + 255) / 256));
        launch_sc_sum_partial(st, sb, gh, n_cm + n_xf, 2u, B->gh_sums.as<u32>());
        // reuse k_static_terms with nm_cap = 0: writes only the B and H scalars
        launch_static_terms(st, B->gh_sums.as<u32>(), sb, B->rp_acc.as<u32>(), 0u, n_sig_terms, s2.as<u32>(), tp2.as<u32>());
        HIPCHK(ctx, hipMemsetAsync(r2.p, 0, 64, st));
        rc = msm_device(ctx, s2.as<u32>(), pts, tp2.as<u32>(), hb.n_points, (u64)n_sig_terms + 2, r2.as<u32>(), false, false, msm_deps(), st, B->lane);
        u32 h2[16];
        if (rc == 0) { hipError_t e = hipMemcpyAsync(h2, r2.p, 64, hipMemcpyDeviceToHost, st); if (e == hipSuccess) e = hipStreamSynchronize(st); if (e != hipSuccess) rc = XHE_E_HIP; }
        if (rc) return rc;
        if (!h2[8]) { *verdict = {0, XHE_ERR_GENERIC_PROOF, -1}; return 0; }
    }
    *verdict = {0, XHE_ERR_RANGE_PROOF, first_rp == 0xffffffffu ? (int64_t)-1 : (int64_t)first_rp};
    return 0;
}

// The launch sequence of one verification and its verdict.  Everything is enqueued on the context's streams; the only host
// synchronisation is the wait for the 64-byte read-back of the verdict words.
static int32_t batch_verify_impl(xhe_ctx *ctx, xhe_batch *B, xhe_verdict *verdict, const uint8_t seed[32]) {
    int32_t rc = batch_verify_enqueue(ctx, B, seed, false);
    if (rc != 0) {
        // an error return leaves in the middle of the launch sequence: nothing of it may still run when the caller goes on
        for (hipStream_t s : {ctx->stream, ctx->side[0], ctx->side[1], ctx->side[2]}) (void)hipStreamSynchronize(s);
        B->in_flight = false; if (B->tables_lock.owns_lock()) B->tables_lock.unlock();
        return rc;
    }
    return batch_verify_finish(ctx, B, verdict);
}
