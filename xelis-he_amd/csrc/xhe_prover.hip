// xhe_range_proofs_prove (include/xhe_builder.h): batched aggregated range proofs on the context's GPU.  Host side only:
// chunking, device buffers, copies; the arithmetic is in xhe_k_prover.hip.  Generates inputs for the verification
// path (the counterpart of RangeProof::prove_multiple at /root/reference/src/tx/builder.rs:525-533), never runs inside it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "../../include/xhe_builder.h"
#include "xhe_ctx.h"

using namespace xhe;

extern "C" int32_t xhe_range_proofs_prove(xhe_ctx *ctx, uint64_t n, uint32_t m, const uint8_t *transcripts, const uint64_t *values, const uint8_t *openings32,
                                          const uint8_t *seeds32, const uint64_t *counters, uint8_t *proofs) {
    if (!ctx || (n && (!transcripts || !values || !openings32 || !seeds32 || !counters || !proofs))) return XHE_E_ARG;
    if (m == 0 || (m & (m - 1)) || m > XHE_RANGE_PROOF_GPU_MAX_VALUES) { set_error(ctx, "xhe_range_proofs_prove: m must be a power of two, at most XHE_RANGE_PROOF_GPU_MAX_VALUES"); return XHE_E_ARG; }
    if (n == 0) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const u32 nm = 64 * m;
    u32 lg = 6; while ((1u << lg) < nm) lg++;
    if (int32_t rc = ensure_tables(ctx, nm)) return rc;
    std::shared_lock<std::shared_mutex> tables_lock(ctx->tables_mu);
    if (ctx->gen_tab_cap < nm) { set_error(ctx, "xhe_range_proofs_prove: no fixed-base tables for this many generators"); return XHE_E_ARG; }
    const u64 proof_bytes = 32ull * (9 + 2 * lg);
    // proofs per chunk: the three scalar vectors (36 bytes per position each) stay below ~1.5 GiB; the grid's y dimension below 2^15
    u64 chunk = (3ull << 29) / (108ull * nm);
    chunk = std::max<u64>(64, std::min<u64>(chunk, 32768));
    chunk = std::min(chunk, n);
    const u32 segs = range_proof_as_segments(nm), slots = range_proof_lr_slots(nm);
    dev_buf arena; arena.pool = &ctx->pool;
    struct piece { size_t off, bytes; };
    size_t total = 0;
    auto add = [&](u64 bytes) { piece p{total, (size_t)bytes}; total += (size_t)((bytes + 255) & ~255ull); return p; };
    const piece p_seeds = add(32 * chunk), p_ctr = add(8 * chunk), p_values = add(8ull * m * chunk), p_open = add(32ull * m * chunk),
                p_tstate = add(4ull * RP_TSTATE_WORDS * chunk), p_sc = add(36ull * RPS_COUNT * chunk), p_a = add(36ull * nm * chunk), p_b = add(36ull * nm * chunk),
                p_y = add(36ull * nm * chunk), p_pa = add(128ull * segs * chunk), p_ps = add(128ull * segs * chunk), p_lr = add(256ull * slots * chunk),
                p_proofs = add(proof_bytes * chunk);
    HIPCHK(ctx, arena.reserve(total));
    auto at = [&](const piece &p) { return (char *)arena.p + p.off; };
    hipStream_t st = ctx->stream;
    int32_t rc = 0;
    auto chk = [&](hipError_t e, const char *what) { if (e != hipSuccess && rc == 0) { set_error(ctx, std::string(what) + ": " + hipGetErrorString(e)); rc = XHE_E_HIP; } return e == hipSuccess; };
    // the caller's transcripts are 208 bytes each: 200 bytes of state, pos, pos_begin, padding = RP_TSTATE_WORDS words
    static_assert(RP_TSTATE_WORDS * 4 == XHE_RANGE_PROOF_TRANSCRIPT_BYTES, "transcript record size");
    for (u64 lo = 0; lo < n && rc == 0; lo += chunk) {
        const u64 c = std::min(chunk, n - lo);
        chk(hipMemcpyAsync(at(p_seeds), seeds32 + 32 * lo, 32 * c, hipMemcpyHostToDevice, st), "copying seeds");
        chk(hipMemcpyAsync(at(p_ctr), counters + lo, 8 * c, hipMemcpyHostToDevice, st), "copying counters");
        chk(hipMemcpyAsync(at(p_values), values + (u64)m * lo, 8ull * m * c, hipMemcpyHostToDevice, st), "copying values");
        chk(hipMemcpyAsync(at(p_open), openings32 + 32ull * m * lo, 32ull * m * c, hipMemcpyHostToDevice, st), "copying openings");
        chk(hipMemcpyAsync(at(p_tstate), transcripts + (u64)XHE_RANGE_PROOF_TRANSCRIPT_BYTES * lo, (u64)XHE_RANGE_PROOF_TRANSCRIPT_BYTES * c, hipMemcpyHostToDevice, st), "copying transcripts");
        if (rc) break;
        rp_dims d{(u32)c, m, nm, lg, ctx->gen_tab_cap};
        rp_buffers b;
        b.seeds = (const u32 *)at(p_seeds); b.ctr0 = (const u64 *)at(p_ctr); b.values = (const u64 *)at(p_values); b.openings = (const u32 *)at(p_open);
        b.tstate = (u32 *)at(p_tstate); b.sc_store = (u32 *)at(p_sc); b.vec_a = (u32 *)at(p_a); b.vec_b = (u32 *)at(p_b); b.vec_yinv = (u32 *)at(p_y);
        b.part_a = (u32 *)at(p_pa); b.part_s = (u32 *)at(p_ps); b.part_lr = (u32 *)at(p_lr); b.proofs = (u32 *)at(p_proofs);
        launch_range_proofs(st, d, ctx->gen_tab.as<u32>(), b);
        chk(hipGetLastError(), "launching the range-proof kernels");
        chk(hipMemcpyAsync(proofs + proof_bytes * lo, at(p_proofs), proof_bytes * c, hipMemcpyDeviceToHost, st), "copying proofs back");
        chk(hipStreamSynchronize(st), "range-proof generation");
    }
    (void)hipStreamSynchronize(st);
    arena.release();
    return rc;
}
