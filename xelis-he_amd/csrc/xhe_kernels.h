// Host-callable launchers of the gfx950 kernels.  Each xhe_k_*.hip translation unit defines its kernels
// and the launchers declared here; xhe_lib.hip only sequences them on the context's stream.
#pragma once
#include <algorithm>
#include <hip/hip_runtime_api.h>
#include "xhe_verify_types.h"

namespace xhe {

struct msm_plan {
    u32 c;        // window bits
    u32 W;        // windows
    u32 B;        // buckets per window = 2^(c-1)
    u32 L;        // buckets per lane in the reduce stage
    u32 S;        // segments per window = B / (64 L)   (<= 64)
    u32 M;        // sorted entries per accumulate chunk
    u32 tile;     // terms per workgroup of the counting sort
    u32 fine_bits;  // 0: one-pass counting sort over all B buckets; f > 0: two passes, B >> f coarse bins, then 2^f buckets inside each bin
};
msm_plan msm_make_plan(u64 n);
// device pointers of one multiscalar multiplication's workspace (sizes: see msm_workspace_bytes)
struct msm_ws {
    u32 *hist, *offsets, *tile_hist, *sorted, *buckets, *digits, *chunk_sums, *big_list, *segs, *wins, *tmp, *coarse;
};
struct msm_ws_bytes { u64 hist, offsets, tile_hist, sorted, buckets, digits, chunk_sums, big_list, segs, wins, tmp, coarse; };
msm_ws_bytes msm_workspace_bytes(u64 n, const msm_plan &p);
typedef hipError_t (*msm_mark_fn)(void *user, const char *name);
// result: identity flag at d_result[8]; with want_encoding also the 8 words of the sum's encoding at d_result[0..8); d_result[9] = sorted
// entries the accumulation went through, with count_terms d_result[10] += terms with a non-zero scalar (the caller clears d_result before the launch)
// What the multiscalar multiplication waits for on other streams: the sort of the scalars needs none of it.
struct msm_deps {
    hipEvent_t points_ready[2] = {nullptr, nullptr};   // waited for before the first point is gathered
    const u32 *extra_point = nullptr;                   // an XYZT point added to the sum before the identity test ...
    hipEvent_t extra_ready = nullptr;                   // ... valid once this has fired
};
hipError_t msm_launch(hipStream_t st, const u32 *d_scalars, const u32 *d_niels, const u32 *d_term_point, u64 n_points, u64 n, const msm_plan &p,
                      const msm_ws &ws, u32 *d_result, bool want_encoding, const msm_deps &deps, msm_mark_fn mark, void *mark_user, bool count_terms = false);
// Sum of scalars[j] * P_j over the static generators B, H, G_vec[nm_cap], H_vec[nm_cap] from 32 x 256 fixed-base tables laid out
// for capacity table_cap: partial sums per workgroup, then one wave.  partials: ceil(32 (2 + 2 nm_cap) / 256) XYZT records; out: one.
void launch_static_eval(hipStream_t st, const u32 *gen_tab, const u32 *scalars, u32 nm_cap, u32 table_cap, u32 *partials, u32 *out);
static constexpr u32 GEN_TABLE_CAP_MAX = 1024;    // generators per side up to which the context keeps tables (1 MiB per generator)

// One launch for the fills and copies that start a verification: job j writes n_words 32-bit words to dst,
// copied from src or, when src is null, set to value; the last job slot carries the batch-factor seed.
struct setup_jobs {
    static constexpr int MAX = 16;
    u32 *dst[MAX]; const u32 *src[MAX]; u64 n_words[MAX]; u32 value[MAX];
    int n = 0;
    u32 seed[8]; u32 *seed_dst = nullptr;
    void fill(void *d, u32 v, u64 words) { dst[n] = (u32 *)d; src[n] = nullptr; n_words[n] = words; value[n] = v; n++; }
    void copy(void *d, const void *s, u64 words) { dst[n] = (u32 *)d; src[n] = (const u32 *)s; n_words[n] = words; value[n] = 0; n++; }
};
void launch_setup(hipStream_t st, const setup_jobs &jobs);

// points
void launch_decode_xyzt(hipStream_t st, const u32 *in, u64 n, u32 *out, u8 *ok);
void launch_decode_niels(hipStream_t st, const u32 *in, u64 n, u32 *out, u8 *ok);
void launch_encode_xyzt(hipStream_t st, const u32 *in, u64 n, u32 *out);
void launch_reduce_scalars(hipStream_t st, u32 *s, u64 n);
void launch_count_bad(hipStream_t st, const u8 *ok, u64 n, u32 *out);
void launch_base_points(hipStream_t st, u32 *out);
void launch_bp_generators(hipStream_t st, u32 n_parties, u32 cap, u32 *static_pts);
void launch_build_fixed_table(hipStream_t st, const u32 *base_niels, u32 n_bases, u32 n_windows, u32 *tab);
// list: the entries to decode, or null for the bulk pass (everything but the ledger scan's inputs and, unless bulk_takes_keys, the public keys)
// wide = false: the grid stays at 512 workgroups (a part of a batch verified in parts, whose kernels share the chip with the part
// beside it); true: up to 2048
void launch_decode_list(hipStream_t st, const u8 *bytes, const dev_dec *decs, const u32 *list, u32 n, u32 pt_base, u32 *pts, u8 *ok, bool background = false, bool wide = true);
void launch_point_errors(hipStream_t st, const dev_tx *txs, const dev_cm *cms, const dev_xfer *xfers, u32 n_txs, const u8 *ok, u32 *tx_err, u32 *rp_err);
// first_rp[1] receives *noncanonical_first (the upload's range check of the scalars, k_expand_descriptors)
void launch_verdict(hipStream_t st, const u32 *tx_err, const u32 *rp_err, u32 n_txs, unsigned long long *first_err, u32 *first_rp, const u32 *noncanonical_first);
// transcript, hashes, signatures
// tmpl/segs: transcript templates and per-transaction segment tables; list/tiles: the transactions with a template, sorted by
// template and padded to whole tiles; the stream buffer (plan.stream_words 8-byte words) is reused by the passes;
// serial_list: transactions without a template
void launch_transcript(hipStream_t st, const u8 *bytes, const dev_tx *txs, const dev_cm *cms, const dev_xfer *xfers, const u32 *tmpl, const u32 *segs,
                       const transcript_plan &plan, const u32 *list, const transcript_tile *tiles, u64 *stream, const u32 *serial_list, u32 n_serial, u32 *chal,
                       u32 *tx_err, u32 *rp_err);
void launch_blake3_preimages(hipStream_t st, const u8 *bytes, const dev_tx *txs, const u32 *hash_tx, u32 n, u8 *out);
// signatures: sponge state after the whole blocks of pk || msg (25 x n words) | R = s*H - e*P as XYZT records | hash tail, compare
// per distinct signing key the table of multiples the point kernel adds from (SIG_KEY_TABLE_WORDS words each)
void launch_sig_key_tables(hipStream_t st, const u32 *sig_keys, u32 n_keys, const u32 *pts, u32 *key_tab);
// key_tab == null: every signature builds the table of its key's multiples itself (from pts)
void launch_sig_point_absorb(hipStream_t st, const u8 *bytes, const dev_sig *sigs, u32 n, const u32 *key_tab, const u32 *pts, const u32 *htab, u32 *r_out, u64 *state);
void launch_sig_final(hipStream_t st, const u8 *bytes, const dev_sig *sigs, u32 n, const u64 *state, const u32 *r_in, const u8 *ok, const u8 *ms_hash, u32 *tx_err);
// ledger scan
static constexpr u32 SCAN_K = 4;          // operations per thread
static constexpr u32 SCAN_BLOCK = 128;    // threads per block
void launch_ledger_scan(hipStream_t st, const dev_op *ops, u32 n_ops, const u32 *pts_in, const u32 *gtab, u32 *thread_excl, u32 *block_tot, const u32 *slot_last_op,
                        u32 *slot_base, u32 *slot_final, u32 *emit_incl, const u32 *emit_slot, u32 n_emit, u32 emit_pt_base, u32 *pts, u8 *ok);
// the slots' resulting ciphertexts as 64-byte encodings (read back by the caller, not by any kernel)
void launch_ledger_final(hipStream_t st, const u32 *slot_final, const u32 *slot_base, u32 n_slots, u32 *final_cts);
void launch_ledger_carry(hipStream_t st, const u32 *final_cts, const u32 *slot_of_used, u32 n_slots, u32 *ledger_cts);
// descriptor expansion (xhe_k_expand.hip): see expand_args in xhe_verify_types.h
void launch_expand(hipStream_t st, const expand_args &a);
// proof scalars
static constexpr u32 RPC_SCALARS = 69;    // Montgomery-limb scalars in a transaction's range-proof context
// positions in that context (the products they enter: see the table above rp_prep_terms in xhe_k_scalars.hip)
static constexpr u32 RPC_NEG_RHO_Z = 0, RPC_RHO_Z = 1, RPC_COEF_G = 2, RPC_COEF_H2 = 3, RPC_COEF_H3 = 4, RPC_U2 = 5 /* 16 */, RPC_YU = 21 /* 16 */,
                     RPC_Y2 = 37 /* 16 */, RPC_E = 53 /* 16 */;
static_assert(RPC_E + 16 == RPC_SCALARS, "context layout");
// sigma-proof terms, range-proof terms and the per-transaction range-proof context in one launch
struct proof_scalar_args {
    const u8 *bytes; const dev_tx *txs; const dev_cm *cms; const dev_xfer *xfers; u32 n_cm, n_xf, n_txs;
    const u32 *chal, *seed; u32 *scalars, *term_point, *rp_ctx, *gh, *rp_err;
    u32 weight_bits;       // width of the sigma equations' random factors: sigma_weight_bits(window width of the check they enter)
};
// A random factor fills a whole number of the bucket sort's windows, less the top bit: its digits are then spread evenly over
// the buckets of every window it reaches and the signed recoding never carries out of its top window (a narrower top window
// would put a third of all terms into a few buckets of one window, a carry half of them into one bucket).  At least 128 bits.
static inline u32 sigma_weight_bits(u32 c) { return c * ((129 + c - 1) / c) - 1; }
void launch_proof_scalars(hipStream_t st, const proof_scalar_args &a);
// Terms that share a point: pairs (kept term, absorbed term), and runs of terms (run_off[n_runs + 1] into run_terms) whose
// first term takes the run's sum; absorbed scalars become zero
void launch_merge_terms(hipStream_t st, const u32 *pairs, u32 n_pairs, const u32 *run_off, u32 n_runs, const u32 *run_terms, u32 *scalars);
void launch_sc_sum_partial(hipStream_t st, u32 blocks, const u32 *in, u32 n, u32 classes, u32 *out);
// A workgroup takes rows_per_thread transactions of an lg-group, rp_matrix_slots(lg) of them side by side, and writes one row of
// partial column sums per slot: part[ceil(n_rows / rows_per_thread) * slots][2 * 2^lg], summed by launch_rp_colsum
XHE_HD u32 rp_matrix_slots(u32 lg) { return lg >= 8 ? 1u : 256u >> lg; }
static inline u32 rp_rows_per_thread(u64 n_rows, u32 lg) { u64 r = n_rows / 512; u32 s = rp_matrix_slots(lg); return r < s ? s : r > 64 ? 64u : (u32)r; }
static inline u64 rp_matrix_part_rows(u64 n_rows, u32 lg) { u32 rpt = rp_rows_per_thread(n_rows, lg); return ((n_rows + rpt - 1) / rpt) * rp_matrix_slots(lg); }
void launch_rp_matrix(hipStream_t st, const u32 *group_txs, u32 n_rows, u32 rows_per_thread, u32 lg, const u32 *rp_ctx, const u32 *rp_err, u32 *part);
void launch_rp_colsum(hipStream_t st, const u32 *mat, u32 n_rows, u32 lg, u32 nm_cap, u32 *acc);
// The same column sums as an integer matrix product on the matrix cores (xhe_k_rpgemm.hip), for groups with enough transactions
// to fill it.  A group is processed in chunks of chunk_rows transactions: dig holds one chunk's digit tables, bins the partial
// diagonal sums of its slices, rho_part RP_GEMM_RHO_PARTS scalars.
static constexpr u32 RP_GEMM_RHO_PARTS = 64;
static constexpr u32 RP_GEMM_MAX_KB_PER_SLICE = 64;     // 2 048 transactions per slice: |anti-diagonal sum| <= 32 * 2048 * 2^14 < 2^31
struct rp_gemm_plan { u32 l, h, n_lo, n_hi, per, chunk_rows, kb_per_slice, max_slices; u64 dig_bytes, bins_bytes; };
static inline rp_gemm_plan rp_gemm_plan_for(u64 n_rows, u32 lg) {
    rp_gemm_plan p;
    p.l = (lg + 1) / 2; p.h = lg - p.l; p.n_lo = 1u << p.l; p.n_hi = 1u << p.h; p.per = p.n_lo + p.n_hi;
    const u64 bytes_per_tx = 3ull * p.per * 32;
    u64 chunk = std::min<u64>(32768, ((256ull << 20) / bytes_per_tx) & ~63ull);
    chunk = std::min<u64>(chunk, (n_rows + 63) & ~63ull);
    p.chunk_rows = (u32)std::max<u64>(chunk, 64);
    const u32 kb = p.chunk_rows / 32, tiles = (p.n_hi / 8) * (p.n_lo / 8);
    // enough workgroups to fill the device, and a bounded slice (the diagonal sums are 32-bit)
    u32 slices = std::max<u32>((kb + RP_GEMM_MAX_KB_PER_SLICE - 1) / RP_GEMM_MAX_KB_PER_SLICE, std::min<u32>(kb, (1024 + 3 * tiles - 1) / (3 * tiles)));
    p.kb_per_slice = (kb + slices - 1) / slices;
    p.max_slices = (kb + p.kb_per_slice - 1) / p.kb_per_slice;
    p.dig_bytes = bytes_per_tx * p.chunk_rows;
    p.bins_bytes = 3ull * p.max_slices * (256ull << lg);
    return p;
}
void launch_rp_gemm(hipStream_t st, const u32 *group_txs, u32 n_rows, u32 lg, const u32 *rp_ctx, const u32 *rp_err, u8 *dig, int *bins, u32 *rho_part, u32 nm_cap, u32 *acc);
void launch_static_terms(hipStream_t st, const u32 *gh_sums, u32 n_parts, const u32 *acc, u32 nm_cap, u32 static_term_off, u32 *scalars, u32 *term_point);


// ---- batched range-proof generation (xhe_k_prover.hip): inputs of the verification path, not part of it ----
struct rp_dims { u32 n_txs, m, nm, lg, table_cap; };      // proofs in the chunk, values per proof, 64 m, log2(64 m), generators per side in gen_tab
static constexpr u32 RP_TSTATE_WORDS = 52;                // transcript: 25 x u64 state, then pos | pos_begin << 8, padding
// Montgomery-limb scalars kept per proof between the stages (9 words each)
enum : u32 { RPS_Y = 0, RPS_Z, RPS_YINV, RPS_X, RPS_W, RPS_T0, RPS_T1, RPS_T2, RPS_CL, RPS_CR, RPS_U = 10, RPS_UINV = 26, RPS_COUNT = 42 };
struct rp_buffers {
    const u32 *seeds;        // n x 8: seed of the proof's random generator
    const u64 *ctr0;         // n: its counter at the start of the proof
    const u64 *values;       // n x m
    const u32 *openings;     // n x m x 8: canonical scalars
    u32 *tstate;             // n x RP_TSTATE_WORDS: transcript after the value commitments, updated stage by stage
    u32 *sc_store;           // n x RPS_COUNT x 9
    u32 *vec_a, *vec_b, *vec_yinv;   // n x nm x 9
    u32 *part_a, *part_s;    // n x range_proof_as_segments(nm) XYZT records
    u32 *part_lr;            // n x 2 x range_proof_lr_slots(nm) XYZT records
    u32 *proofs;             // n x 8 (9 + 2 lg) words
};
u32 range_proof_lr_slots(u32 nm);
u32 range_proof_as_segments(u32 nm);
void launch_range_proofs(hipStream_t st, const rp_dims &d, const u32 *gen_tab, const rp_buffers &b);

}  // namespace xhe
