/* xhe.h -- C ABI of the MI355X batch-verification engine for XELIS-HE confidential transactions.
 *
 * The reference (xelis-project/xelis-he, a Rust crate) has no FFI layer; the seams this ABI replaces
 * are the Rust calls named beside each entry point.  Every entry point takes plain pointers and sizes;
 * all multi-byte integers in buffers are little-endian unless stated; all buffers are caller-owned
 * host memory unless the name says "_dev".  Return value: 0 = the call ran (the verdict is in the out
 * struct), < 0 = usage or runtime error (XHE_E_*).  A context is bound to one GPU (it owns four HIP streams
 * for kernels and one for the uploads' copies there, all at the default priority: with the default GPU_MAX_HW_QUEUES = 4 that is
 * every hardware queue of the process, so a caller's own stream shares a queue with one of them) and must not be used from two threads at once, with one exception:
 * xhe_batch_upload of one batch may run beside xhe_batch_verify / xhe_batch_apply / xhe_batch_read_ledger of another (that is what
 * xhe_verify_batches does); different contexts are independent.
 */
#ifndef XHE_H
#define XHE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xhe_ctx xhe_ctx;

/* call-level errors */
#define XHE_E_ARG (-1)        /* bad argument (null pointer, size out of range)              */
#define XHE_E_HIP (-2)        /* HIP runtime error, see xhe_last_error()                     */
#define XHE_E_NO_DEVICE (-3)  /* no usable gfx950 device                                     */
#define XHE_E_DECODE (-4)     /* level-0 only: a compressed point of the input does not decode */

/* verdict error codes: the variants of the reference's VerificationError / ProofVerificationError
 * (src/tx/verify.rs:16-21, src/lib.rs:71-89) plus one for records the reference's serde layer would
 * already have refused. */
#define XHE_OK 0
#define XHE_ERR_STATE 1
#define XHE_ERR_INVALID_NONCE 2
#define XHE_ERR_SIGNATURE 3
#define XHE_ERR_DECOMPRESSION 4
#define XHE_ERR_COMMITMENT_EQ 5
#define XHE_ERR_CT_VALIDITY 6
#define XHE_ERR_GENERIC_PROOF 7
#define XHE_ERR_RANGE_PROOF 8
#define XHE_ERR_TRANSCRIPT_IDENTITY 9
#define XHE_ERR_FORMAT 10
#define XHE_ERR_MALFORMED 11

/* transaction types (src/tx/mod.rs:85-93) */
#define XHE_TX_TRANSFERS 0
#define XHE_TX_BURN 1
#define XHE_TX_MULTISIG 2
#define XHE_TX_CALL_CONTRACT 3
#define XHE_TX_DEPLOY_CONTRACT 4

/* One GPU per context.  device_id is the HIP device ordinal.  Returns NULL on failure
 * (xhe_last_error(NULL) then describes it). */
xhe_ctx *xhe_create(int device_id);
void xhe_destroy(xhe_ctx *ctx);
const char *xhe_last_error(const xhe_ctx *ctx);

/* ---- level 0: the operator seams --------------------------------------------------------------- */

/* RistrettoPoint::vartime_multiscalar_mul (src/proofs.rs:50-60): out32 = encoding of sum scalars[i] *
 * points[i]; scalars are 32-byte little-endian integers taken modulo the group order; points are
 * RFC 9496 encodings.  is_identity receives RistrettoPoint::is_identity() of the sum (src/proofs.rs:62).
 * n may be 0 (result: identity).  Returns XHE_E_DECODE if a point does not decode. */
int32_t xhe_msm_vartime(xhe_ctx *ctx, uint64_t n, const uint8_t *scalars32, const uint8_t *points32,
                        uint8_t out32[32], uint8_t *is_identity);

/* xhe_msm_vartime for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc):
 * same arguments and results, but the staging buffers and the whole workspace are ranges of one file-scope device array of
 * 12 MiB that every GPU the library runs on holds until the process ends; the call makes no hipMalloc / hipFree.  It is
 * not capturable into a HIP graph: it copies from and to the caller's host buffers and waits for its stream.  No counterpart in the reference.  n above XHE_MSM_STATIC_MAX_TERMS (or a workspace that does not
 * fit the array) is XHE_E_ARG with a message in xhe_last_error().  All calls on one GPU share that GPU's array, contexts
 * included: calls that overlap in time on one GPU race, so serialise them. */
#define XHE_MSM_STATIC_MAX_TERMS 4096
int32_t xhe_msm_vartime_static(xhe_ctx *ctx, uint64_t n, const uint8_t *scalars32, const uint8_t *points32,
                               uint8_t out32[32], uint8_t *is_identity);

/* CompressedRistretto::decompress over a batch (src/compressed.rs:28-33,57-62,78-83,100-105):
 * out_xyzt128 receives canonical X,Y,Z,T (4 x 32 bytes) per point, ok[i] = 1 when the encoding is valid
 * (the output of an invalid one is the identity). */
int32_t xhe_ristretto_decompress_batch(xhe_ctx *ctx, uint64_t n, const uint8_t *in32, uint8_t *out_xyzt128,
                                       uint8_t *ok);
/* RistrettoPoint::compress over a batch (src/compressed.rs:18-20,44-49,72-74,94-96). */
int32_t xhe_ristretto_compress_batch(xhe_ctx *ctx, uint64_t n, const uint8_t *in_xyzt128, uint8_t *out32);

/* ---- level 1: Transaction::verify_batch (src/tx/verify.rs:487-517) --------------------------------
 *
 * Transaction record (the reference has no wire format: Transaction::to_bytes, verify.rs:623-688, is
 * only a signing pre-image).  One record =
 *   header, 16 bytes: u8 type, u8 flags (bit0: the multisig field is Some), u16 n_items (Transfers:
 *     transfers; MultiSig: signers; CallContract: assets; else 0), u16 n_commitments, u16 n_multisig,
 *     u32 opaque_len (DeployContract: code bytes; CallContract: params bytes; else 0), u16 rp_len
 *     (range-proof bytes), u16 zero;
 *   Transfers only: n_items x u32 extra_len (byte length of ExtraData::to_bytes, 0 = None);
 *   the signing pre-image exactly as Transaction::to_bytes lays it out, without multisig entries:
 *     version u8, source[32], fee BE u64, nonce BE u64, type payload, range proof, then per commitment
 *     asset[32], commitment[32], CommitmentEqProof[192];
 *   n_multisig x (u8 signer index, signature[64]);
 *   signature[64] = s || e.
 * A batch is the concatenation of records plus offsets[n+1] (offsets[i] = first byte of record i,
 * offsets[n] = total length).
 */
typedef struct xhe_ledger_view {
    uint64_t n_accounts;
    const uint8_t *account_keys;        /* n_accounts x 32: compressed public keys                  */
    const uint64_t *account_nonces;     /* n_accounts                                               */
    uint64_t n_balances;
    const uint32_t *balance_account;    /* n_balances: index into accounts                          */
    const uint8_t *balance_assets;      /* n_balances x 32                                          */
    const uint8_t *balance_cts;         /* n_balances x 64: commitment || handle                    */
    uint64_t n_multisig;
    const uint32_t *multisig_account;   /* n_multisig: index into accounts                          */
    const uint8_t *multisig_threshold;  /* n_multisig                                               */
    const uint32_t *multisig_offsets;   /* n_multisig + 1: ranges into multisig_signers             */
    const uint8_t *multisig_signers;    /* 32 bytes each                                            */
} xhe_ledger_view;

typedef struct xhe_ledger_out {
    uint8_t *balance_cts;      /* n_balances x 64, may alias the input; written only on accept       */
    uint64_t *account_nonces;  /* n_accounts (the reference stores the nonce back unchanged)         */
    int64_t *multisig_set_by;  /* n_accounts: index of the last transaction of the batch that set the
                                  account's multisig configuration (take signers/threshold from that
                                  record), -1 if none                                                */
} xhe_ledger_out;

typedef struct xhe_verdict {
    int32_t accept;        /* 1 = Ok(())                                                            */
    int32_t err_code;      /* XHE_OK or XHE_ERR_*                                                   */
    int64_t first_bad_tx;  /* index of the failing transaction, -1 when only the aggregate check failed */
} xhe_verdict;

/* Verifies the batch against the ledger snapshot `in` (BlockchainVerificationState of verify.rs:25-77,
 * flattened).  On accept `out` (may be NULL) receives the ledger after all transactions were applied in
 * order.  Two deliberate differences from the reference: on reject `out` is not written
 * (the reference leaves its state half-updated), and an unknown account or asset is XHE_ERR_STATE (the
 * reference's mock ledger panics).  rng_seed seeds the batch-factor generator (the reference draws
 * them from the OS RNG, src/proofs.rs:181,326); the verdict does not depend on it.
 * A batch of 64 MiB of records or 32 000 transactions and more is cut into consecutive parts inside the call (the host's parsing of part k + 1 runs
 * beside the device's verification of part k, every part against the ledger the part before it left); verdict, index and
 * ledger are those of the batch in one piece.  XHE_SPLIT_PARTS in the environment sets the number of parts (1: never cut). */
int32_t xhe_verify_batch(xhe_ctx *ctx, const uint8_t *records, const uint64_t *offsets, uint64_t n_txs,
                         const xhe_ledger_view *in, const xhe_ledger_out *out, xhe_verdict *verdict,
                         const uint8_t rng_seed[32]);

/* xhe_verify_batch over several independent batches (each against its own ledger snapshot, e.g. the blocks a node is catching up on),
 * pipelined inside the library: a helper thread parses and uploads the batches ahead while batch k is verified, and batch k + 1 is
 * enqueued on the device behind batch k before batch k's verdict is read, so the host work, the host-to-device copy and the
 * latency-bound start of one batch hide behind the GPU work of the one before.  verdicts[k] and batches[k].out are what
 * xhe_verify_batch would give for batch k alone.  Returns the first call-level error: the batches whose verification had not
 * finished by then read {accept 0, XHE_ERR_STATE, -1} (never accept), and xhe_last_error() names the batch the call stopped at. */
typedef struct xhe_batch_args {
    const uint8_t *records; const uint64_t *offsets; uint64_t n_txs;
    const xhe_ledger_view *in; const xhe_ledger_out *out;   /* out may be NULL */
    const uint8_t *rng_seed;                                /* 32 bytes */
} xhe_batch_args;
int32_t xhe_verify_batches(xhe_ctx *ctx, uint64_t n_batches, const xhe_batch_args *batches, xhe_verdict *verdicts);

/* ---- several GPUs: the replica model of the reference's bench (benches/tx.rs:252-343), applied to ONE batch ------------------
 *
 * xhe_verify_batch_sharded is xhe_verify_batch over n_ctx contexts (one per GPU; the same device may appear twice): the batch is
 * cut along the connected components of its account graph (a transaction depends on every earlier one that touches its accounts,
 * verify.rs:301-341; into consecutive ranges when the components do not balance, see xhe_plan_shards), every shard is verified by its
 * own context, on a thread of its own, against the accounts and balances of the ledger snapshot that its transactions name, and the
 * verdicts combine by the reference's precedence (first failing transaction in batch order, verify.rs:492-498, then GenericProof,
 * then RangeProof, verify.rs:500-514).  verdict, first_bad_tx (an index into the whole batch) and `out` are those of the
 * single-context call.  xhe_partition_batch exposes the cut (shard_of_tx[i] in [0, n_shards); returns 1 and the index when a record
 * is malformed), xhe_verdict_key / xhe_verdict_from_key the 64-bit key whose MIN over the shards is the batch's verdict: a
 * process-per-GPU driver all-reduces that one word (RCCL, MIN) instead of calling xhe_verify_batch_sharded. */
int32_t xhe_partition_batch(const uint8_t *records, const uint64_t *offsets, uint64_t n_txs, uint32_t n_shards, uint32_t *shard_of_tx,
                            int64_t *first_malformed);
/* The cut xhe_verify_batch_sharded itself makes.  When the account groups do not balance -- the heaviest shard would carry more than
 * 1.25 of the even share, e.g. the reference bench's single sender for every transaction (benches/tx.rs:129-189) -- and no
 * transaction of the batch changes a multisig configuration, the batch is cut into n_shards consecutive ranges of equal weight
 * instead (*consecutive_ranges = 1, shard_of_tx ascends): range s then has to start from the ledger the ranges in front of it leave
 * (xhe_batch_apply range after range; xhe_batch_set_ledger hands the state to an uploaded range).  Otherwise as xhe_partition_batch.
 * xhe_slice_shard copies one shard's records out of the batch (out_offsets: n_local + 1 entries, out_global_index: n_local;
 * with all three output pointers NULL only the sizes are returned). */
int32_t xhe_plan_shards(const uint8_t *records, const uint64_t *offsets, uint64_t n_txs, uint32_t n_shards, uint32_t *shard_of_tx,
                        int32_t *consecutive_ranges, int64_t *first_malformed);
int32_t xhe_slice_shard(const uint8_t *records, const uint64_t *offsets, uint64_t n_txs, const uint32_t *shard_of_tx, uint32_t shard,
                        uint8_t *out_records, uint64_t *out_offsets, uint64_t *out_global_index, uint64_t *n_local, uint64_t *n_bytes);
int32_t xhe_verify_batch_sharded(xhe_ctx *const *ctxs, uint32_t n_ctx, const uint8_t *records, const uint64_t *offsets, uint64_t n_txs,
                                 const xhe_ledger_view *in, const xhe_ledger_out *out, xhe_verdict *verdict, const uint8_t rng_seed[32]);
/* global_index[k] = position in the whole batch of the shard's k-th transaction (n_local entries) */
uint64_t xhe_verdict_key(const xhe_verdict *shard_verdict, const uint64_t *global_index, uint64_t n_local);
void xhe_verdict_from_key(uint64_t key, xhe_verdict *verdict);

/* The same verification on a batch already resident in HBM: xhe_batch_upload parses, checks and
 * uploads the records and the ledger once; xhe_batch_verify runs only GPU work plus a 16-byte
 * read-back.  xhe_batch_read_ledger copies the resulting ledger out after an accepting verify. */
typedef struct xhe_batch xhe_batch;
int32_t xhe_batch_upload(xhe_ctx *ctx, const uint8_t *records, const uint64_t *offsets, uint64_t n_txs,
                         const xhe_ledger_view *in, xhe_batch **batch_out, xhe_verdict *structural_verdict);
int32_t xhe_batch_verify(xhe_ctx *ctx, xhe_batch *batch, xhe_verdict *verdict, const uint8_t rng_seed[32]);
int32_t xhe_batch_read_ledger(xhe_ctx *ctx, xhe_batch *batch, const xhe_ledger_out *out);
/* Replaces the ledger ciphertexts the uploaded batch starts from: balance_cts = n_balances x 64 bytes in the order of the ledger view
 * the batch was uploaded with (keys, assets, nonces and multisig configurations stay).  The next xhe_batch_verify / xhe_batch_apply
 * runs against them.  For callers that learn a batch's starting state after they uploaded it: xhe_verify_batch_sharded hands a
 * consecutive range of a batch the ledger the ranges in front of it leave (the state every later transaction of
 * Transaction::verify_batch sees, verify.rs:492-498).  The batch must not be in use by another call. */
int32_t xhe_batch_set_ledger(xhe_ctx *ctx, xhe_batch *batch, const uint8_t *balance_cts);
/* Transaction::apply_without_verify (/root/reference/src/tx/verify.rs:545-619) for every transaction of the uploaded
 * batch in order: only the Twisted-ElGamal ledger arithmetic (sender balance minus fee/burn/deposits and transfer
 * ciphertexts, receiver balance plus transfer ciphertext, elgamal.rs:280-377), no signature, nonce or proof check.
 * On accept xhe_batch_read_ledger returns the resulting ledger.  Where the reference returns the state's error or
 * panics ("ill-formed ciphertext"), the verdict is XHE_ERR_STATE / XHE_ERR_DECOMPRESSION with the transaction's index.
 * multisig_set_by reports MultiSig transactions with a well-formed payload only. */
int32_t xhe_batch_apply(xhe_ctx *ctx, xhe_batch *batch, xhe_verdict *verdict);
void xhe_batch_free(xhe_ctx *ctx, xhe_batch *batch);

/* Diagnostics for parity tests (no counterpart in the reference): internal values of an uploaded batch.
 *   XHE_DEBUG_CHALLENGES  after xhe_batch_verify: the Fiat-Shamir challenges of every transaction in transaction order, as
 *                         ProtocolTranscript::challenge_scalar yields them (src/transcript.rs:46-51): per source commitment
 *                         c, w; per transfer c, w; then the range proof's y, z, x, w, u_1..u_lg.  32 bytes each.
 *   XHE_DEBUG_GENERATORS  B, B_blinding (= H), then the first k generators G_i and the first k H_i of BP_GENS
 *                         (src/proofs.rs:19-22) as the kernels hold them, re-encoded: 32 bytes each, k = (cap / 32 - 2) / 2
 *                         capped at the batch's generator count.
 *   XHE_DEBUG_STATIC_SCALARS  after xhe_batch_verify: the scalars of B, B_blinding, G_0.., H_0.. in the merged check (sums over
 *                         all range proofs of the batch, each proof weighted with the verifier's random factor), 32 bytes
 *                         each, canonical; 2 + 2 * (the batch's generator count) values.
 * Returns the number of 32-byte values written, or < 0. */
#define XHE_DEBUG_CHALLENGES 1
#define XHE_DEBUG_GENERATORS 2
#define XHE_DEBUG_STATIC_SCALARS 3
int64_t xhe_batch_debug_read(xhe_ctx *ctx, xhe_batch *batch, int32_t what, uint8_t *out, uint64_t cap_bytes);

/* Per-stage GPU time of the last xhe_batch_verify / xhe_msm_vartime on this context, measured with HIP
 * events on the stream each stage runs on (stages on different streams overlap; "total" spans the main
 * stream).  names/ms arrays of length cap; returns the number of stages. */
int32_t xhe_last_stage_times(const xhe_ctx *ctx, const char **names, float *ms, int32_t cap);
/* How much the context measures: 2 (default) every stage, 1 only "before_accumulate", "msm_accumulate" (the bucket
 * accumulation kernel), "after_accumulate" and "total", 0 nothing.  Every measured stage boundary is an event on
 * its stream; level 0 or 1 keeps them out of the launch sequence.  No counterpart in the reference (diagnostics). */
int32_t xhe_set_stage_timing(xhe_ctx *ctx, int32_t level);

/* Counters of the multiscalar check, summed over the verifications that finished on this context since the last reset (every part of
 * a batch that xhe_verify_batch cut into parts counts as one verification): launches of the bucket accumulation kernel, its GPU time
 * from a pair of HIP events around it on the stream it ran on, the sorted entries it went through (non-zero digits over all windows)
 * and the terms with a non-zero scalar that entered the bucket sort (after terms that share a point were merged).  Collected only
 * while enabled (two event records per verification).  No counterpart in the reference (measurement). */
typedef struct xhe_msm_stats { uint64_t launches; uint64_t entries; uint64_t terms_nonzero; double accumulate_ms; } xhe_msm_stats;
int32_t xhe_msm_stats_enable(xhe_ctx *ctx, int32_t on);
int32_t xhe_msm_stats_read(xhe_ctx *ctx, xhe_msm_stats *out, int32_t reset);

/* The library's XHE_* environment switches (diagnostics, A/B measurements, tests; none is needed in
 * production) are read once, when the first context of the process is created.  This call reads them again: for tests and
 * measurement tools that change a switch inside one process.  Not to be called while another thread is inside the library.
 * No counterpart in the reference. */
void xhe_settings_reload(void);

/* ---- Balance decoding (ECDLPInstance::decode, src/elgamal.rs:68-92) ----------------------------------------
 * The reference decodes a decrypted point m*G into the integer m with the ecdlp module of its curve25519-dalek fork and that
 * module's pre-computed table files; module and file format are outside the reference's sources, so the tables here are this
 * library's own (built on the GPU, never written to disk) and only the contract is the reference's: the integer, or none.
 *
 * xhe_ecdlp_table_create builds the baby-step table for 2^table_bits steps (8..30; 2^(table_bits+4) bytes of device memory:
 * 256 MiB at 24, 4 GiB at 28, 16 GiB at 30) - the counterpart of loading an ECDLPTablesFileView.  A table belongs to its context
 * and, like it, serves one call at a time.
 *
 * xhe_ecdlp_decode: points32 = n Ristretto encodings (what ECDLPInstance holds, src/elgamal.rs:68).  amounts[i] = m with
 * m*G = point i and 0 <= m < 2^max_bits, or -1 when there is none in that range (the reference's None) or the encoding is
 * invalid (status[i] = 1 then; status may be NULL).  max_bits 1..62 and at most table_bits + 25.  n <= 2^24.
 * xhe_elgamal_decrypt_decode: cts64 = n ciphertexts (commitment || handle) under one secret key; the point is C - s*D
 * (ElGamalSecretKey::decrypt, src/elgamal.rs:140-145), decoded as above.  ECDLPArguments' range start and progress callback
 * have no counterpart: the range starts at 0 and a call is a few milliseconds. */
typedef struct xhe_ecdlp_table xhe_ecdlp_table;
int32_t xhe_ecdlp_table_create(xhe_ctx *ctx, uint32_t table_bits, xhe_ecdlp_table **out);
void xhe_ecdlp_table_free(xhe_ecdlp_table *table);
int32_t xhe_ecdlp_decode(xhe_ecdlp_table *table, const uint8_t *points32, uint64_t n, uint32_t max_bits, int64_t *amounts, uint8_t *status);
int32_t xhe_elgamal_decrypt_decode(xhe_ecdlp_table *table, const uint8_t secret32[32], const uint8_t *cts64, uint64_t n, uint32_t max_bits, int64_t *amounts,
                                   uint8_t *status);
/* The last decode call on this table: instances, giant steps walked (one hash-table look-up each), GPU time of the walk from a
 * pair of HIP events on its stream, lanes used.  No counterpart in the reference (measurement). */
typedef struct xhe_ecdlp_stats { uint64_t instances; uint64_t steps; double walk_ms; uint64_t lanes; } xhe_ecdlp_stats;
int32_t xhe_ecdlp_last_stats(const xhe_ecdlp_table *table, xhe_ecdlp_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* XHE_H */
