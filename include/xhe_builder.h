/* xhe_builder.h -- C ABI of the CPU-side transaction builder (prover) and synthetic-batch generator.
 *
 * These entry points create inputs for xhe_verify_batch (include/xhe.h): keys, Twisted-ElGamal
 * ciphertexts, transactions with their sigma proofs, aggregated range proof and signatures, and
 * ledgers.  They replace, for a C caller, the reference's TransactionBuilder::{build_unsigned, build}
 * (src/tx/builder.rs:320-554), TransactionUnsigned::{hash, set_multisig, sign} (:190-218),
 * ElGamalKeypair::{keygen_with_secret, sign} (src/elgamal.rs:194-218), ElGamalPubkey::encrypt (:109-125),
 * Transaction::apply_without_verify (src/tx/verify.rs:545-619) and the batch construction of
 * benches/tx.rs:129-189.  They run on the CPU and are not part of the verification path.
 */
#ifndef XHE_BUILDER_H
#define XHE_BUILDER_H
#include <stdint.h>
#include "xhe.h"

#ifdef __cplusplus
extern "C" {
#endif

/* TransactionBuilder (src/tx/builder.rs:77-84) plus the balances its GetBlockchainAccountBalance state
 * would return for every used asset. */
typedef struct xhe_tx_request {
    uint8_t version;
    uint8_t type;                       /* XHE_TX_*                                                  */
    uint32_t n_transfers;               /* Transfers                                                 */
    const uint8_t *transfer_assets;     /* n_transfers x 32                                          */
    const uint64_t *transfer_amounts;   /* n_transfers                                               */
    const uint8_t *transfer_dests;      /* n_transfers x 32                                          */
    const uint8_t *const *transfer_extra;   /* nullable: per transfer extra-data bytes               */
    const uint32_t *transfer_extra_len;     /* nullable: per transfer length, 0 = None               */
    uint8_t burn_asset[32];             /* Burn                                                      */
    uint64_t burn_amount;
    uint32_t n_signers;                 /* MultiSig                                                  */
    const uint8_t *signers;             /* n_signers x 32                                            */
    uint8_t threshold;
    uint8_t contract[32];               /* CallContract                                              */
    uint32_t n_contract_assets;
    const uint8_t *contract_assets;     /* n_contract_assets x 32                                    */
    const uint64_t *contract_amounts;
    const uint8_t *opaque;              /* CallContract params bytes / DeployContract code           */
    uint32_t opaque_len;
    uint64_t fee, nonce;
    uint32_t n_balances;                /* state: one entry per asset the transaction uses           */
    const uint8_t *balance_assets;      /* n_balances x 32                                           */
    const uint64_t *balance_amounts;    /* plaintext balances                                        */
    const uint8_t *balance_cts;         /* n_balances x 64 current ciphertexts                       */
} xhe_tx_request;

int32_t xhe_keygen(const uint8_t secret32[32], uint8_t pub32[32]);
int32_t xhe_keygen_from_seed(const uint8_t seed32[32], uint8_t secret32[32], uint8_t pub32[32]);
int32_t xhe_encrypt(const uint8_t pub32[32], uint64_t amount, const uint8_t seed32[32], uint8_t ct64[64]);
int32_t xhe_ct_decrypts_to(const uint8_t secret32[32], const uint8_t ct64[64], uint64_t amount);
/* ElGamalSecretKey::decrypt (src/elgamal.rs:140-145): out32 = the encoding of amount*G = C - s*D; -1 when a point does not decode */
int32_t xhe_decrypt_to_point(const uint8_t secret32[32], const uint8_t ct64[64], uint8_t out32[32]);
int32_t xhe_sign(const uint8_t secret32[32], const uint8_t *msg, uint64_t n, const uint8_t seed32[32], uint8_t sig64[64]);
int32_t xhe_signature_verify_host(const uint8_t sig64[64], const uint8_t *msg, uint64_t n, const uint8_t pub32[32]);

/* returns 0, -1 bad argument, -2 output buffer too small (out_len holds the needed size),
 * 101 decompression, 102 insufficient funds, 103 format, 104 missing balance */
int32_t xhe_tx_build_unsigned(const xhe_tx_request *rq, const uint8_t source_secret[32], const uint8_t seed32[32], uint8_t *out,
                              uint64_t cap, uint64_t *out_len, uint8_t *new_balances, uint32_t *n_new_balances);
int32_t xhe_tx_hash(const uint8_t *record, uint64_t len, uint8_t hash32[32]);
int32_t xhe_tx_finalize(const uint8_t *unsigned_record, uint64_t len, int32_t multisig_present, uint32_t n_multisig, const uint8_t *ids,
                        const uint8_t *sigs, const uint8_t source_secret[32], const uint8_t seed32[32], uint8_t *out, uint64_t cap,
                        uint64_t *out_len);
int32_t xhe_tx_apply_host(const uint8_t *record, uint64_t len, uint64_t n_accounts, const uint8_t *account_keys, uint64_t n_balances,
                          const uint32_t *balance_account, const uint8_t *balance_assets, uint8_t *balance_cts);

/* GPU-assisted proving: the aggregated 64-bit range proofs of n transactions at once (RangeProof::prove_multiple as
 * TransactionBuilder::build_unsigned calls it, src/tx/builder.rs:525-533; bulletproofs 5.2.0 of the xelis fork), every
 * proof over m values (a power of two, at most XHE_RANGE_PROOF_GPU_MAX_VALUES).  Per proof: the Merlin transcript after
 * "rangeproof v1", n, m and the value commitments were appended (200 bytes of STROBE state, pos, pos_begin, 6 bytes of
 * padding), the m values and their m openings (canonical scalars), and the seed and counter of the random generator the
 * CPU prover would draw from (SHAKE256(seed || LE64(counter)) -> wide-reduced scalar per draw; draws in order:
 * a_blinding, s_blinding, s_L[i] and s_R[i] interleaved, t1_blinding, t2_blinding).  proofs receives 32 (9 + 2 lg) bytes
 * per proof, lg = log2(64 m): the same bytes the CPU prover writes for the same inputs.  Creates inputs for
 * xhe_verify_batch; not part of the verification path. */
#define XHE_RANGE_PROOF_GPU_MAX_VALUES 16
#define XHE_RANGE_PROOF_TRANSCRIPT_BYTES 208
int32_t xhe_range_proofs_prove(xhe_ctx *ctx, uint64_t n, uint32_t m, const uint8_t *transcripts, const uint64_t *values, const uint8_t *openings32,
                               const uint8_t *seeds32, const uint64_t *counters, uint8_t *proofs);

typedef struct xhe_synth xhe_synth;
xhe_synth *xhe_synth_transfers(uint64_t n_txs, uint32_t n_transfers, uint32_t n_senders, const uint8_t seed32[32], int32_t threads);
/* the same batch, byte for byte, with the range proofs generated on ctx's GPU (ctx == NULL, or more than
 * XHE_RANGE_PROOF_GPU_MAX_VALUES value commitments per transaction: all on the CPU) */
xhe_synth *xhe_synth_transfers_gpu(xhe_ctx *ctx, uint64_t n_txs, uint32_t n_transfers, uint32_t n_senders, const uint8_t seed32[32], int32_t threads);
/* n_groups x 16 mixed transactions over disjoint account groups: 2 multisig set-ups, 4 one-asset transfers, 4 two-asset
 * double transfers, 2 native + 2 second-asset burns, 2 co-signed transfers from the accounts set up in the same group. */
xhe_synth *xhe_synth_mixed(uint64_t n_groups, const uint8_t seed32[32], int32_t threads);
/* the same batch, byte for byte, with the range proofs generated on ctx's GPU, one xhe_range_proofs_prove call per shape
 * (1, 2 and 4 value commitments; ctx == NULL: all on the CPU) */
xhe_synth *xhe_synth_mixed_gpu(xhe_ctx *ctx, uint64_t n_groups, const uint8_t seed32[32], int32_t threads);
void xhe_synth_free(xhe_synth *s);
const uint8_t *xhe_synth_records(const xhe_synth *s, uint64_t *len);
const uint64_t *xhe_synth_offsets(const xhe_synth *s, uint64_t *n_txs);
const xhe_ledger_view *xhe_synth_ledger(const xhe_synth *s);
const uint8_t *xhe_synth_secrets(const xhe_synth *s);   /* per sender pair: sender secret || receiver secret */

#ifdef __cplusplus
}
#endif
#endif /* XHE_BUILDER_H */
