// Merlin transcript of every transaction (thread per transaction) and BLAKE3 of multisig pre-images.
#include "xhe_dev_common.h"
#include "xhe_kernels.h"

namespace xhe {

// ---- transcript ------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_zero32_dev(const u8 *p) { u8 a = 0; for (int i = 0; i < 32; i++) a |= p[i]; return a == 0; }

__global__ void __launch_bounds__(64) k_transcript(const u8 *__restrict__ bytes, const dev_tx *__restrict__ txs, const dev_cm *__restrict__ cms,
                                                   const dev_xfer *__restrict__ xfers, u32 n_txs, u32 *__restrict__ chal, u32 *__restrict__ tx_err,
                                                   u32 *__restrict__ rp_err) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_txs) return;
    const dev_tx d = txs[i];
    const u8 *rec = bytes + d.rec;
    transcript t; t.init("transaction-proof");
    t.append_u64("version", d.version);
    t.append_message("source_pubkey", rec + d.pre_off + 1, 32);
    t.append_u64("fee", d.fee);
    t.append_u64("nonce", d.nonce);
    u32 ci = d.chal_off;
    for (u32 k = 0; k < d.n_commitments; k++) {
        const u8 *cm = bytes + cms[d.cm_first + k].off;
        t.append_message("dom-sep", (const u8 *)"new-commitment-proof", 20);
        t.append_message("new_source_commitment_asset", cm, 32);
        t.append_message("new_source_commitment", cm + 32, 32);
        t.append_message("dom-sep", (const u8 *)"equality-proof", 14);
        bool ok = t.validate_and_append_point("Y_0", cm + 64);
        ok = t.validate_and_append_point("Y_1", cm + 96) && ok;
        ok = t.validate_and_append_point("Y_2", cm + 128) && ok;
        if (!ok) raise_err(tx_err, i, errkey(ST_COMMITMENT, k, 2, XHE_ERR_TRANSCRIPT_IDENTITY));
        store_scalar_words(chal + 8ull * ci++, t.challenge_scalar("c"));
        t.append_message("z_s", cm + 160, 32); t.append_message("z_x", cm + 192, 32); t.append_message("z_r", cm + 224, 32);
        store_scalar_words(chal + 8ull * ci++, t.challenge_scalar("w"));
    }
    if (d.type == XHE_TX_TRANSFERS) {
        for (u32 k = 0; k < d.n_transfers; k++) {
            const u8 *x = bytes + xfers[d.xfer_first + k].off, *pr = bytes + xfers[d.xfer_first + k].proof_off;
            t.append_message("dom-sep", (const u8 *)"transfer-proof", 14);
            t.append_message("dest_pubkey", x + 32, 32);
            t.append_message("amount_commitment", x + 64, 32);
            t.append_message("amount_sender_handle", x + 96, 32);
            t.append_message("amount_receiver_handle", x + 128, 32);
            t.append_message("dom-sep", (const u8 *)"validity-proof", 14);
            bool ok = t.validate_and_append_point("Y_0", pr);
            ok = t.validate_and_append_point("Y_1", pr + 32) && ok;
            ok = t.validate_and_append_point("Y_2", pr + 64) && ok;
            if (!ok) raise_err(tx_err, i, errkey(ST_TRANSFER, k, 3, XHE_ERR_TRANSCRIPT_IDENTITY));
            store_scalar_words(chal + 8ull * ci++, t.challenge_scalar("c"));
            t.append_message("z_r", pr + 96, 32); t.append_message("z_x", pr + 128, 32);
            store_scalar_words(chal + 8ull * ci++, t.challenge_scalar("w"));
        }
    } else if (d.type == XHE_TX_BURN) {
        const u8 *p = rec + d.payload_off;
        t.append_message("dom-sep", (const u8 *)"burn-proof", 10);
        t.append_message("asset", p, 32);
        u64 amount = 0;
        for (int b = 0; b < 8; b++) amount = (amount << 8) | p[32 + b];
        t.append_u64("amount", amount);
    } else if (d.type == XHE_TX_MULTISIG) {
        const u8 *p = rec + d.payload_off;
        t.append_message("dom-sep", (const u8 *)"multisig-proof", 14);
        t.append_u64("threshold", p[0]);
        for (u32 k = 0; k < d.n_items; k++) t.append_message("signer", p + 1 + 32 * k, 32);
    }
    // aggregated range proof (dalek-bulletproofs RangeProof::verify_multiple transcript)
    const u8 *rp = rec + d.rp_off;
    bool rp_ok = true;
    t.append_message("dom-sep", (const u8 *)"rangeproof v1", 13);
    t.append_u64("n", 64); t.append_u64("m", d.m);
    for (u32 k = 0; k < d.n_commitments; k++) t.append_point("V", bytes + cms[d.cm_first + k].off + 32);
    for (u32 k = 0; k < d.n_transfers; k++) t.append_point("V", bytes + xfers[d.xfer_first + k].off + 64);
    {
        u8 zero[32];
        for (int b = 0; b < 32; b++) zero[b] = 0;
        for (u32 k = d.n_commitments + d.n_transfers; k < d.m; k++) t.append_point("V", zero);
    }
    rp_ok = t.validate_and_append_point("A", rp) && rp_ok;
    rp_ok = t.validate_and_append_point("S", rp + 32) && rp_ok;
    store_scalar_words(chal + 8ull * ci++, t.challenge_scalar("y"));
    store_scalar_words(chal + 8ull * ci++, t.challenge_scalar("z"));
    rp_ok = t.validate_and_append_point("T_1", rp + 64) && rp_ok;
    rp_ok = t.validate_and_append_point("T_2", rp + 96) && rp_ok;
    store_scalar_words(chal + 8ull * ci++, t.challenge_scalar("x"));
    t.append_message("t_x", rp + 128, 32); t.append_message("t_x_blinding", rp + 160, 32); t.append_message("e_blinding", rp + 192, 32);
    store_scalar_words(chal + 8ull * ci++, t.challenge_scalar("w"));
    t.append_message("dom-sep", (const u8 *)"ipp v1", 6);
    t.append_u64("n", 64ull * d.m);
    for (u32 k = 0; k < d.lg; k++) {
        rp_ok = t.validate_and_append_point("L", rp + 224 + 64 * k) && rp_ok;
        rp_ok = t.validate_and_append_point("R", rp + 224 + 64 * k + 32) && rp_ok;
        store_scalar_words(chal + 8ull * ci++, t.challenge_scalar("u"));
    }
    if (!rp_ok) rp_err[i] = 1;
}


// ---- signatures ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_blake3_preimages(const u8 *__restrict__ bytes, const dev_tx *__restrict__ txs, const u32 *__restrict__ hash_tx, u32 n,
                                                         u8 *__restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dev_tx d = txs[hash_tx[i]];
    u8 h[32];
    blake3_hash(h, bytes + d.rec + d.pre_off, d.pre_len);
    for (int k = 0; k < 32; k++) out[32ull * i + k] = h[k];
}


void launch_transcript(hipStream_t st, const u8 *bytes, const dev_tx *txs, const dev_cm *cms, const dev_xfer *xfers, u32 n_txs, u32 *chal, u32 *tx_err, u32 *rp_err) {
    if (n_txs) hipLaunchKernelGGL(k_transcript, dim3(grid_for(n_txs, 64)), dim3(64), 0, st, bytes, txs, cms, xfers, n_txs, chal, tx_err, rp_err);
}
void launch_blake3_preimages(hipStream_t st, const u8 *bytes, const dev_tx *txs, const u32 *hash_tx, u32 n, u8 *out) {
    if (n) hipLaunchKernelGGL(k_blake3_preimages, dim3(grid_for(n, 64)), dim3(64), 0, st, bytes, txs, hash_tx, n, out);
}

}  // namespace xhe
