// Merlin transcript of every transaction (thread per transaction) and BLAKE3 of multisig pre-images.
#include "xhe_dev_common.h"
#include "xhe_kernels.h"
#include "xhe_transcript_template.h"

namespace xhe {

// ---- transcript ------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_zero32_dev(const u8 *p) { u8 a = 0; for (int i = 0; i < 32; i++) a |= p[i]; return a == 0; }

// Thread per transaction, STROBE interpreted byte by byte: the path for the shapes the template kernel
// does not take (list = their transaction indices).
__global__ void __launch_bounds__(64) k_transcript_serial(const u8 *__restrict__ bytes, const dev_tx *__restrict__ txs, const dev_cm *__restrict__ cms,
                                                          const dev_xfer *__restrict__ xfers, const u32 *__restrict__ list, u32 n_list, u32 *__restrict__ chal,
                                                          u32 *__restrict__ tx_err, u32 *__restrict__ rp_err) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_list) return;
    const u32 i = list[g];
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


// ---- template transcript: 25 lanes hold the Keccak state of one transaction ---------------------------
// Lane l = x + 5y of a 32-lane group holds state word A[x][y]; theta, pi and chi become lane shuffles
// inside the group, rho a per-lane rotate.  Two transactions share a wavefront.  The transcript itself
// is data: per block of 168 bytes a list of byte descriptors (xhe_transcript_template.h), so absorbing is
// a gather and no lane interprets STROBE.
struct keccak_lane_consts {
    int col[4];        // the other four lanes of this lane's column
    int xm1, xp1;      // lanes holding columns x-1 and x+1 (any row)
    int pi_src;        // lane whose rotated word lands here
    int x1, x2;        // lanes x+1, x+2 of this lane's row
    u32 rot;           // rho offset of this lane's own word
};
__device__ __forceinline__ keccak_lane_consts keccak_lane_setup(u32 l) {
    keccak_lane_consts k;
    if (l >= 25) { for (int i = 0; i < 4; i++) k.col[i] = (int)l; k.xm1 = k.xp1 = k.pi_src = k.x1 = k.x2 = (int)l; k.rot = 0; return k; }
    u32 x = l % 5, y = l / 5;
    for (u32 i = 0; i < 4; i++) k.col[i] = (int)(x + 5 * ((y + 1 + i) % 5));
    k.xm1 = (int)((x + 4) % 5 + 5 * y); k.xp1 = (int)((x + 1) % 5 + 5 * y);
    k.pi_src = (int)((x + 3 * y) % 5 + 5 * x);                  // B[y][2x+3y] = rot(A[x][y]) solved for the source
    k.x1 = (int)((x + 1) % 5 + 5 * y); k.x2 = (int)((x + 2) % 5 + 5 * y);
    const u8 RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    k.rot = RHO[l];
    return k;
}
__device__ __forceinline__ u64 shfl64_group(u64 v, int src) {
    u32 lo = (u32)__shfl((int)(u32)v, src, 32), hi = (u32)__shfl((int)(u32)(v >> 32), src, 32);
    return (u64)lo | ((u64)hi << 32);
}
__device__ __forceinline__ u64 keccak_f1600_lanes(u64 a, const keccak_lane_consts &k, u32 l) {
    const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#pragma unroll 2
    for (int round = 0; round < 24; round++) {
        u64 c = a ^ shfl64_group(a, k.col[0]) ^ shfl64_group(a, k.col[1]) ^ shfl64_group(a, k.col[2]) ^ shfl64_group(a, k.col[3]);
        u64 d = shfl64_group(c, k.xm1) ^ rotl64(shfl64_group(c, k.xp1), 1);
        a ^= d;
        u64 r = (a << k.rot) | (a >> ((64 - k.rot) & 63));
        u64 b = shfl64_group(r, k.pi_src);
        u64 b1 = shfl64_group(b, k.x1), b2 = shfl64_group(b, k.x2);
        a = b ^ (~b1 & b2);
        if (l == 0) a ^= RC[round];
    }
    return a;
}

// the eight bytes of block `b` that belong to lane l (l < 21), assembled from the descriptors
__device__ __forceinline__ u64 transcript_gather(const u32 *__restrict__ desc, u32 b, u32 l, const u8 *__restrict__ rec, const u32 *__restrict__ seg) {
    if (l >= 21) return 0;
    const uint4 *dp = (const uint4 *)(desc + (u64)b * TT_BLOCK_BYTES + 8 * l);
    uint4 d0 = dp[0], d1 = dp[1];
    u32 ds[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    u64 word = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        bool is_const = (ds[k] & TT_CONST) != 0;
        u32 off = is_const ? 0u : seg[ds[k] >> 16] + (ds[k] & 0xffffu);
        u32 byte = rec[off];
        byte = is_const ? (ds[k] & 0xffu) : byte;
        word |= (u64)byte << (8 * k);
    }
    return word;
}

__global__ void __launch_bounds__(256) k_transcript(const u8 *__restrict__ bytes, const dev_tx *__restrict__ txs, const u32 *__restrict__ tmpl,
                                                    const u32 *__restrict__ segs, u32 n_txs, u32 *__restrict__ chal, u32 *__restrict__ tx_err,
                                                    u32 *__restrict__ rp_err) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    u32 i = g >> 5, l = g & 31;
    if (i >= n_txs) return;
    const u32 tmpl_off = txs[i].tmpl_off;
    if (tmpl_off == TT_NONE) return;
    const u8 *rec = bytes + txs[i].rec;
    const u32 *seg = segs + txs[i].seg_off;
    const u32 chal_off = txs[i].chal_off;
    const u32 *hdr = tmpl + tmpl_off;
    const u32 n_blocks = hdr[TT_H_BLOCKS], n_checks = hdr[TT_H_CHECKS];
    const u32 *chal_after = hdr + TT_H_WORDS, *checks = chal_after + n_blocks, *desc = tmpl + hdr[TT_H_DESC];
    // encodings that must not be all-zero
    for (u32 c = l; c < n_checks; c += 32) {
        u32 ds = checks[2 * c], key = checks[2 * c + 1];
        const u8 *p = rec + seg[ds >> 16] + (ds & 0xffffu);
        u32 acc = 0;
        for (int k = 0; k < 32; k++) acc |= p[k];
        if (!acc) { if (key) raise_err(tx_err, i, key); else rp_err[i] = 1; }
    }
    const keccak_lane_consts kc = keccak_lane_setup(l);
    u64 a = 0;
    u64 word = transcript_gather(desc, 0, l, rec, seg);
    for (u32 b = 0; b < n_blocks; b++) {
        a ^= word;
        if (b + 1 < n_blocks) word = transcript_gather(desc, b + 1, l, rec, seg);     // in flight during the permutation
        a = keccak_f1600_lanes(a, kc, l);
        u32 ci = chal_after[b];
        if (ci != TT_NONE) {
            u32 w[16];
#pragma unroll
            for (int k = 0; k < 8; k++) { w[2 * k] = (u32)__shfl((int)(u32)a, k, 32); w[2 * k + 1] = (u32)__shfl((int)(u32)(a >> 32), k, 32); }
            sc x = sc_from_wide_words(w);
            if (l == 0) store_scalar_words(chal + 8ull * (chal_off + ci), x);
            if (l < 8) a = 0;
        }
    }
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


void launch_transcript(hipStream_t st, const u8 *bytes, const dev_tx *txs, const dev_cm *cms, const dev_xfer *xfers, u32 n_txs, const u32 *tmpl, const u32 *segs,
                       const u32 *serial_list, u32 n_serial, u32 *chal, u32 *tx_err, u32 *rp_err) {
    if (n_txs > n_serial) hipLaunchKernelGGL(k_transcript, dim3(grid_for(32ull * n_txs, 256)), dim3(256), 0, st, bytes, txs, tmpl, segs, n_txs, chal, tx_err, rp_err);
    if (n_serial) hipLaunchKernelGGL(k_transcript_serial, dim3(grid_for(n_serial, 64)), dim3(64), 0, st, bytes, txs, cms, xfers, serial_list, n_serial, chal, tx_err, rp_err);
}
void launch_blake3_preimages(hipStream_t st, const u8 *bytes, const dev_tx *txs, const u32 *hash_tx, u32 n, u8 *out) {
    if (n) hipLaunchKernelGGL(k_blake3_preimages, dim3(grid_for(n, 64)), dim3(64), 0, st, bytes, txs, hash_tx, n, out);
}

}  // namespace xhe
