// Merlin transcript of every transaction and BLAKE3 of multisig pre-images.
#include "xhe_dev_common.h"
#include "xhe_settings.h"
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


// ---- template transcript: staged byte stream, then one thread per transaction --------------------------
// The transcript itself is data: per block of 168 bytes a list of byte descriptors
// (xhe_transcript_template.h), so absorbing is a gather and no thread interprets STROBE.  Two kernels:
//   k_transcript_stage   gathers every 8-byte state word of every block of every transaction (massively
//                        parallel, at most two unaligned 8-byte loads from the record per word) into a stream laid out
//                        [tile of 64 transactions][block][word][lane], and runs the all-zero-encoding checks;
//   k_transcript_keccak  one thread per transaction with the 25 state words in registers: per block 21
//                        coalesced 8-byte loads, XOR, Keccak-f[1600]; a challenge is squeezed where the
//                        template says so.
// Transactions are taken from a list sorted by template and padded per template to whole tiles, so a
// wavefront runs one template and its control flow is uniform.
static constexpr u32 TR_PAD = 0xffffffffu;
// XHE_TRANSCRIPT_PAIR_MAX_WAVES: largest pass (in wavefronts of the pair kernel) that still runs on lane pairs; 0 = never
static u64 transcript_pair_max_waves() { const char *e = setting("XHE_TRANSCRIPT_PAIR_MAX_WAVES"); return e ? strtoull(e, nullptr, 10) : 2048ull; }
static constexpr u32 STAGE_ROWS = 7;         // words of a block handled per pass of a 448-thread workgroup

// the eight bytes of state word l (< 21) of block b through its word descriptor (transcript_word)
__device__ __forceinline__ u64 transcript_gather(const u32 *__restrict__ desc, u32 b, u32 l, const u8 *__restrict__ rec, const u32 *__restrict__ seg) {
    const uint4 *dp = (const uint4 *)(desc + ((u64)b * TT_WORDS + l) * TT_WDESC);
    uint4 d0 = dp[0], d1 = dp[1];
    u32 d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    return transcript_word(d, rec, seg);
}

__global__ void __launch_bounds__(64 * STAGE_ROWS) k_transcript_stage(const u8 *__restrict__ bytes, const dev_tx *__restrict__ txs, const u32 *__restrict__ tmpl,
                                                                      const u32 *__restrict__ segs, const u32 *__restrict__ list, const transcript_tile *__restrict__ tiles,
                                                                      u32 tile0, u64 *__restrict__ stream, u32 *__restrict__ tx_err, u32 *__restrict__ rp_err, u32 unzip) {
    wave_priority_high();
    const u32 tile = tile0 + blockIdx.x, lane = threadIdx.x & 63, row = threadIdx.x >> 6;
    const transcript_tile tt = tiles[tile];
    const u32 i = list[64ull * tile + lane];
    if (i == TR_PAD) return;
    const u8 *rec = bytes + txs[i].rec;
    const u32 *seg = segs + txs[i].seg_off;
    const u32 *hdr = tmpl + tt.tmpl_off;
    const u32 n_blocks = hdr[TT_H_BLOCKS], n_checks = hdr[TT_H_CHECKS];
    const u32 *checks = hdr + TT_H_WORDS + n_blocks, *desc = tmpl + hdr[TT_H_DESC];
    if (blockIdx.y == 0) {
        // encodings that must not be all-zero (validate_and_append_point)
        for (u32 c = row; c < n_checks; c += STAGE_ROWS) {
            u32 ds = checks[2 * c], key = checks[2 * c + 1];
            const u8 *p = rec + seg[ds >> 16] + (ds & 0xffffu);
            u32 acc = 0;
#pragma unroll
            for (int k = 0; k < 32; k++) acc |= p[k];
            if (!acc) { if (key) raise_err(tx_err, i, key); else rp_err[i] = 1; }
        }
    }
    u64 *out = stream + tt.stream_off + lane;
    for (u32 b = blockIdx.y; b < n_blocks; b += gridDim.y) {
#pragma unroll
        for (u32 w = row; w < 21; w += STAGE_ROWS) {
            u64 v = transcript_gather(desc, b, w, rec, seg);
            out[(u64)(b * 21 + w) * 64] = unzip ? bit_unzip64(v) : v;         // for k_transcript_keccak_pair: even bits | odd bits
        }
    }
}

__global__ void __launch_bounds__(64) k_transcript_keccak(const dev_tx *__restrict__ txs, const u32 *__restrict__ tmpl, const u32 *__restrict__ list,
                                                          const transcript_tile *__restrict__ tiles, u32 tile0, const u64 *__restrict__ stream, u32 *__restrict__ chal) {
    wave_priority_high();
    const u32 tile = tile0 + blockIdx.x, lane = threadIdx.x;
    const u32 tmpl_off = __builtin_amdgcn_readfirstlane(tiles[tile].tmpl_off);
    const u64 stream_off = tiles[tile].stream_off;
    const u32 i = list[64ull * tile + lane];
    if (i == TR_PAD) return;
    const u32 chal_off = txs[i].chal_off;
    const u32 *hdr = tmpl + tmpl_off;
    const u32 n_blocks = __builtin_amdgcn_readfirstlane(hdr[TT_H_BLOCKS]);
    const u32 *chal_after = hdr + TT_H_WORDS;
    const u64 *in = stream + stream_off + lane;
    u64 a[25], nx[21];
#pragma unroll
    for (int w = 0; w < 25; w++) a[w] = 0;
#pragma unroll
    for (int w = 0; w < 21; w++) nx[w] = in[64 * w];
    for (u32 b = 0; b < n_blocks; b++) {
#pragma unroll
        for (int w = 0; w < 21; w++) a[w] ^= nx[w];
        if (b + 1 < n_blocks) {                        // in flight during the permutation
#pragma unroll
            for (int w = 0; w < 21; w++) nx[w] = in[(u64)((b + 1) * 21 + w) * 64];
        }
        keccak_f1600_regs(a);
        u32 ci = __builtin_amdgcn_readfirstlane(chal_after[b]);
        if (ci != TT_NONE) {
            u32 w[16];
#pragma unroll
            for (int k = 0; k < 8; k++) { w[2 * k] = (u32)a[k]; w[2 * k + 1] = (u32)(a[k] >> 32); a[k] = 0; }
            store_scalar_words(chal + 8ull * (chal_off + ci), sc_from_wide_words(w));
        }
    }
}

// The same on a pair of lanes per transaction (bit-interleaved state, keccak_f1600_pair): a workgroup of two wavefronts per
// tile, lane pair t of the workgroup runs transaction t of the tile.  Reads the stream k_transcript_stage wrote with
// unzip = 1 (word = even bits | odd bits << 32): lane 2t takes the low, lane 2t+1 the high 32 bits.
__global__ void __launch_bounds__(128) k_transcript_keccak_pair(const dev_tx *__restrict__ txs, const u32 *__restrict__ tmpl, const u32 *__restrict__ list,
                                                               const transcript_tile *__restrict__ tiles, u32 tile0, const u64 *__restrict__ stream, u32 *__restrict__ chal) {
    wave_priority_high();
    const u32 tile = tile0 + blockIdx.x, t = threadIdx.x >> 1, odd = threadIdx.x & 1;
    const u32 tmpl_off = __builtin_amdgcn_readfirstlane(tiles[tile].tmpl_off);
    const u64 stream_off = tiles[tile].stream_off;
    const u32 i = list[64ull * tile + t];
    if (i == TR_PAD) return;
    const u32 chal_off = txs[i].chal_off;
    const u32 *hdr = tmpl + tmpl_off;
    const u32 n_blocks = __builtin_amdgcn_readfirstlane(hdr[TT_H_BLOCKS]);
    const u32 *chal_after = hdr + TT_H_WORDS;
    const u32 *in = (const u32 *)(stream + stream_off) + threadIdx.x;
    const keccak_pair_consts kc = keccak_pair_setup(odd);
    u32 s[25], nx[21];
#pragma unroll
    for (int w = 0; w < 25; w++) s[w] = 0;
#pragma unroll
    for (int w = 0; w < 21; w++) nx[w] = in[128 * w];
    for (u32 b = 0; b < n_blocks; b++) {
#pragma unroll
        for (int w = 0; w < 21; w++) s[w] ^= nx[w];
        if (b + 1 < n_blocks) {                        // in flight during the permutation
#pragma unroll
            for (int w = 0; w < 21; w++) nx[w] = in[(u64)((b + 1) * 21 + w) * 128];
        }
        keccak_f1600_pair(s, kc, odd);
        u32 ci = __builtin_amdgcn_readfirstlane(chal_after[b]);
        if (ci != TT_NONE) {
            // both lanes rebuild the 64 bytes; the even lane reduces and stores
            u32 w[16];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                u32 other = pair_swap(s[k]);
                u64 v = bit_zip64(odd ? other : s[k], odd ? s[k] : other);
                w[2 * k] = (u32)v; w[2 * k + 1] = (u32)(v >> 32); s[k] = 0;
            }
            if (!odd) store_scalar_words(chal + 8ull * (chal_off + ci), sc_from_wide_words(w));
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


void launch_transcript(hipStream_t st, const u8 *bytes, const dev_tx *txs, const dev_cm *cms, const dev_xfer *xfers, const u32 *tmpl, const u32 *segs,
                       const transcript_plan &plan, const u32 *list, const transcript_tile *tiles, u64 *stream, const u32 *serial_list, u32 n_serial, u32 *chal,
                       u32 *tx_err, u32 *rp_err) {
    for (const transcript_pass &ps : plan.passes) {
        u32 gy = ps.max_blocks < 64 ? ps.max_blocks : 64;
        // A pair of lanes per transaction while the wavefronts of a pass still find a SIMD each (the chain of permutations is
        // bound by one wavefront's issue rate and a lane of a pair issues ~45 % fewer instructions); one lane per transaction
        // beyond that, where the total instruction count decides (it is ~10 % higher for the pairs).
        const bool pairs = 2ull * ps.n_tiles <= transcript_pair_max_waves();
        hipLaunchKernelGGL(k_transcript_stage, dim3(ps.n_tiles, gy ? gy : 1), dim3(64 * STAGE_ROWS), 0, st, bytes, txs, tmpl, segs, list, tiles, ps.tile0, stream, tx_err, rp_err,
                           pairs ? 1u : 0u);
        if (pairs) hipLaunchKernelGGL(k_transcript_keccak_pair, dim3(ps.n_tiles), dim3(128), 0, st, txs, tmpl, list, tiles, ps.tile0, stream, chal);
        else hipLaunchKernelGGL(k_transcript_keccak, dim3(ps.n_tiles), dim3(64), 0, st, txs, tmpl, list, tiles, ps.tile0, stream, chal);
    }
    if (n_serial) hipLaunchKernelGGL(k_transcript_serial, dim3(grid_for(n_serial, 64)), dim3(64), 0, st, bytes, txs, cms, xfers, serial_list, n_serial, chal, tx_err, rp_err);
}
void launch_blake3_preimages(hipStream_t st, const u8 *bytes, const dev_tx *txs, const u32 *hash_tx, u32 n, u8 *out) {
    if (n) hipLaunchKernelGGL(k_blake3_preimages, dim3(grid_for(n, 64)), dim3(64), 0, st, bytes, txs, hash_tx, n, out);
}

}  // namespace xhe
