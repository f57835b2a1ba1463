// Kernels of Transaction::verify_batch (/root/reference/src/tx/verify.rs:487-517) on gfx950.
//
//   k_decode_list       thread per compressed point: RFC 9496 decode -> 128-byte affine-Niels record
//   k_transcript        thread per transaction: the whole Merlin transcript, all Fiat-Shamir challenges
//   k_scan_*            the ledger as a prefix sum: every balance read by a commitment-equality proof is
//                       the prefix of the (account, asset) slot's operations in transaction order
//   k_sigma_cm/_xf      thread per sigma proof: the 7/8 batched scalars of proofs.rs:188-208,336-358
//   k_rp_prep/_matrix/_colsum  range-proof verification scalars, summed over the batch per generator
//   k_schnorr           thread per signature: R = s*H - e*P, SHA3-512(pk || msg || R) mod l == e
//   k_point_errors, k_verdict  error precedence of pre_verify folded to (first failing tx, code)
// The multiscalar multiplication itself is xhe_msm.hip.h.
#pragma once
#include "xhe_hash.h"
#include "xhe_msm.hip.h"
#include "xhe_verify_types.h"
#include "../../include/xhe.h"

namespace xhe {

__device__ __forceinline__ void load_words_unaligned(u32 w[8], const u8 *__restrict__ p) {
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = (u32)p[4 * i] | ((u32)p[4 * i + 1] << 8) | ((u32)p[4 * i + 2] << 16) | ((u32)p[4 * i + 3] << 24);
}
__device__ __forceinline__ sc load_scalar_bytes(const u8 *__restrict__ p) { u32 w[8]; load_words_unaligned(w, p); return sc_from_words(w); }
__device__ __forceinline__ void store_scalar_words(u32 *__restrict__ dst, const sc &x) {
    u32 w[8]; sc_to_words(w, x);
#pragma unroll
    for (int i = 0; i < 8; i++) dst[i] = w[i];
}
__device__ __forceinline__ sc load_scalar_words(const u32 *__restrict__ src) {
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = src[i];
    return sc_from_words(w);
}
// Montgomery-form limbs in memory (9 words)
__device__ __forceinline__ void store_sc_limbs(u32 *__restrict__ dst, const sc &x) {
#pragma unroll
    for (int i = 0; i < 9; i++) dst[i] = x.v[i];
}
__device__ __forceinline__ sc load_sc_limbs(const u32 *__restrict__ src) {
    sc x;
#pragma unroll
    for (int i = 0; i < 9; i++) x.v[i] = src[i];
    return x;
}
__device__ __forceinline__ void raise_err(u32 *tx_err, u32 tx, u32 key) { atomicMin(&tx_err[tx], key); }

// batch factor: Keccak-f over (seed, domain, index), 64 bytes reduced mod l
__device__ __forceinline__ sc prf_scalar(const u32 *__restrict__ seed8, u32 domain, u64 index) {
    u64 st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) st[i] = (u64)seed8[2 * i] | ((u64)seed8[2 * i + 1] << 32);
    st[4] = index; st[5] = domain; st[6] = 0x1f; st[16] = 0x8000000000000000ULL;
    keccak_f1600(st);
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { w[2 * i] = (u32)st[i]; w[2 * i + 1] = (u32)(st[i] >> 32); }
    return sc_from_wide_words(w);
}

// ---- decode ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) k_decode_list(const u8 *__restrict__ bytes, const dev_dec *__restrict__ decs, u32 n, u32 pt_base,
                                                        u32 *__restrict__ pts, u8 *__restrict__ ok) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8]; load_words_unaligned(w, bytes + decs[i].off);
    ge p; bool good = ristretto_decode_words(p, w);
    ge_niels_store(pts + 32ull * (pt_base + i), ge_to_niels_affine(p));
    ok[pt_base + i] = good ? 1 : 0;
}

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

// ---- fixed-base tables: tab[w][b] = b * 256^w * P as affine Niels, b in 0..255 -----------------------
__device__ __forceinline__ ge ge_from_niels_affine(const ge_niels &q) {
    // x = (ypx - ymx)/2, y = (ypx + ymx)/2
    fe inv2 = fe_zero();            // (p+1)/2 = 2^254 - 9
    inv2.v[0] = (1u << 26) - 9;
#pragma unroll
    for (int i = 1; i < 9; i++) inv2.v[i] = (i & 1) ? FE_MASK25 : FE_MASK26;
    inv2.v[9] = (1u << 24) - 1;
    ge p;
    p.X = fe_mul(fe_sub(q.ypx, q.ymx), inv2);
    p.Y = fe_mul(fe_add(q.ypx, q.ymx), inv2);
    p.Z = fe_one();
    p.T = fe_mul(p.X, p.Y);
    return p;
}
__device__ __forceinline__ ge_niels ge_to_niels_normalize(const ge &p) {
    fe zi = fe_invert(p.Z);
    ge a; a.X = fe_mul(p.X, zi); a.Y = fe_mul(p.Y, zi); a.Z = fe_one(); a.T = fe_mul(a.X, a.Y);
    return ge_to_niels_affine(a);
}
__global__ void __launch_bounds__(256, 2) k_build_fixed_table(const u32 *__restrict__ base_niels, u32 n_windows, u32 *__restrict__ tab) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_windows * 256) return;
    u32 w = g >> 8, b = g & 255;
    ge p = ge_from_niels_affine(ge_niels_load(base_niels));
    for (u32 k = 0; k < 8 * w; k++) p = ge_double(p);
    ge r = ge_identity();
    for (int bit = 7; bit >= 0; bit--) { r = ge_double(r); if ((b >> bit) & 1) r = ge_add(r, p); }
    ge_niels_store(tab + 32ull * g, ge_to_niels_normalize(r));
}

// ---- ledger scan -----------------------------------------------------------------------------------
struct ct_acc { ge C, D; };
__device__ __forceinline__ ct_acc ct_identity() { ct_acc a; a.C = ge_identity(); a.D = ge_identity(); return a; }
__device__ __forceinline__ ct_acc ct_add(const ct_acc &a, const ct_acc &b) { ct_acc r; r.C = ge_add(a.C, b.C); r.D = ge_add(a.D, b.D); return r; }
__device__ __forceinline__ void ct_store(u32 *dst, const ct_acc &a) { ge_store(dst, a.C); ge_store(dst + 32, a.D); }
__device__ __forceinline__ ct_acc ct_load(const u32 *src) { ct_acc a; a.C = ge_load(src); a.D = ge_load(src + 32); return a; }

__device__ __forceinline__ void scan_apply_op(ct_acc &acc, const dev_op &op, const u32 *__restrict__ pts, const u32 *__restrict__ gtab) {
    if (op.kind == 2) {
        u64 a = op.amount;
#pragma unroll 1
        for (u32 w = 0; w < 8 && a; w++, a >>= 8) {
            u32 b = (u32)a & 255;
            if (b) acc.C = ge_msub(acc.C, ge_niels_load(gtab + 32ull * (w * 256 + b)));
        }
    } else {
        bool neg = op.kind == 1;
        acc.C = ge_madd_signed(acc.C, ge_niels_load(pts + 32ull * op.c_pt), neg);
        acc.D = ge_madd_signed(acc.D, ge_niels_load(pts + 32ull * op.d_pt), neg);
    }
}

static constexpr u32 SCAN_K = 4;          // operations per thread
static constexpr u32 SCAN_BLOCK = 128;    // threads per block

// inclusive Hillis-Steele scan of one ct per thread through LDS; returns the inclusive value of thread t
__device__ __forceinline__ ct_acc block_scan_ct(ct_acc v, u32 *lds /* SCAN_BLOCK * 64 words */) {
    u32 t = threadIdx.x;
    for (u32 d = 1; d < SCAN_BLOCK; d <<= 1) {
        ct_store(lds + 64 * t, v);
        __syncthreads();
        if (t >= d) { ct_acc o = ct_load(lds + 64 * (t - d)); v = ct_add(o, v); }
        __syncthreads();
    }
    return v;
}

// phase 1: per-thread sums, block scan; writes each thread's exclusive prefix inside its block and the block total
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_local(const dev_op *__restrict__ ops, u32 n_ops, const u32 *__restrict__ pts, const u32 *__restrict__ gtab,
                                                           u32 *__restrict__ thread_excl, u32 *__restrict__ block_tot) {
    __shared__ u32 lds[SCAN_BLOCK * 64];
    u32 t = threadIdx.x, gt = blockIdx.x * SCAN_BLOCK + t;
    ct_acc acc = ct_identity();
    for (u32 k = 0; k < SCAN_K; k++) {
        u32 o = gt * SCAN_K + k;
        if (o < n_ops) scan_apply_op(acc, ops[o], pts, gtab);
    }
    ct_acc incl = block_scan_ct(acc, lds);
    // exclusive = inclusive of the previous thread
    ct_store(lds + 64 * t, incl);
    __syncthreads();
    ct_acc excl = t ? ct_load(lds + 64 * (t - 1)) : ct_identity();
    ct_store(thread_excl + 64ull * gt, excl);
    if (t == SCAN_BLOCK - 1) ct_store(block_tot + 64ull * blockIdx.x, incl);
}
// phase 2: one block turns the block totals into exclusive block prefixes (in place)
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_blocks(u32 *__restrict__ block_tot, u32 n_blocks) {
    __shared__ u32 lds[SCAN_BLOCK * 64];
    u32 t = threadIdx.x;
    u32 per = (n_blocks + SCAN_BLOCK - 1) / SCAN_BLOCK;
    u32 lo = t * per, hi = lo + per; if (hi > n_blocks) hi = n_blocks;
    ct_acc acc = ct_identity();
    for (u32 b = lo; b < hi; b++) acc = ct_add(acc, ct_load(block_tot + 64ull * b));
    ct_acc incl = block_scan_ct(acc, lds);
    ct_store(lds + 64 * t, incl);
    __syncthreads();
    ct_acc run = t ? ct_load(lds + 64 * (t - 1)) : ct_identity();
    for (u32 b = lo; b < hi; b++) {
        ct_acc v = ct_load(block_tot + 64ull * b);
        ct_store(block_tot + 64ull * b, run);
        run = ct_add(run, v);
    }
}
// phase 3: global inclusive prefixes at the operations that need them
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply(const dev_op *__restrict__ ops, u32 n_ops, const u32 *__restrict__ pts, const u32 *__restrict__ gtab,
                                                           const u32 *__restrict__ thread_excl, const u32 *__restrict__ block_excl,
                                                           const u32 *__restrict__ slot_last_op, u32 *__restrict__ slot_base, u32 *__restrict__ slot_final,
                                                           u32 *__restrict__ emit_incl) {
    u32 gt = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    if (gt * SCAN_K >= n_ops) return;
    ct_acc acc = ct_add(ct_load(block_excl + 64ull * blockIdx.x), ct_load(thread_excl + 64ull * gt));
    for (u32 k = 0; k < SCAN_K; k++) {
        u32 o = gt * SCAN_K + k;
        if (o >= n_ops) break;
        dev_op op = ops[o];
        if (op.first) ct_store(slot_base + 64ull * op.slot, acc);          // prefix before the slot's initial balance
        scan_apply_op(acc, op, pts, gtab);
        if (op.emit != 0xffffffffu) ct_store(emit_incl + 64ull * op.emit, acc);
        if (slot_last_op[op.slot] == o) ct_store(slot_final + 64ull * op.slot, acc);
    }
}
// phase 4a: sender ciphertexts read by the equality proofs, as affine-Niels points
__global__ void __launch_bounds__(256, 2) k_scan_emit(const u32 *__restrict__ emit_incl, const u32 *__restrict__ emit_slot, const u32 *__restrict__ slot_base,
                                                      u32 n_emit, u32 emit_pt_base, u32 *__restrict__ pts, u8 *__restrict__ ok) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;     // one thread per point (2 per emitted ciphertext)
    if (g >= 2 * n_emit) return;
    u32 e = g >> 1, which = g & 1;
    ge incl = ge_load(emit_incl + 64ull * e + 32 * which);
    ge base = ge_load(slot_base + 64ull * emit_slot[e] + 32 * which);
    ge r = ge_sub(incl, base);
    ge_niels_store(pts + 32ull * (emit_pt_base + g), ge_to_niels_normalize(r));
    ok[emit_pt_base + g] = 1;
}
// phase 4b: final balance of every touched slot, compressed
__global__ void __launch_bounds__(256, 2) k_scan_final(const u32 *__restrict__ slot_final, const u32 *__restrict__ slot_base, u32 n_slots, u32 *__restrict__ out_words) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * n_slots) return;
    u32 s = g >> 1, which = g & 1;
    ge r = ge_sub(ge_load(slot_final + 64ull * s + 32 * which), ge_load(slot_base + 64ull * s + 32 * which));
    u32 w[8]; ristretto_encode_words(w, r);
#pragma unroll
    for (int i = 0; i < 8; i++) out_words[8ull * g + i] = w[i];
}

// ---- sigma proofs ----------------------------------------------------------------------------------
// gh[2p], gh[2p+1]: this proof's contributions to the scalars of G and H (Montgomery limbs)
__global__ void __launch_bounds__(256, 2) k_sigma_cm(const u8 *__restrict__ bytes, const dev_cm *__restrict__ cms, const dev_tx *__restrict__ txs, u32 n,
                                                     const u32 *__restrict__ chal, const u32 *__restrict__ seed, u32 *__restrict__ scalars,
                                                     u32 *__restrict__ term_point, u32 *__restrict__ gh) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dev_cm c = cms[i];
    const dev_tx d = txs[c.tx];
    const u8 *p = bytes + c.off;
    sc ch_c = load_scalar_words(chal + 8ull * (d.chal_off + 2 * c.idx)), ch_w = load_scalar_words(chal + 8ull * (d.chal_off + 2 * c.idx + 1));
    sc z_s = load_scalar_bytes(p + 160), z_x = load_scalar_bytes(p + 192), z_r = load_scalar_bytes(p + 224);
    sc bf = prf_scalar(seed, 1, i);
    sc ww = sc_mul(ch_w, ch_w);
    sc w_bf = sc_mul(ch_w, bf), ww_bf = sc_mul(ww, bf);
    sc neg_bf = sc_neg(bf), neg_w_bf = sc_neg(w_bf), neg_ww_bf = sc_neg(ww_bf);
    u32 *s = scalars + 8ull * c.term_off;
    u32 *tp = term_point + c.term_off;
    store_scalar_words(s, sc_mul(z_s, bf));                  tp[0] = d.src_pt;          // z_s * P_source
    store_scalar_words(s + 8, neg_bf);                        tp[1] = c.pt + 1;          // -Y_0
    store_scalar_words(s + 16, sc_mul(w_bf, z_s));            tp[2] = c.src_ct_pt + 1;   // w z_s * D_source
    store_scalar_words(s + 24, sc_mul(neg_w_bf, ch_c));       tp[3] = c.src_ct_pt;       // -w c * C_source
    store_scalar_words(s + 32, neg_w_bf);                     tp[4] = c.pt + 2;          // -w * Y_1
    store_scalar_words(s + 40, sc_mul(neg_ww_bf, ch_c));      tp[5] = c.pt;              // -ww c * C_destination
    store_scalar_words(s + 48, neg_ww_bf);                    tp[6] = c.pt + 3;          // -ww * Y_2
    // G: (w + ww) z_x bf ; H: (-c + ww z_r) bf
    sc g = sc_mul(sc_add(w_bf, ww_bf), z_x);
    sc h = sc_add(sc_mul(neg_bf, ch_c), sc_mul(ww_bf, z_r));
    store_sc_limbs(gh + 18ull * i, sc_norm(g));
    store_sc_limbs(gh + 18ull * i + 9, sc_norm(h));
}
__global__ void __launch_bounds__(256, 2) k_sigma_xf(const u8 *__restrict__ bytes, const dev_xfer *__restrict__ xfers, const dev_tx *__restrict__ txs, u32 n,
                                                     const u32 *__restrict__ chal, const u32 *__restrict__ seed, u32 *__restrict__ scalars,
                                                     u32 *__restrict__ term_point, u32 *__restrict__ gh) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dev_xfer x = xfers[i];
    const dev_tx d = txs[x.tx];
    const u8 *pr = bytes + x.proof_off;
    u32 cbase = d.chal_off + 2 * d.n_commitments + 2 * x.idx;
    sc ch_c = load_scalar_words(chal + 8ull * cbase), ch_w = load_scalar_words(chal + 8ull * (cbase + 1));
    sc z_r = load_scalar_bytes(pr + 96), z_x = load_scalar_bytes(pr + 128);
    sc bf = prf_scalar(seed, 2, i);
    sc w_bf = sc_mul(ch_w, bf), ww_bf = sc_mul(sc_mul(ch_w, ch_w), bf);
    sc neg_bf = sc_neg(bf), neg_w_bf = sc_neg(w_bf), neg_ww_bf = sc_neg(ww_bf);
    u32 *s = scalars + 8ull * x.term_off;
    u32 *tp = term_point + x.term_off;
    store_scalar_words(s, sc_mul(neg_bf, ch_c));              tp[0] = x.pt;              // -c * C
    store_scalar_words(s + 8, neg_bf);                        tp[1] = x.pt + 3;          // -Y_0
    store_scalar_words(s + 16, sc_mul(w_bf, z_r));            tp[2] = x.dest_pt;         // w z_r * P_dest
    store_scalar_words(s + 24, sc_mul(neg_w_bf, ch_c));       tp[3] = x.pt + 2;          // -w c * D_dest
    store_scalar_words(s + 32, neg_w_bf);                     tp[4] = x.pt + 4;          // -w * Y_1
    store_scalar_words(s + 40, sc_mul(ww_bf, z_r));           tp[5] = d.src_pt;          // ww z_r * P_source
    store_scalar_words(s + 48, sc_mul(neg_ww_bf, ch_c));      tp[6] = x.pt + 1;          // -ww c * D_source
    store_scalar_words(s + 56, neg_ww_bf);                    tp[7] = x.pt + 5;          // -ww * Y_2
    store_sc_limbs(gh + 18ull * i, sc_norm(sc_mul(z_x, bf)));
    store_sc_limbs(gh + 18ull * i + 9, sc_norm(sc_mul(z_r, bf)));
}

// sum of n scalars (Montgomery limbs, stride words apart) per residue class j = 0..classes-1:
// out[b*classes + j] = sum over i in block b's share of in[(i*classes + j)]
__device__ __forceinline__ sc sc_reduce_mont(const sc &a) { return sc_mul(a, sc_one()); }
__global__ void __launch_bounds__(256) k_sc_sum_partial(const u32 *__restrict__ in, u32 n, u32 classes, u32 *__restrict__ out) {
    __shared__ u32 lds[256 * 9];
    u32 t = threadIdx.x;
    for (u32 j = 0; j < classes; j++) {
        sc acc = sc_zero(); u32 cnt = 0;
        for (u32 i = blockIdx.x * 256 + t; i < n; i += gridDim.x * 256) {
            acc = sc_add(acc, load_sc_limbs(in + 9ull * ((u64)i * classes + j)));
            if (++cnt == 16) { acc = sc_reduce_mont(acc); cnt = 0; }
        }
        acc = sc_reduce_mont(acc);
        for (u32 d = 128; d >= 1; d >>= 1) {
            store_sc_limbs(lds + 9 * t, acc);
            __syncthreads();
            if (t < d) acc = sc_reduce_mont(sc_add(acc, load_sc_limbs(lds + 9 * (t + d))));
            __syncthreads();
        }
        if (t == 0) store_sc_limbs(out + 9ull * (blockIdx.x * classes + j), acc);
    }
}

// ---- range proofs ----------------------------------------------------------------------------------
// per-transaction context for the matrix kernel, Montgomery limbs (9 words each)
static constexpr u32 RPC_NEG_RHO_Z = 0, RPC_RHO_Z = 1, RPC_COEF_G = 2, RPC_COEF_H2 = 3, RPC_COEF_H3 = 4, RPC_U2 = 5 /* 16 */, RPC_Y2INV = 21 /* 16 */,
                     RPC_E = 37 /* 6 */, RPC_F = 43 /* 10 */, RPC_SCALARS = 53;

__global__ void __launch_bounds__(64) k_rp_prep(const u8 *__restrict__ bytes, const dev_tx *__restrict__ txs, const dev_cm *__restrict__ cms,
                                                const dev_xfer *__restrict__ xfers, u32 n_txs, const u32 *__restrict__ chal, const u32 *__restrict__ seed,
                                                u32 *__restrict__ scalars, u32 *__restrict__ term_point, u32 *__restrict__ rp_ctx, u32 *__restrict__ gh,
                                                u32 *__restrict__ rp_err) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_txs) return;
    const dev_tx d = txs[i];
    const u32 lg = d.lg, A = d.n_commitments, T = d.n_transfers;
    const u32 n_dyn = 4 + 2 * lg + A + T;
    u32 *s = scalars + 8ull * d.rp_term_off;
    u32 *tp = term_point + d.rp_term_off;
    // term -> point map is valid whatever happens below
    for (u32 k = 0; k < 4 + 2 * lg; k++) tp[k] = d.rp_pt + k;
    for (u32 k = 0; k < A; k++) tp[4 + 2 * lg + k] = cms[d.cm_first + k].pt;
    for (u32 k = 0; k < T; k++) tp[4 + 2 * lg + A + k] = xfers[d.xfer_first + k].pt;
    bool valid = lg >= 6 && lg <= 15 && (64ull * d.m == (1ull << lg)) && d.m <= 512;
    if (!valid) {
        rp_err[i] = 1;
        for (u32 k = 0; k < 8 * n_dyn; k++) s[k] = 0;
        store_sc_limbs(gh + 18ull * i, sc_zero()); store_sc_limbs(gh + 18ull * i + 9, sc_zero());
        return;
    }
    const u8 *rp = bytes + d.rec + d.rp_off;
    const u32 *ch = chal + 8ull * (d.chal_off + 2 * A + 2 * T);
    sc y = load_scalar_words(ch), z = load_scalar_words(ch + 8), x = load_scalar_words(ch + 16), w = load_scalar_words(ch + 24);
    sc t_x = load_scalar_bytes(rp + 128), t_x_bl = load_scalar_bytes(rp + 160), e_bl = load_scalar_bytes(rp + 192);
    sc a = load_scalar_bytes(rp + 224 + 64 * lg), b = load_scalar_bytes(rp + 224 + 64 * lg + 32);
    sc rho = prf_scalar(seed, 3, i), c = prf_scalar(seed, 4, i);
    u32 *ctx = rp_ctx + 9ull * RPC_SCALARS * i;

    // one inversion for y and all u_j: prefix products then back-substitution
    sc u[16], pre[17];
    pre[0] = y;
    for (u32 j = 0; j < lg; j++) { u[j] = load_scalar_words(ch + 32 + 8 * j); pre[j + 1] = sc_mul(pre[j], u[j]); }
    sc inv = sc_invert(pre[lg]);
    sc allinv = sc_one();
    sc rho_c = sc_mul(rho, c);
    for (int j = (int)lg - 1; j >= 0; j--) {
        sc u_inv = sc_mul(inv, pre[j]);          // 1/u_j
        inv = sc_mul(inv, u[j]);
        allinv = sc_mul(allinv, u_inv);
        sc u2 = sc_mul(u[j], u[j]), ui2 = sc_mul(u_inv, u_inv);
        store_scalar_words(s + 8 * (4 + 2 * j), sc_mul(rho, u2));            // L_j
        store_scalar_words(s + 8 * (4 + 2 * j + 1), sc_mul(rho, ui2));       // R_j
        store_sc_limbs(ctx + 9 * (RPC_U2 + (lg - 1 - j)), u2);               // bit (lg-1-j) of the index selects u_j^2
    }
    sc y_inv = inv;                               // after removing every u_j
    sc zz = sc_mul(z, z);
    store_scalar_words(s, rho);                                  // A
    store_scalar_words(s + 8, sc_mul(rho, x));                   // S
    store_scalar_words(s + 16, sc_mul(rho_c, x));                // T_1
    store_scalar_words(s + 24, sc_mul(sc_mul(rho_c, x), x));     // T_2
    {
        sc v = sc_mul(rho_c, zz);
        for (u32 k = 0; k < A + T; k++) { store_scalar_words(s + 8 * (4 + 2 * lg + k), v); v = sc_mul(v, z); }
    }
    // delta(y, z) = (z - zz) * sum_{i<nm} y^i - z^3 * (2^64 - 1) * sum_{j<m} z^j, sums as products of (1 + q^(2^k))
    sc sum_y = sc_one(), sum_z = sc_one();
    {
        sc q = y;
        for (u32 k = 0; k < lg; k++) { sum_y = sc_mul(sum_y, sc_add(sc_one(), q)); q = sc_mul(q, q); }
        q = z;
        for (u32 k = 0; k + 6 < lg; k++) { sum_z = sc_mul(sum_z, sc_add(sc_one(), q)); q = sc_mul(q, q); }
    }
    sc sum_2 = sc_from_u64(0xffffffffffffffffULL);
    sc delta = sc_sub(sc_mul(sc_sub(z, zz), sum_y), sc_mul(sc_mul(sc_mul(zz, z), sum_2), sum_z));
    sc b_scalar = sc_add(sc_mul(w, sc_sub(t_x, sc_mul(a, b))), sc_mul(c, sc_sub(delta, t_x)));
    sc bb_scalar = sc_neg(sc_add(e_bl, sc_mul(c, t_x_bl)));
    store_sc_limbs(gh + 18ull * i, sc_norm(sc_mul(rho, b_scalar)));
    store_sc_limbs(gh + 18ull * i + 9, sc_norm(sc_mul(rho, bb_scalar)));
    // context
    sc rho_z = sc_mul(rho, z);
    store_sc_limbs(ctx + 9 * RPC_RHO_Z, rho_z);
    store_sc_limbs(ctx + 9 * RPC_NEG_RHO_Z, sc_neg(rho_z));
    store_sc_limbs(ctx + 9 * RPC_COEF_G, sc_neg(sc_mul(sc_mul(rho, a), allinv)));
    store_sc_limbs(ctx + 9 * RPC_COEF_H2, sc_mul(rho, zz));
    store_sc_limbs(ctx + 9 * RPC_COEF_H3, sc_neg(sc_mul(sc_mul(rho, b), allinv)));
    {
        sc q = y_inv;                             // y^-(2^k)
        for (u32 k = 0; k < lg; k++) { store_sc_limbs(ctx + 9 * (RPC_Y2INV + k), q); q = sc_mul(q, q); }
        q = sc_add(y_inv, y_inv);                 // (2/y)^(2^k)
        for (u32 k = 0; k < 6; k++) { store_sc_limbs(ctx + 9 * (RPC_E + k), q); q = sc_mul(q, q); }
        // (z * y^-64)^(2^k): y^-64 = Y2INV[6]
        sc y64 = y_inv;
        for (u32 k = 0; k < 6; k++) y64 = sc_mul(y64, y64);
        q = sc_mul(z, y64);
        for (u32 k = 0; k + 6 < lg; k++) { store_sc_limbs(ctx + 9 * (RPC_F + k), q); q = sc_mul(q, q); }
    }
}

// thread per (row, i): the scalars of G_i and H_i contributed by one transaction of the lg-group
__global__ void __launch_bounds__(256, 2) k_rp_matrix(const u32 *__restrict__ group_txs, u32 row0, u32 n_rows, u32 lg, const u32 *__restrict__ rp_ctx,
                                                      const u32 *__restrict__ rp_err, u32 *__restrict__ mat) {
    u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 nm = 1u << lg;
    if (g >= (u64)n_rows * nm) return;
    u32 r = (u32)(g >> lg), idx = (u32)g & (nm - 1);
    u32 tx = group_txs[row0 + r];
    u32 *out_g = mat + 9ull * ((u64)r * 2 * nm + idx), *out_h = mat + 9ull * ((u64)r * 2 * nm + nm + idx);
    if (rp_err[tx]) { store_sc_limbs(out_g, sc_zero()); store_sc_limbs(out_h, sc_zero()); return; }
    const u32 *ctx = rp_ctx + 9ull * RPC_SCALARS * tx;
    sc sprod = load_sc_limbs(ctx + 9 * RPC_COEF_G), tprod = load_sc_limbs(ctx + 9 * RPC_COEF_H3), eprod = load_sc_limbs(ctx + 9 * RPC_COEF_H2);
    for (u32 b = 0; b < lg; b++) {
        sc u2 = load_sc_limbs(ctx + 9 * (RPC_U2 + b));
        if ((idx >> b) & 1) {
            sprod = sc_mul(sprod, u2);
            tprod = sc_mul(tprod, load_sc_limbs(ctx + 9 * (RPC_Y2INV + b)));
            eprod = sc_mul(eprod, load_sc_limbs(ctx + 9 * ((b < 6) ? (RPC_E + b) : (RPC_F + b - 6))));
        } else {
            tprod = sc_mul(tprod, u2);
        }
    }
    store_sc_limbs(out_g, sc_add(load_sc_limbs(ctx + 9 * RPC_NEG_RHO_Z), sprod));
    store_sc_limbs(out_h, sc_norm(sc_add_lazy(sc_add(load_sc_limbs(ctx + 9 * RPC_RHO_Z), eprod), tprod)));
}
// thread per column: acc[col] += sum over rows
__global__ void __launch_bounds__(256) k_rp_colsum(const u32 *__restrict__ mat, u32 n_rows, u32 lg, u32 nm_cap, u32 *__restrict__ acc /* 2*nm_cap scalars */) {
    u32 col = blockIdx.x * blockDim.x + threadIdx.x;
    u32 nm = 1u << lg;
    if (col >= 2 * nm) return;
    u32 dst = col < nm ? col : nm_cap + (col - nm);
    sc a = load_sc_limbs(acc + 9ull * dst); u32 cnt = 0;
    for (u32 r = 0; r < n_rows; r++) {
        a = sc_add(a, load_sc_limbs(mat + 9ull * ((u64)r * 2 * nm + col)));
        if (++cnt == 16) { a = sc_reduce_mont(a); cnt = 0; }
    }
    store_sc_limbs(acc + 9ull * dst, sc_norm(sc_reduce_mont(a)));
}
// static terms: scalars of B, H (sums of the per-proof parts) and of G_i, H_i (column sums)
__global__ void __launch_bounds__(256) k_static_terms(const u32 *__restrict__ gh_sums /* n_parts x 2 */, u32 n_parts, const u32 *__restrict__ acc, u32 nm_cap,
                                                      u32 static_term_off, u32 *__restrict__ scalars, u32 *__restrict__ term_point) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 + 2 * nm_cap) return;
    sc v;
    if (g < 2) {
        v = sc_zero();
        for (u32 k = 0; k < n_parts; k++) v = sc_reduce_mont(sc_add(v, load_sc_limbs(gh_sums + 9ull * (2 * k + g))));
    } else v = load_sc_limbs(acc + 9ull * (g - 2));
    store_scalar_words(scalars + 8ull * (static_term_off + g), v);
    term_point[static_term_off + g] = g;
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

__global__ void __launch_bounds__(64) k_schnorr(const u8 *__restrict__ bytes, const dev_sig *__restrict__ sigs, u32 n, const u32 *__restrict__ pts,
                                                const u8 *__restrict__ ok, const u32 *__restrict__ htab, const u8 *__restrict__ ms_hash,
                                                u32 *__restrict__ tx_err) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dev_sig sg = sigs[i];
    if (!ok[sg.key_pt]) { if (sg.key_err_key) raise_err(tx_err, sg.tx, sg.key_err_key); return; }
    u32 sw[8], ew[8];
    load_words_unaligned(sw, bytes + sg.sig_off); load_words_unaligned(ew, bytes + sg.sig_off + 32);
    // s * H from the fixed-base table (32 windows of 8 bits)
    ge R = ge_identity();
#pragma unroll 1
    for (u32 w = 0; w < 32; w++) {
        u32 b = (sw[w >> 2] >> ((w & 3) * 8)) & 255;
        if (b) R = ge_madd(R, ge_niels_load(htab + 32ull * (w * 256 + b)));
    }
    // e * P with 4-bit windows
    ge P = ge_from_niels_affine(ge_niels_load(pts + 32ull * sg.key_pt));
    ge tbl[16];
    tbl[0] = ge_identity(); tbl[1] = P;
#pragma unroll 1
    for (int k = 2; k < 16; k++) tbl[k] = ge_add(tbl[k - 1], P);
    ge Q = ge_identity();
#pragma unroll 1
    for (int k = 63; k >= 0; k--) {
        Q = ge_double(ge_double(ge_double(ge_double(Q))));
        u32 nib = (ew[k >> 3] >> ((k & 7) * 4)) & 15;
        Q = ge_add(Q, tbl[nib]);
    }
    R = ge_sub(R, Q);
    u32 rw[8]; ristretto_encode_words(rw, R);
    u8 rb[32];
    for (int k = 0; k < 8; k++) { rb[4 * k] = (u8)rw[k]; rb[4 * k + 1] = (u8)(rw[k] >> 8); rb[4 * k + 2] = (u8)(rw[k] >> 16); rb[4 * k + 3] = (u8)(rw[k] >> 24); }
    sponge sp; sp.init(72, 0x06);
    sp.absorb(bytes + sg.key_off, 32);
    if (sg.hash_idx != 0xffffffffu) sp.absorb(ms_hash + 32ull * sg.hash_idx, 32);
    else sp.absorb(bytes + sg.msg_off, sg.msg_len);
    sp.absorb(rb, 32);
    u8 h[64]; sp.squeeze(h, 64);
    u32 cw[8]; sc_to_words(cw, sc_from_wide_bytes(h));
    u32 diff = 0;
    for (int k = 0; k < 8; k++) diff |= cw[k] ^ ew[k];
    if (diff) raise_err(tx_err, sg.tx, sg.err_key);
}

// ---- error precedence --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_point_errors(const dev_tx *__restrict__ txs, const dev_cm *__restrict__ cms, const dev_xfer *__restrict__ xfers, u32 n_txs,
                                                      const u8 *__restrict__ ok, u32 *__restrict__ tx_err, u32 *__restrict__ rp_err) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_txs) return;
    const dev_tx d = txs[i];
    u32 key = ERRKEY_NONE;
    auto raise = [&](u32 k) { if (k < key) key = k; };
    for (u32 k = 0; k < d.n_transfers; k++) {
        const dev_xfer x = xfers[d.xfer_first + k];
        if (!ok[x.pt] || !ok[x.pt + 1] || !ok[x.pt + 2]) raise(errkey(ST_DECOMPRESS, 0, 0, XHE_ERR_DECOMPRESSION));
        if (!ok[x.dest_pt]) raise(errkey(ST_TRANSFER, k, 0, XHE_ERR_DECOMPRESSION));
        if (x.first_touch && !x.state_err && (!ok[x.slot_pt] || !ok[x.slot_pt + 1])) raise(errkey(ST_TRANSFER, k, 2, XHE_ERR_DECOMPRESSION));
        if (!ok[x.pt + 3] || !ok[x.pt + 4] || !ok[x.pt + 5]) raise(errkey(ST_TRANSFER, k, 4, XHE_ERR_CT_VALIDITY));
    }
    for (u32 k = 0; k < d.n_commitments; k++) {
        const dev_cm c = cms[d.cm_first + k];
        if (!ok[c.pt]) raise(errkey(ST_DECOMPRESS, 0, 0, XHE_ERR_DECOMPRESSION));
        if (c.first_touch && (!ok[c.slot_pt] || !ok[c.slot_pt + 1])) raise(errkey(ST_COMMITMENT, k, 1, XHE_ERR_DECOMPRESSION));
        if (!ok[c.pt + 1] || !ok[c.pt + 2] || !ok[c.pt + 3]) raise(errkey(ST_COMMITMENT, k, 3, XHE_ERR_COMMITMENT_EQ));
    }
    if (!ok[d.src_pt]) raise(errkey(ST_DECOMPRESS, 0, 0, XHE_ERR_DECOMPRESSION));
    if (key != ERRKEY_NONE) atomicMin(&tx_err[i], key);
    bool rp_bad = false;
    for (u32 k = 0; k < 4u + 2u * d.lg; k++) rp_bad |= !ok[d.rp_pt + k];
    if (rp_bad) rp_err[i] = 1;
}
// out[0..1]: (first failing tx << 32 | key) as u64 min; out[2]: first tx with a range-proof failure
__global__ void __launch_bounds__(256) k_verdict(const u32 *__restrict__ tx_err, const u32 *__restrict__ rp_err, u32 n_txs, unsigned long long *__restrict__ first_err,
                                                 u32 *__restrict__ first_rp) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_txs) return;
    u32 k = tx_err[i];
    if (k != ERRKEY_NONE) atomicMin(first_err, ((unsigned long long)i << 32) | k);
    if (rp_err[i]) atomicMin(first_rp, i);
}

}  // namespace xhe
