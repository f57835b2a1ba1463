// Descriptor expansion: the per-transaction, per-transfer and per-commitment descriptors, the decode lists (one entry per
// point), the ledger operations, the signature descriptors, the transcript segment tables and the lists of terms that share a
// point, written on the device from the compact records the host's ordered walk leaves (expand_tx / expand_xf / expand_cm,
// xhe_verify_types.h).  The host touches every record once (parse, ledger look-ups, the walk); everything that is one entry
// per point, term or operation — about 230 MB for 100 000 transactions of 8 transfers — is produced here instead of being
// written by host threads, staged and copied.  The control checks whose outcome the walk hands over as flags are those of
// Transaction::pre_verify (/root/reference/src/tx/verify.rs:210-225 nonce, :161-199 assets, :259-292 multisig, :301-303 and
// :354-360 ledger look-ups); the keys keep the reference's error precedence (xhe_verify_types.h).
#include "xhe_dev_common.h"
#include "xhe_kernels.h"

namespace xhe {

__global__ void __launch_bounds__(128) k_expand_descriptors(expand_args a) {
    constexpr u32 NONE = 0xffffffffu;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_txs) return;
    const expand_tx t = a.txs[i];
    const u32 T = t.n_transfers, A = t.n_commitments, lg = t.lg;
    const u64 rec = t.rec;
    dev_tx d;
    d.rec = rec; d.pre_off = t.pre_off; d.pre_len = t.pre_len; d.payload_off = t.payload_off; d.rp_off = t.rp_off; d.cm_off = t.cm_off;
    d.ms_off = t.ms_off; d.sig_off = t.sig_off;
    d.xfer_first = t.xfer_first; d.cm_first = t.cm_first;
    d.n_transfers = (u16)T; d.n_commitments = t.n_commitments; d.n_multisig = t.n_multisig; d.n_items = t.n_items;
    d.type = t.type; d.lg = t.lg; d.version = t.version; d.flags = t.flags; d.m = t.m; d.fee = t.fee; d.nonce = t.nonce;
    d.chal_off = t.chal_off; d.rp_term_off = t.rp_term_off; d.hash_idx = t.hash_idx; d.rp_row = t.rp_row; d.tmpl_off = t.tmpl_off; d.seg_off = t.seg_off;
    d.reserved = 0;
    const expand_xf *xf = a.xfs + t.xfer_first;
    const expand_cm *cm = a.cms + t.cm_first;
    {
        u32 *sg = a.segs + t.seg_off;
        *sg++ = t.pre_off; *sg++ = t.rp_off;
        for (u32 k = 0; k < A; k++) *sg++ = t.cm_off + 256 * k;
        for (u32 k = 0; k < T; k++) { *sg++ = xf[k].off; *sg++ = xf[k].off + xf[k].len - 160; }
    }
    u32 key = t.step_err, akey = ERRKEY_NONE;
    auto raise = [&](u32 k2) { if (k2 < key) key = k2; };
    auto raise_state = [&](u32 k2) { raise(k2); if (k2 < akey) akey = k2; };
    // the block of points: range proof | source key if the ledger does not know it | per transfer C, D_s, D_r, Y_0, Y_1, Y_2 |
    // per commitment C_new, Y_0, Y_1, Y_2 | destination keys the ledger does not know
    u32 pos = t.blk, ei = t.early_base, bi = t.bulk_base;
    auto put = [&](u64 off, u32 early) {
        a.decs[pos].off = off;
        if (early == 1) a.early_decs[ei++] = pos; else if (early == 2) a.key_decs[t.key_base] = pos; else a.bulk_decs[bi++] = pos;
        return a.n_static + pos++;
    };
    d.rp_pt = put(rec + t.rp_off, 0);
    for (u32 k = 1; k < 4; k++) put(rec + t.rp_off + 32 * k, 0);
    for (u32 k = 0; k < 2 * lg; k++) put(rec + t.rp_off + 224 + 32 * k, 0);
    // source account, nonce
    if (!t.src_known) { raise_state(errkey(ST_NONCE, 0, 0, XHE_ERR_STATE)); d.src_pt = put(rec + t.pre_off + 1, 2); }
    else {
        if (!t.nonce_ok) raise(errkey(ST_NONCE, 0, 1, XHE_ERR_INVALID_NONCE));
        d.src_pt = a.n_static + t.src_pos;
    }
    if (!t.assets_ok) raise(errkey(ST_ASSETS, 0, 0, XHE_ERR_FORMAT));
    // main signature over pre-image || multisig entries, then the co-signatures
    {
        u32 si = t.sig_base;
        dev_sig s; s.tx = i; s.key_pt = d.src_pt; s.key_off = rec + t.pre_off + 1; s.msg_off = rec + t.pre_off; s.msg_len = t.pre_len + 65u * t.n_multisig;
        s.hash_idx = NONE; s.sig_off = rec + t.sig_off; s.err_key = errkey(ST_SIGNATURE, 0, 0, XHE_ERR_SIGNATURE); s.key_err_key = 0;
        s.key_slot = t.src_keyslot;
        a.sigs[si++] = s;
        for (u32 e = t.ms_first; e < t.ms_first + t.ms_count; e++) {
            const expand_ms me = a.mss[e];
            dev_sig c; c.tx = i; c.key_pt = a.n_static + me.key_pos; c.key_off = me.key_off; c.msg_off = 0; c.msg_len = 32; c.hash_idx = t.hash_idx;
            c.sig_off = rec + t.ms_off + 65ull * me.k + 1; c.err_key = errkey(ST_MULTISIG, me.k, 3, XHE_ERR_SIGNATURE);
            c.key_err_key = errkey(ST_MULTISIG, me.k, 2, XHE_ERR_DECOMPRESSION);
            c.key_slot = me.key_slot;
            a.sigs[si++] = c;
        }
    }
    // transfers' points: C, D_s, D_r, Y_0, Y_1, Y_2
    const u32 xf_pt0 = a.n_static + pos;
    for (u32 k = 0; k < T; k++) {
        const u64 o = rec + xf[k].off, po = o + xf[k].len - 160;
        put(o + 64, 1); put(o + 96, 1); put(o + 128, 1);
        put(po, 0); put(po + 32, 0); put(po + 64, 0);
    }
    // commitments: the sender's slot loses every transfer of the asset, the plain amount (burn, contract deposit) and, with the
    // emission of the sender's new ciphertext, the fee (get_sender_output_ct, verify.rs:107-144)
    for (u32 k = 0; k < A; k++) {
        const u32 ci = t.cm_first + k;
        const expand_cm ec = cm[k];
        dev_cm c; c.tx = i; c.idx = k; c.off = rec + t.cm_off + 256ull * k;
        c.pt = put(c.off + 32, 0); put(c.off + 64, 0); put(c.off + 96, 0); put(c.off + 128, 0);
        c.term_off = t.cm_term + 7 * k;
        c.src_ct_pt = a.emit_base + 2 * ci; c.slot_pt = 0; c.first_touch = 0;
        // C_new is the point of the proof's term 5 and of the range proof's value commitment k; P_source that of term 0
        a.merge_pairs[2ull * (ci + t.xfer_first)] = c.term_off + 5; a.merge_pairs[2ull * (ci + t.xfer_first) + 1] = t.rp_term_off + 4 + 2 * lg + k;
        if (t.src_key_term != NONE) a.key_terms[t.src_key_term + k] = c.term_off;
        if (ec.slot == NONE) raise_state(errkey(ST_COMMITMENT, k, 0, XHE_ERR_STATE));
        else {
            c.slot_pt = a.n_static + ec.slot_pos; c.first_touch = ec.fresh;
            u32 o = ec.op_pos;
            for (u32 x = 0; x < T; x++)
                if (xf[x].cm == k) {
                    dev_op op; op.slot = ec.slot; op.kind = 1; op.amount = 0; op.emit = NONE; op.first = 0;
                    op.c_pt = xf_pt0 + 6 * x; op.d_pt = op.c_pt + 1;      // (C, D_sender)
                    a.ops[o++] = op;
                }
            if (ec.amount) { dev_op op; op.slot = ec.slot; op.kind = 2; op.c_pt = 0; op.d_pt = 0; op.amount = ec.amount; op.emit = NONE; op.first = 0; a.ops[o++] = op; }
            dev_op fin; fin.slot = ec.slot; fin.kind = 2; fin.c_pt = 0; fin.d_pt = 0; fin.amount = ec.native ? t.fee : 0; fin.emit = ci; fin.first = 0;
            a.ops[o++] = fin;
        }
        a.out_cms[ci] = c;
    }
    // transfers
    for (u32 k = 0; k < T; k++) {
        const u32 xi = t.xfer_first + k;
        const expand_xf ex = xf[k];
        dev_xfer x; x.tx = i; x.idx = k; x.off = rec + ex.off; x.proof_off = x.off + ex.len - 160;
        x.pt = xf_pt0 + 6 * k;
        x.term_off = t.xf_term + 8 * k;
        x.slot_pt = 0; x.first_touch = 0; x.state_err = 0;
        // C is the point of the proof's term 0 and of value commitment A + k; P_source that of term 5, P_dest of term 2
        a.merge_pairs[2ull * (t.cm_first + A + xi)] = x.term_off; a.merge_pairs[2ull * (t.cm_first + A + xi) + 1] = t.rp_term_off + 4 + 2 * lg + A + k;
        if (t.src_key_term != NONE) a.key_terms[t.src_key_term + A + k] = x.term_off + 5;
        if (ex.key_term != NONE) a.key_terms[ex.key_term] = x.term_off + 2;
        if (!ex.has_account) {
            x.dest_pt = put(x.off + 32, 0);
            x.state_err = 1;
            raise_state(errkey(ST_TRANSFER, k, 1, XHE_ERR_STATE));
        } else {
            x.dest_pt = a.n_static + ex.dest_pos;
            if (ex.slot == NONE) { x.state_err = 1; raise_state(errkey(ST_TRANSFER, k, 1, XHE_ERR_STATE)); }
            else {
                x.slot_pt = a.n_static + ex.slot_pos; x.first_touch = ex.fresh;
                dev_op op; op.slot = ex.slot; op.kind = 0; op.c_pt = x.pt; op.d_pt = x.pt + 2; op.amount = 0; op.emit = NONE; op.first = 0;
                a.ops[ex.op_pos] = op;
            }
        }
        a.out_xfers[xi] = x;
    }
    a.out_txs[i] = d;
    a.tx_err[i] = key; a.apply_err[i] = akey;
    // Scalar::from_canonical_bytes of the reference's deserialisation: the proofs' responses, the signatures, the range proof's
    // t_x, t_x_blinding, e_blinding and a, b (the same list as record_scalars_canonical on the host)
    {
        bool canonical = true;
        auto check = [&](u64 off, u32 count) {
            for (u32 s = 0; s < count; s++) { u32 w[8]; load_words_unaligned(w, a.bytes + off + 32ull * s); canonical = canonical && sc_words_canonical(w); }
        };
        for (u32 k = 0; k < T; k++) check(rec + xf[k].off + xf[k].len - 64, 2);
        for (u32 k = 0; k < A; k++) check(rec + t.cm_off + 256ull * k + 160, 3);
        check(rec + t.sig_off, 2);
        for (u32 k = 0; k < t.n_multisig; k++) check(rec + t.ms_off + 65ull * k + 1, 2);
        check(rec + t.rp_off + 128, 3);
        check(rec + t.rp_off + 224 + 64ull * lg, 2);
        if (!canonical) atomicMin(a.noncanonical_first, i);
    }
}

// the points shared between transactions (behind the transactions' blocks in decs[] and in the two decode lists they belong to)
// and every slot's initial operation
__global__ void __launch_bounds__(256) k_expand_shared(expand_args a) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < a.n_shared) a.decs[a.blk_end + g] = a.shared_decs[g];
    if (g < a.n_shared_early) a.early_decs[a.early_end + g] = a.shared_early[g];
    if (g < a.n_shared_key) a.key_decs[a.key_end + g] = a.shared_key[g];
    if (g < a.n_slots) {
        dev_op init; init.slot = g; init.kind = 0; init.c_pt = a.n_static + a.slot_pos[g]; init.d_pt = init.c_pt + 1; init.amount = 0; init.emit = 0xffffffffu; init.first = 1;
        a.ops[a.slot_base[g]] = init;
        a.slot_last_op[g] = a.slot_base[g + 1] - 1;
    }
}

void launch_expand(hipStream_t st, const expand_args &a) {
    if (a.n_txs) hipLaunchKernelGGL(k_expand_descriptors, dim3(grid_for(a.n_txs, 128)), dim3(128), 0, st, a);
    const u32 most = std::max(std::max(a.n_shared, a.n_shared_early), std::max(a.n_shared_key, a.n_slots));
    if (most) hipLaunchKernelGGL(k_expand_shared, dim3(grid_for(most, 256)), dim3(256), 0, st, a);
}

}  // namespace xhe
