// Host ingest of a batch: parses the records, runs the checks of Transaction::pre_verify that are pure
// control logic (nonce, asset/commitment consistency, multisig structure, ledger look-ups;
// /root/reference/src/tx/verify.rs:210-225,259-292,301-303,354-360,401-418) and lays out the flat tables
// of xhe_verify_types.h for the kernels.  No group or field arithmetic happens here.
//
// Transactions are walked in order against a copy of the ledger's key structure (which accounts, which
// (account, asset) slots, which multisig configurations exist); a MultiSig transaction updates the
// configuration seen by the transactions after it, as in the reference's sequential pre_verify.  Each
// transaction's checks assume the transactions before it passed: the verdict takes the failure of the
// lowest transaction index, for which that assumption holds.
#pragma once
#include <string.h>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/xhe.h"
#include "xhe_record.h"
#include "xhe_transcript_template.h"
#include "xhe_verify_types.h"

namespace xhe {

// fixed-size byte keys for the ledger look-ups (public keys are uniformly distributed: their first bytes are the hash)
template <size_t N> struct byte_key {
    u8 b[N];
    bool operator==(const byte_key &o) const { return !memcmp(b, o.b, N); }
};
template <size_t N> struct byte_key_hash {
    size_t operator()(const byte_key<N> &k) const { u64 h, t; memcpy(&h, k.b, 8); memcpy(&t, k.b + N - 8, 8); return (size_t)(h ^ (t * 0x9e3779b97f4a7c15ull)); }
};
typedef byte_key<32> key32;
typedef byte_key<36> key36;          // asset(32) || account index(4)
static inline key32 make_key32(const u8 *p) { key32 k; memcpy(k.b, p, 32); return k; }

struct host_batch {
    u64 n_txs = 0;
    u64 bytes_len = 0, rec_bytes = 0, n_signer_keys = 0;      // device byte buffer: records || account keys || balance cts || multisig signers || spare
    u64 keys_base = 0, cts_base = 0, signers_base = 0;
    std::vector<dev_tx> txs;
    std::vector<dev_cm> cms;
    std::vector<dev_xfer> xfers;
    std::vector<dev_dec> decs;             // decode list; point index = n_static + position
    std::vector<u32> early_decs;           // positions in decs[] of the points the ledger scan reads (ledger and transfer ciphertexts)
    std::vector<u32> key_decs;             // positions in decs[] of the public keys the signatures are checked against
    std::vector<u32> bulk_decs;            // positions in decs[] of every other point
    std::vector<u32> sig_keys;             // point index of every distinct key a signature of the batch is checked against
    std::vector<dev_op> ops;               // sorted by slot
    std::vector<dev_sig> sigs;
    std::vector<u32> tx_err;               // host-detected precedence keys (ERRKEY_NONE if none)
    std::vector<u32> apply_err;            // the subset apply_without_verify knows: unknown account / asset only
    std::vector<u32> rp_dec_tx;            // per decode entry: tx whose range proof owns the point, 0xffffffff otherwise
    std::vector<u32> slot_of_used;         // used slot -> ledger balance index
    std::vector<u32> slot_last_op;         // used slot -> index of its last op in ops[]
    std::vector<u32> slot_first_op;
    std::vector<u32> ms_hash_tx;           // transactions whose pre-image BLAKE3 digest is needed
    std::vector<std::vector<u32>> lg_groups;   // transactions by lg (index = lg)
    std::vector<int64_t> multisig_set_by;  // per account
    std::vector<u32> tmpl;                 // transcript templates of the batch's shapes
    std::vector<u32> segs;                 // per-transaction segment tables
    std::vector<u32> serial_txs;           // transactions whose transcript runs in the serial kernel
    std::vector<u32> tr_list;              // the others, sorted by template, each template padded to whole tiles of 64 (0xffffffff)
    std::vector<transcript_tile> tr_tiles; // one per 64 entries of tr_list
    transcript_plan tr_plan;
    u32 n_static = 0, nm_cap = 0;
    u32 n_points = 0;                      // static + decoded + emitted
    u32 emit_base = 0, n_emit = 0;         // emitted ciphertext points (2 per commitment)
    u32 n_chal = 0;
    u32 n_terms = 0, sigma_terms = 0, static_term_off = 0;
    xhe_verdict structural{1, XHE_OK, -1};
};

// returns false when a record is malformed (hb.structural says which)
static inline bool ingest_batch(host_batch &hb, const u8 *records, const u64 *offsets, u64 n, const xhe_ledger_view &lv) {
    hb.n_txs = n;
    std::vector<tx_record> recs(n);
    u32 max_lg = 6;
    for (u64 i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i] || !record_parse(recs[i], records + offsets[i], offsets[i + 1] - offsets[i])) {
            hb.structural = {0, XHE_ERR_MALFORMED, (int64_t)i};
            return false;
        }
        const tx_record &t = recs[i];
        // a HashMap cannot hold one asset twice (src/tx/mod.rs:78)
        if (t.type == 3)
            for (u32 a = 0; a < t.n_items; a++)
                for (u32 b = a + 1; b < t.n_items; b++)
                    if (!memcmp(t.payload() + 32 + 40 * a, t.payload() + 32 + 40 * b, 32)) { hb.structural = {0, XHE_ERR_MALFORMED, (int64_t)i}; return false; }
        if (t.lg() > max_lg) max_lg = t.lg();
    }
    if (max_lg > 15) max_lg = 15;      // party capacity 512 (src/proofs.rs:20); larger proofs fail in the range-proof stage
    hb.nm_cap = 1u << max_lg;
    hb.n_static = 2 + 2 * hb.nm_cap;

    // ---- byte buffer ----
    u64 rec_bytes = offsets[n];
    hb.keys_base = (rec_bytes + 63) & ~63ull;
    hb.cts_base = hb.keys_base + 32 * lv.n_accounts;
    u64 n_signers = lv.n_multisig ? lv.multisig_offsets[lv.n_multisig] : 0;
    hb.signers_base = hb.cts_base + 64 * lv.n_balances;
    // the device's byte buffer: records | account keys | ledger ciphertexts | multisig signer keys | 64 spare bytes.  It is
    // assembled from the caller's arrays when the batch is uploaded (byte_pieces); no host copy is kept.
    hb.rec_bytes = rec_bytes; hb.n_signer_keys = n_signers;
    hb.bytes_len = hb.signers_base + 32 * n_signers + 64;

    // ---- ledger key structure ----
    std::unordered_map<key32, u32, byte_key_hash<32>> account_of;
    account_of.reserve(lv.n_accounts * 2 + 16);
    for (u64 i = 0; i < lv.n_accounts; i++) account_of.emplace(make_key32(lv.account_keys + 32 * i), (u32)i);
    std::unordered_map<key36, u32, byte_key_hash<36>> slot_of;
    slot_of.reserve(lv.n_balances * 2 + 16);
    auto slot_key = [](const u8 *asset, u32 account) { key36 k; memcpy(k.b, asset, 32); memcpy(k.b + 32, &account, 4); return k; };
    for (u64 i = 0; i < lv.n_balances; i++) slot_of.emplace(slot_key(lv.balance_assets + 32 * i, lv.balance_account[i]), (u32)i);
    struct ms_cfg { std::vector<u64> signer_off; u32 threshold; };
    std::unordered_map<u32, ms_cfg> ms_of;
    for (u64 i = 0; i < lv.n_multisig; i++) {
        ms_cfg c; c.threshold = lv.multisig_threshold[i];
        for (u32 j = lv.multisig_offsets[i]; j < lv.multisig_offsets[i + 1]; j++) c.signer_off.push_back(hb.signers_base + 32ull * j);
        ms_of[lv.multisig_account[i]] = c;
    }
    hb.multisig_set_by.assign(lv.n_accounts, -1);

    // ---- point index allocation ----
    auto new_point = [&](u64 off, u32 rp_tx, u32 early = 0) -> u32 {          // early: 1 = ledger scan input, 2 = key
        if (early == 1) hb.early_decs.push_back((u32)hb.decs.size());
        if (early == 2) hb.key_decs.push_back((u32)hb.decs.size());
        if (early == 0) hb.bulk_decs.push_back((u32)hb.decs.size());
        hb.decs.push_back({off, early, 0u}); hb.rp_dec_tx.push_back(rp_tx);
        return hb.n_static + (u32)hb.decs.size() - 1;
    };
    std::unordered_map<u32, u32> sig_key_slot;        // point idx of a signing key -> position in sig_keys
    auto key_slot_of = [&](u32 pt) -> u32 {
        auto it = sig_key_slot.find(pt);
        if (it != sig_key_slot.end()) return it->second;
        u32 slot = (u32)hb.sig_keys.size();
        hb.sig_keys.push_back(pt); sig_key_slot.emplace(pt, slot);
        return slot;
    };
    std::unordered_map<u32, u32> account_pt;          // account idx -> point idx
    std::unordered_map<u64, u32> signer_pt;           // byte offset of a signer key -> point idx
    std::unordered_map<u32, u32> used_slot;           // ledger balance idx -> used-slot idx
    std::vector<u32> slot_pt;                         // used-slot idx -> point idx of initial C (D = +1)
    std::vector<std::vector<dev_op>> slot_ops;        // per used slot, in order
    auto get_account_pt = [&](u32 a) { auto it = account_pt.find(a); if (it != account_pt.end()) return it->second; u32 p = new_point(hb.keys_base + 32ull * a, 0xffffffffu, 2); account_pt[a] = p; return p; };
    // returns used-slot idx; *fresh = true when this is the first touch
    auto touch_slot = [&](u32 bal_idx, bool *fresh) {
        auto it = used_slot.find(bal_idx);
        if (it != used_slot.end()) { *fresh = false; return it->second; }
        u32 u = (u32)slot_pt.size();
        used_slot[bal_idx] = u;
        u32 p = new_point(hb.cts_base + 64ull * bal_idx, 0xffffffffu, 1);
        new_point(hb.cts_base + 64ull * bal_idx + 32, 0xffffffffu, 1);
        slot_pt.push_back(p); hb.slot_of_used.push_back(bal_idx);
        slot_ops.emplace_back();
        dev_op init{}; init.slot = u; init.kind = 0; init.c_pt = p; init.d_pt = p + 1; init.emit = 0xffffffffu; init.first = 1;
        slot_ops[u].push_back(init);
        *fresh = true;
        return u;
    };

    hb.txs.resize(n); hb.tx_err.assign(n, ERRKEY_NONE); hb.apply_err.assign(n, ERRKEY_NONE);
    {
        // sizes are known from the parsed records up to the ledger's first-touch points: reserve once
        u64 n_xf = 0, n_cm = 0, n_pts = 0, n_ms = 0;
        for (u64 i = 0; i < n; i++) {
            u64 T = recs[i].transfer_off.size(), A = recs[i].n_commitments;
            n_xf += T; n_cm += A; n_ms += recs[i].n_multisig; n_pts += 4 + 2ull * recs[i].lg() + 6 * T + 4 * A + 1;
        }
        n_pts += 2 * n_xf + 2 * n_cm;                  // ledger ciphertexts touched for the first time, receiver keys
        hb.decs.reserve(n_pts); hb.rp_dec_tx.reserve(n_pts); hb.early_decs.reserve(5 * n_xf + 2 * n_cm); hb.key_decs.reserve(n + n_xf); hb.bulk_decs.reserve(n_pts);
        hb.cms.reserve(n_cm); hb.xfers.reserve(n_xf); hb.sigs.reserve(n + n_ms); hb.segs.reserve(2 * n + n_cm + 2 * n_xf);
    }
    std::unordered_map<transcript_shape, u32, transcript_shape_hash> tmpl_of;
    hb.lg_groups.assign(16, {});
    u32 term = 0, chal = 0;
    auto raise = [&](u64 i, u32 key) { if (key < hb.tx_err[i]) hb.tx_err[i] = key; };
    auto raise_state = [&](u64 i, u32 key) { raise(i, key); if (key < hb.apply_err[i]) hb.apply_err[i] = key; };

    // pass 1: sigma-proof terms first so that they are contiguous (the sigma-only multiscalar check uses them)
    std::vector<u32> cm_term(n), xf_term(n);
    for (u64 i = 0; i < n; i++) { cm_term[i] = term; term += 7 * recs[i].n_commitments; xf_term[i] = term; term += 8 * (u32)recs[i].transfer_off.size(); }
    hb.sigma_terms = term;

    for (u64 i = 0; i < n; i++) {
        const tx_record &t = recs[i];
        dev_tx &d = hb.txs[i];
        memset(&d, 0, sizeof d);
        u64 rec = offsets[i];
        d.rec = rec; d.pre_off = t.pre_off; d.pre_len = t.pre_len; d.payload_off = t.payload_off; d.rp_off = t.rp_off; d.cm_off = t.cm_off;
        d.ms_off = t.ms_off; d.sig_off = t.sig_off;
        d.n_transfers = (u16)t.transfer_off.size(); d.n_commitments = t.n_commitments; d.n_multisig = t.n_multisig; d.n_items = t.n_items;
        d.type = t.type; d.lg = (u8)t.lg(); d.version = t.version; d.flags = t.flags; d.m = t.m(); d.fee = t.fee; d.nonce = t.nonce;
        d.hash_idx = 0xffffffffu;
        d.xfer_first = (u32)hb.xfers.size(); d.cm_first = (u32)hb.cms.size();
        u32 A = t.n_commitments, T = d.n_transfers, lg = t.lg();
        d.chal_off = chal; chal += 2 * A + 2 * T + 4 + lg;
        d.rp_term_off = term; term += 4 + 2 * lg + A + T;
        if (lg <= 15) { d.rp_row = (u32)hb.lg_groups[lg].size(); hb.lg_groups[lg].push_back((u32)i); }
        {
            transcript_shape sh{t.type, A, T, t.n_items, d.m, lg};
            if (transcript_shape_supported(sh)) {
                auto it = tmpl_of.find(sh);
                d.tmpl_off = it != tmpl_of.end() ? it->second : (tmpl_of[sh] = build_transcript_template(hb.tmpl, sh));
                d.seg_off = (u32)hb.segs.size();
                transcript_segments(hb.segs, t);
            } else { d.tmpl_off = TT_NONE; hb.serial_txs.push_back((u32)i); }
        }

        // per-tx points: range proof, commitments, transfers
        d.rp_pt = new_point(rec + t.rp_off, (u32)i);
        for (u32 k = 1; k < 4; k++) new_point(rec + t.rp_off + 32 * k, (u32)i);
        for (u32 k = 0; k < 2 * lg; k++) new_point(rec + t.rp_off + 224 + 32 * k, (u32)i);

        // source account, nonce (verify.rs:210-221)
        auto acc_it = account_of.find(make_key32(t.source()));
        bool src_known = acc_it != account_of.end();
        u32 src_acc = src_known ? acc_it->second : 0;
        if (!src_known) raise_state(i, errkey(ST_NONCE, 0, 0, XHE_ERR_STATE));
        else if (lv.account_nonces[src_acc] != t.nonce) raise(i, errkey(ST_NONCE, 0, 1, XHE_ERR_INVALID_NONCE));
        d.src_pt = src_known ? get_account_pt(src_acc) : new_point(rec + t.pre_off + 1, 0xffffffffu, 2);
        if (!record_commitment_assets_ok(t)) raise(i, errkey(ST_ASSETS, 0, 0, XHE_ERR_FORMAT));

        // main signature over pre-image || multisig entries (verify.rs:253-256)
        {
            dev_sig s{}; s.tx = (u32)i; s.key_pt = d.src_pt; s.key_off = rec + t.pre_off + 1; s.msg_off = rec + t.pre_off; s.msg_len = t.pre_len + 65u * t.n_multisig;
            s.hash_idx = 0xffffffffu; s.sig_off = rec + t.sig_off; s.err_key = errkey(ST_SIGNATURE, 0, 0, XHE_ERR_SIGNATURE); s.key_err_key = 0;
            s.key_slot = key_slot_of(s.key_pt);
            hb.sigs.push_back(s);
        }
        // multisig (verify.rs:259-292)
        auto ms_it = src_known ? ms_of.find(src_acc) : ms_of.end();
        if (ms_it != ms_of.end()) {
            const ms_cfg &cfg = ms_it->second;
            if (!t.has_multisig() || t.n_multisig == 0 || t.n_multisig != cfg.threshold) raise(i, errkey(ST_MULTISIG, 0, 0, XHE_ERR_FORMAT));
            else {
                d.hash_idx = (u32)hb.ms_hash_tx.size(); hb.ms_hash_tx.push_back((u32)i);
                for (u32 k = 0; k < t.n_multisig; k++) {
                    u8 idx = t.ms_entry(k)[0];
                    for (u32 j = 0; j < t.n_multisig; j++) if (j != k && t.ms_entry(j)[0] == idx) raise(i, errkey(ST_MULTISIG, k, 1, XHE_ERR_FORMAT));
                    if (idx < cfg.signer_off.size()) {
                        u64 koff = cfg.signer_off[idx];
                        auto sp = signer_pt.find(koff);
                        u32 kp = sp != signer_pt.end() ? sp->second : (signer_pt[koff] = new_point(koff, 0xffffffffu, 2));
                        dev_sig s{}; s.tx = (u32)i; s.key_pt = kp; s.key_off = koff; s.msg_off = 0; s.msg_len = 32; s.hash_idx = d.hash_idx;
                        s.sig_off = rec + t.ms_off + 65ull * k + 1; s.err_key = errkey(ST_MULTISIG, k, 3, XHE_ERR_SIGNATURE);
                        s.key_err_key = errkey(ST_MULTISIG, k, 2, XHE_ERR_DECOMPRESSION);
                        s.key_slot = key_slot_of(kp);
                        hb.sigs.push_back(s);
                    }
                }
            }
        } else if (t.has_multisig()) raise(i, errkey(ST_MULTISIG, 0, 0, XHE_ERR_FORMAT));

        // transfers' points: C, D_s, D_r, Y_0, Y_1, Y_2
        std::vector<u32> xf_pt(T);
        for (u32 k = 0; k < T; k++) {
            u64 o = rec + t.transfer_off[k];
            u64 po = rec + t.transfer_off[k] + t.transfer_len[k] - 160;
            xf_pt[k] = new_point(o + 64, 0xffffffffu, 1); new_point(o + 96, 0xffffffffu, 1); new_point(o + 128, 0xffffffffu, 1);
            new_point(po, 0xffffffffu); new_point(po + 32, 0xffffffffu); new_point(po + 64, 0xffffffffu);
        }
        // commitments (verify.rs:296-341)
        for (u32 k = 0; k < A; k++) {
            dev_cm c{}; c.tx = (u32)i; c.idx = k; c.off = rec + t.cm_off + 256ull * k;
            c.pt = new_point(c.off + 32, 0xffffffffu); new_point(c.off + 64, 0xffffffffu); new_point(c.off + 96, 0xffffffffu); new_point(c.off + 128, 0xffffffffu);
            c.term_off = cm_term[i] + 7 * k;
            c.src_ct_pt = 0; c.slot_pt = 0; c.first_touch = 0;
            const u8 *asset = t.commitment(k);
            auto sl = src_known ? slot_of.find(slot_key(asset, src_acc)) : slot_of.end();
            u32 cm_index = (u32)hb.cms.size();
            if (sl == slot_of.end()) raise_state(i, errkey(ST_COMMITMENT, k, 0, XHE_ERR_STATE));
            else {
                bool fresh; u32 u = touch_slot(sl->second, &fresh);
                c.slot_pt = slot_pt[u]; c.first_touch = fresh ? 1 : 0;
                // get_sender_output_ct (verify.rs:107-144): transfers of this asset, burn / contract amount, fee
                for (u32 x = 0; x < T; x++)
                    if (!memcmp(asset, t.transfer(x), 32)) {
                        dev_op op{}; op.slot = u; op.kind = 1; op.emit = 0xffffffffu; op.first = 0;
                        op.c_pt = xf_pt[x]; op.d_pt = xf_pt[x] + 1;      // (C, D_sender)
                        slot_ops[u].push_back(op);
                    }
                u64 amount = 0;
                if (t.type == 1) { if (!memcmp(asset, t.payload(), 32)) amount = rec_be64(t.payload() + 32); }
                else if (t.type == 3) { for (u32 x = 0; x < t.n_items; x++) if (!memcmp(asset, t.payload() + 32 + 40 * x, 32)) amount = rec_be64(t.payload() + 64 + 40 * x); }
                if (amount) { dev_op op{}; op.slot = u; op.kind = 2; op.amount = amount; op.emit = 0xffffffffu; slot_ops[u].push_back(op); }
                dev_op fin{}; fin.slot = u; fin.kind = 2; fin.amount = rec_is_zero32(asset) ? t.fee : 0; fin.emit = cm_index;
                slot_ops[u].push_back(fin);
            }
            hb.cms.push_back(c);
        }
        // transfers (verify.rs:345-395)
        for (u32 k = 0; k < T; k++) {
            dev_xfer x{}; x.tx = (u32)i; x.idx = k; x.off = rec + t.transfer_off[k]; x.proof_off = rec + t.transfer_off[k] + t.transfer_len[k] - 160;
            x.pt = xf_pt[k];
            x.term_off = xf_term[i] + 8 * k;
            const u8 *asset = t.transfer(k), *dest = t.transfer(k) + 32;
            auto da = account_of.find(make_key32(dest));
            if (da == account_of.end()) {
                x.dest_pt = new_point(x.off + 32, 0xffffffffu);
                x.state_err = 1;
                raise_state(i, errkey(ST_TRANSFER, k, 1, XHE_ERR_STATE));
            } else {
                x.dest_pt = get_account_pt(da->second);
                auto sl = slot_of.find(slot_key(asset, da->second));
                if (sl == slot_of.end()) { x.state_err = 1; raise_state(i, errkey(ST_TRANSFER, k, 1, XHE_ERR_STATE)); }
                else {
                    bool fresh; u32 u = touch_slot(sl->second, &fresh);
                    x.slot_pt = slot_pt[u]; x.first_touch = fresh ? 1 : 0;
                    dev_op op{}; op.slot = u; op.kind = 0; op.c_pt = x.pt; op.d_pt = x.pt + 2; op.emit = 0xffffffffu;
                    slot_ops[u].push_back(op);
                }
            }
            hb.xfers.push_back(x);
        }
        // MultiSig payload rules and state update (verify.rs:401-428)
        if (t.type == 2) {
            if (!record_multisig_payload_ok(t)) raise(i, errkey(ST_TRANSFER, 0, 0, XHE_ERR_FORMAT));
            else if (src_known) {
                if (t.n_items == 0) ms_of.erase(src_acc);
                else {
                    ms_cfg c; c.threshold = t.payload()[0];
                    for (u32 k = 0; k < t.n_items; k++) c.signer_off.push_back(rec + t.payload_off + 1 + 32ull * k);
                    ms_of[src_acc] = c;
                }
                hb.multisig_set_by[src_acc] = (int64_t)i;
            }
        }
        if (lg > 15 || ((u64)64 * d.m != (1ull << lg))) { /* range-proof stage reports it */ }
    }
    // ---- transcript tiles: transactions grouped by template (first appearance order), 64 per tile ----
    {
        std::vector<u32> order;                       // template offsets in order of first appearance
        std::unordered_map<u32, std::vector<u32>> by_tmpl;
        for (u64 i = 0; i < n; i++) {
            u32 to = hb.txs[i].tmpl_off;
            if (to == TT_NONE) continue;
            auto it = by_tmpl.find(to);
            if (it == by_tmpl.end()) { order.push_back(to); it = by_tmpl.emplace(to, std::vector<u32>()).first; }
            it->second.push_back((u32)i);
        }
        transcript_pass cur{0, 0, 0};
        u64 cur_words = 0;
        const u64 cap_words = transcript_stream_cap_words();
        for (u32 to : order) {
            const std::vector<u32> &v = by_tmpl[to];
            const u32 n_blocks = hb.tmpl[to + TT_H_BLOCKS];
            const u64 tile_words = 21ull * 64 * n_blocks;
            for (size_t k = 0; k < v.size(); k += 64) {
                if (cur.n_tiles && cur_words + tile_words > cap_words) {
                    hb.tr_plan.passes.push_back(cur);
                    cur = {(u32)hb.tr_tiles.size(), 0, 0}; cur_words = 0;
                }
                hb.tr_tiles.push_back({cur_words, to, n_blocks});
                for (size_t j = k; j < k + 64; j++) hb.tr_list.push_back(j < v.size() ? v[j] : 0xffffffffu);
                cur.n_tiles++; cur.max_blocks = std::max(cur.max_blocks, n_blocks);
                cur_words += tile_words;
                hb.tr_plan.stream_words = std::max(hb.tr_plan.stream_words, cur_words);
            }
        }
        if (cur.n_tiles) hb.tr_plan.passes.push_back(cur);
    }
    hb.n_chal = chal;
    hb.static_term_off = term;
    hb.n_terms = term + hb.n_static;

    // ---- ops sorted by slot, emitted ciphertext points ----
    hb.emit_base = hb.n_static + (u32)hb.decs.size();
    hb.n_emit = 2 * (u32)hb.cms.size();
    hb.n_points = hb.emit_base + hb.n_emit;
    for (size_t c = 0; c < hb.cms.size(); c++) hb.cms[c].src_ct_pt = hb.emit_base + 2 * (u32)c;
    hb.slot_first_op.resize(slot_ops.size()); hb.slot_last_op.resize(slot_ops.size());
    for (size_t u = 0; u < slot_ops.size(); u++) {
        hb.slot_first_op[u] = (u32)hb.ops.size();
        for (auto &op : slot_ops[u]) hb.ops.push_back(op);
        hb.slot_last_op[u] = (u32)hb.ops.size() - 1;
    }
    return true;
}

}  // namespace xhe
