// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under xelis-he_amd/ may include, link or call this.
//
// Sequential restatement of Transaction::pre_verify / verify / verify_batch / apply_without_verify
// (/root/reference/src/tx/verify.rs:203-619) over the flat transaction record and ledger snapshot
// the C ABI uses, with an in-memory ledger equivalent to mock::Ledger (/root/reference/src/lib.rs:106-201).
#pragma once
#include <map>
#include <array>
#include <vector>
#include "xo_proofs.h"

namespace xo {

// ---- flat ledger snapshot (same field order as xhe_ledger_view in include/xhe.h) ----------------
struct ledger_view {
    u64 n_accounts; const u8 *account_keys; const u64 *account_nonces;
    u64 n_balances; const u32 *balance_account; const u8 *balance_assets; const u8 *balance_cts;
    u64 n_multisig; const u32 *multisig_account; const u8 *multisig_threshold; const u32 *multisig_offsets; const u8 *multisig_signers;
};
struct ledger_out {
    u8 *balance_cts; u64 *account_nonces; int64_t *multisig_set_by;
};

typedef std::array<u8, 32> key32;
static inline key32 mk32(const u8 *p) { key32 k; memcpy(k.data(), p, 32); return k; }

// mock::Ledger equivalent (lib.rs:106-201)
struct ledger {
    struct account { u64 nonce; std::map<key32, std::array<u8, 64>> balances; size_t index; };
    std::map<key32, account> accounts;
    std::map<key32, std::pair<std::vector<key32>, u8>> multisig;
    std::vector<int64_t> multisig_set_by;
    void load(const ledger_view &v) {
        std::vector<key32> keys(v.n_accounts);
        for (u64 i = 0; i < v.n_accounts; i++) {
            keys[i] = mk32(v.account_keys + 32 * i);
            account &a = accounts[keys[i]]; a.nonce = v.account_nonces[i]; a.index = i;
        }
        for (u64 i = 0; i < v.n_balances; i++) {
            std::array<u8, 64> ct; memcpy(ct.data(), v.balance_cts + 64 * i, 64);
            accounts[keys[v.balance_account[i]]].balances[mk32(v.balance_assets + 32 * i)] = ct;
        }
        for (u64 i = 0; i < v.n_multisig; i++) {
            std::vector<key32> s;
            for (u32 j = v.multisig_offsets[i]; j < v.multisig_offsets[i + 1]; j++) s.push_back(mk32(v.multisig_signers + 32 * j));
            multisig[keys[v.multisig_account[i]]] = std::make_pair(s, v.multisig_threshold[i]);
        }
        multisig_set_by.assign(v.n_accounts, -1);
    }
    void store(const ledger_view &v, const ledger_out &o) const {
        for (u64 i = 0; i < v.n_accounts; i++)
            if (o.account_nonces) o.account_nonces[i] = accounts.at(mk32(v.account_keys + 32 * i)).nonce;
        for (u64 i = 0; i < v.n_balances; i++) {
            const account &a = accounts.at(mk32(v.account_keys + 32 * v.balance_account[i]));
            if (o.balance_cts) memcpy(o.balance_cts + 64 * i, a.balances.at(mk32(v.balance_assets + 32 * i)).data(), 64);
        }
        if (o.multisig_set_by) for (u64 i = 0; i < v.n_accounts; i++) o.multisig_set_by[i] = multisig_set_by[i];
    }
};

// ---- flat transaction record ---------------------------------------------------------------------
// header (16 bytes, little-endian): u8 type, u8 flags(bit0: multisig present), u16 n_items,
// u16 n_commitments, u16 n_multisig, u32 opaque_len, u16 rp_len, u16 reserved;
// then (type 0 only) n_items x u32 extra_len; then the signing pre-image exactly as
// Transaction::to_bytes lays it out (verify.rs:623-688); then n_multisig x (u8 id, 64-byte signature);
// then the 64-byte signature s || e.
enum { TX_TRANSFERS = 0, TX_BURN = 1, TX_MULTISIG = 2, TX_CALL_CONTRACT = 3, TX_DEPLOY_CONTRACT = 4 };
struct tx_view {
    u8 type, flags; u16 n_items, n_commitments, n_multisig, rp_len; u32 opaque_len;
    const u32 *extra_len;
    const u8 *pre; size_t pre_len;        // pre-image without multisig entries
    const u8 *ms;                          // multisig entries
    const u8 *sig;
    u8 version; const u8 *source; u64 fee, nonce;
    const u8 *payload; size_t payload_len;
    std::vector<const u8 *> transfers;    // each: asset,dest,C,D_s,D_r,[extra],proof(160)
    std::vector<size_t> transfer_len;
    const u8 *rp; const u8 *commitments;  // n_commitments x (asset32, commitment32, proof192)
    bool has_multisig() const { return flags & 1; }
    const u8 *transfer_proof(size_t i) const { return transfers[i] + transfer_len[i] - 160; }
};
static inline u64 load_be64(const u8 *p) { u64 x = 0; for (int i = 0; i < 8; i++) x = (x << 8) | p[i]; return x; }
static inline u16 load_le16(const u8 *p) { return (u16)(p[0] | (p[1] << 8)); }
static inline u32 load_le32(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

static inline bool canonical_scalars(const u8 *p, int count) {
    for (int i = 0; i < count; i++) if (!sc_is_canonical_bytes(p + 32 * i)) return false;
    return true;
}

// Parses and checks everything the reference's deserialisation would (lengths, canonical scalars).
static inline bool tx_parse(tx_view &t, const u8 *rec, size_t len) {
    if (len < 16) return false;
    t.type = rec[0]; t.flags = rec[1]; t.n_items = load_le16(rec + 2); t.n_commitments = load_le16(rec + 4);
    t.n_multisig = load_le16(rec + 6); t.opaque_len = load_le32(rec + 8); t.rp_len = load_le16(rec + 12);
    if (t.type > TX_DEPLOY_CONTRACT || (t.flags & ~1u) || load_le16(rec + 14) != 0) return false;
    if (!t.has_multisig() && t.n_multisig) return false;
    size_t off = 16;
    t.extra_len = nullptr;
    std::vector<u32> extra;
    if (t.type == TX_TRANSFERS) {
        if (len < off + 4 * (size_t)t.n_items) return false;
        t.extra_len = (const u32 *)(rec + off);
        for (size_t i = 0; i < t.n_items; i++) extra.push_back(load_le32(rec + off + 4 * i));
        off += 4 * (size_t)t.n_items;
    }
    t.pre = rec + off;
    size_t payload;
    switch (t.type) {
        case TX_TRANSFERS: { payload = 0; for (u32 e : extra) { if (e != 0 && e < 64) return false; payload += 320 + e; } if (t.opaque_len) return false; break; }
        case TX_BURN: payload = 40; if (t.n_items || t.opaque_len) return false; break;
        case TX_MULTISIG: payload = 1 + 32 * (size_t)t.n_items; if (t.opaque_len) return false; break;
        case TX_CALL_CONTRACT: payload = 32 + 40 * (size_t)t.n_items + t.opaque_len; break;
        default: payload = t.opaque_len; if (t.n_items) return false; break;
    }
    t.pre_len = 49 + payload + t.rp_len + 256 * (size_t)t.n_commitments;
    size_t total = off + t.pre_len + 65 * (size_t)t.n_multisig + 64;
    if (len != total) return false;
    t.version = t.pre[0]; t.source = t.pre + 1; t.fee = load_be64(t.pre + 33); t.nonce = load_be64(t.pre + 41);
    t.payload = t.pre + 49; t.payload_len = payload;
    t.transfers.clear(); t.transfer_len.clear();
    if (t.type == TX_TRANSFERS) {
        const u8 *p = t.payload;
        for (u32 e : extra) { t.transfers.push_back(p); t.transfer_len.push_back(320 + e); p += 320 + e; }
    }
    t.rp = t.payload + payload;
    t.commitments = t.rp + t.rp_len;
    t.ms = t.pre + t.pre_len;
    t.sig = t.ms + 65 * (size_t)t.n_multisig;
    // scalar canonicity (serde would have refused the transaction)
    for (size_t i = 0; i < t.transfers.size(); i++) if (!canonical_scalars(t.transfer_proof(i) + 96, 2)) return false;
    for (size_t i = 0; i < t.n_commitments; i++) if (!canonical_scalars(t.commitments + 256 * i + 64 + 96, 3)) return false;
    if (!canonical_scalars(t.sig, 2)) return false;
    for (size_t i = 0; i < t.n_multisig; i++) if (!canonical_scalars(t.ms + 65 * i + 1, 2)) return false;
    // RangeProof::from_bytes
    if (t.rp_len % 32 != 0 || t.rp_len < 9 * 32 || ((t.rp_len / 32 - 9) & 1) || (t.rp_len / 32 - 9) / 2 >= 32) return false;
    if (!canonical_scalars(t.rp + 128, 3) || !canonical_scalars(t.rp + t.rp_len - 64, 2)) return false;
    return true;
}

static inline bool is_zero32(const u8 *p) { u8 a = 0; for (int i = 0; i < 32; i++) a |= p[i]; return a == 0; }

// verify.rs:161-199
static inline bool verify_commitment_assets(const tx_view &t) {
    auto has = [&](const u8 *asset) {
        for (size_t i = 0; i < t.n_commitments; i++) if (!memcmp(t.commitments + 256 * i, asset, 32)) return true;
        return false;
    };
    u8 native[32] = {0};
    if (!has(native)) return false;
    for (size_t i = 0; i < t.n_commitments; i++)
        for (size_t j = 0; j < t.n_commitments; j++)
            if (i != j && !memcmp(t.commitments + 256 * i, t.commitments + 256 * j, 32)) return false;
    switch (t.type) {
        case TX_TRANSFERS: for (size_t i = 0; i < t.transfers.size(); i++) if (!has(t.transfers[i])) return false; return true;
        case TX_BURN: return has(t.payload);
        case TX_CALL_CONTRACT: for (size_t i = 0; i < t.n_items; i++) if (!has(t.payload + 32 + 40 * i)) return false; return true;
        default: return true;
    }
}

struct ct { ge C, D; };
static inline ct ct_zero() { ct r = {ge_identity(), ge_identity()}; return r; }
static inline ct ct_add(const ct &a, const ct &b) { ct r = {ge_add(a.C, b.C), ge_add(a.D, b.D)}; return r; }
static inline ct ct_sub(const ct &a, const ct &b) { ct r = {ge_sub(a.C, b.C), ge_sub(a.D, b.D)}; return r; }
static inline ct ct_add_scalar(const ct &a, u64 x) { ct r = {ge_add(a.C, ge_scalarmul(sc_from_u64(x), ge_basepoint())), a.D}; return r; }
static inline bool ct_decode(ct &out, const u8 b[64]) { return ristretto_decode(out.C, b) && ristretto_decode(out.D, b + 32); }
static inline std::array<u8, 64> ct_encode(const ct &c) { std::array<u8, 64> r; ristretto_encode(r.data(), c.C); ristretto_encode(r.data() + 32, c.D); return r; }

struct transfer_pts { ge C, Ds, Dr; };

// verify.rs:107-144
static inline ct sender_output_ct(const tx_view &t, const u8 *asset, const std::vector<transfer_pts> &tp) {
    ct bal = ct_zero();
    if (is_zero32(asset)) bal = ct_add_scalar(bal, t.fee);
    switch (t.type) {
        case TX_TRANSFERS:
            for (size_t i = 0; i < t.transfers.size(); i++)
                if (!memcmp(asset, t.transfers[i], 32)) { ct c = {tp[i].C, tp[i].Ds}; bal = ct_add(bal, c); }
            break;
        case TX_BURN: if (!memcmp(asset, t.payload, 32)) bal = ct_add_scalar(bal, load_be64(t.payload + 32)); break;
        case TX_CALL_CONTRACT:
            for (size_t i = 0; i < t.n_items; i++)
                if (!memcmp(asset, t.payload + 32 + 40 * i, 32)) bal = ct_add_scalar(bal, load_be64(t.payload + 64 + 40 * i));
            break;
        default: break;
    }
    return bal;
}

struct prepared_tx { transcript tr; std::vector<const u8 *> V_bytes; std::vector<ge> V_points; };
static const u8 ZERO32[32] = {0};

// verify.rs:203-485.  `tx_index` is only recorded into multisig_set_by.
static inline int pre_verify(const tx_view &t, ledger &st, batch_collector &bc, drbg &rng, prepared_tx &out, int64_t tx_index) {
    auto acc_it = st.accounts.find(mk32(t.source));
    if (acc_it == st.accounts.end()) return XO_ERR_STATE;
    if (acc_it->second.nonce != t.nonce) return XO_ERR_INVALID_NONCE;
    acc_it->second.nonce = t.nonce;                               // verify.rs:219-221 (stored back unchanged)
    if (!verify_commitment_assets(t)) return XO_ERR_FORMAT;

    std::vector<transfer_pts> tp(t.transfers.size());
    for (size_t i = 0; i < t.transfers.size(); i++) {
        const u8 *p = t.transfers[i];
        if (!ristretto_decode(tp[i].C, p + 64) || !ristretto_decode(tp[i].Ds, p + 96) || !ristretto_decode(tp[i].Dr, p + 128))
            return XO_ERR_DECOMPRESSION;
    }
    std::vector<ge> new_commitments(t.n_commitments);
    for (size_t i = 0; i < t.n_commitments; i++)
        if (!ristretto_decode(new_commitments[i], t.commitments + 256 * i + 32)) return XO_ERR_DECOMPRESSION;
    ge source;
    if (!ristretto_decode(source, t.source)) return XO_ERR_DECOMPRESSION;

    // verify.rs:146-158
    out.tr.init("transaction-proof");
    out.tr.append_u64("version", t.version);
    out.tr.append_message("source_pubkey", t.source, 32);
    out.tr.append_u64("fee", t.fee);
    out.tr.append_u64("nonce", t.nonce);

    // signature over pre-image including multisig entries (verify.rs:253-256, 679-685)
    if (!signature_verify(t.sig, t.pre, t.pre_len + 65 * (size_t)t.n_multisig, source, t.source)) return XO_ERR_SIGNATURE;

    // verify.rs:259-292
    auto ms_it = st.multisig.find(mk32(t.source));
    if (ms_it != st.multisig.end()) {
        if (!t.has_multisig()) return XO_ERR_FORMAT;
        const std::vector<key32> &signers = ms_it->second.first;
        u8 threshold = ms_it->second.second;
        if (t.n_multisig == 0 || t.n_multisig != threshold) return XO_ERR_FORMAT;
        u8 hash[32]; blake3_hash(hash, t.pre, t.pre_len);
        for (size_t i = 0; i < t.n_multisig; i++) {
            u8 idx = t.ms[65 * i];
            for (size_t j = 0; j < t.n_multisig; j++) if (i != j && t.ms[65 * j] == idx) return XO_ERR_FORMAT;
            if (idx < signers.size()) {
                ge signer;
                if (!ristretto_decode(signer, signers[idx].data())) return XO_ERR_DECOMPRESSION;
                if (!signature_verify(t.ms + 65 * i + 1, hash, 32, signer, signers[idx].data())) return XO_ERR_SIGNATURE;
            }
        }
    } else if (t.has_multisig()) {
        return XO_ERR_FORMAT;
    }

    // 1. CommitmentEqProofs (verify.rs:296-341)
    for (size_t i = 0; i < t.n_commitments; i++) {
        const u8 *cm = t.commitments + 256 * i;
        auto bal_it = acc_it->second.balances.find(mk32(cm));
        if (bal_it == acc_it->second.balances.end()) return XO_ERR_STATE;
        ct cur;
        if (!ct_decode(cur, bal_it->second.data())) return XO_ERR_DECOMPRESSION;
        ct output = sender_output_ct(t, cm, tp);
        ct new_ct = ct_sub(cur, output);
        out.tr.append_message("dom-sep", (const u8 *)"new-commitment-proof", 20);
        out.tr.append_message("new_source_commitment_asset", cm, 32);
        out.tr.append_message("new_source_commitment", cm + 32, 32);
        int rc = commitment_eq_pre_verify(cm + 64, source, new_ct.C, new_ct.D, new_commitments[i], out.tr, bc, rng);
        if (rc) return rc;
        bal_it->second = ct_encode(new_ct);
    }

    // 2. CiphertextValidityProofs / type-specific transcript (verify.rs:344-430)
    if (t.type == TX_TRANSFERS) {
        for (size_t i = 0; i < t.transfers.size(); i++) {
            const u8 *p = t.transfers[i];
            ge receiver;
            if (!ristretto_decode(receiver, p + 32)) return XO_ERR_DECOMPRESSION;
            auto r_it = st.accounts.find(mk32(p + 32));
            if (r_it == st.accounts.end()) return XO_ERR_STATE;
            auto rb_it = r_it->second.balances.find(mk32(p));
            if (rb_it == r_it->second.balances.end()) return XO_ERR_STATE;
            ct cur;
            if (!ct_decode(cur, rb_it->second.data())) return XO_ERR_DECOMPRESSION;
            ct recv = {tp[i].C, tp[i].Dr};
            rb_it->second = ct_encode(ct_add(cur, recv));
            out.tr.append_message("dom-sep", (const u8 *)"transfer-proof", 14);
            out.tr.append_message("dest_pubkey", p + 32, 32);
            out.tr.append_message("amount_commitment", p + 64, 32);
            out.tr.append_message("amount_sender_handle", p + 96, 32);
            out.tr.append_message("amount_receiver_handle", p + 128, 32);
            int rc = ct_validity_pre_verify(t.transfer_proof(i), tp[i].C, receiver, source, tp[i].Dr, tp[i].Ds, out.tr, bc, rng);
            if (rc) return rc;
        }
    } else if (t.type == TX_BURN) {
        out.tr.append_message("dom-sep", (const u8 *)"burn-proof", 10);
        out.tr.append_message("asset", t.payload, 32);
        out.tr.append_u64("amount", load_be64(t.payload + 32));
    } else if (t.type == TX_MULTISIG) {
        u8 threshold = t.payload[0];
        size_t ns = t.n_items;
        if (threshold > ns || (ns != 0 && threshold == 0)) return XO_ERR_FORMAT;
        for (size_t i = 0; i < ns; i++)
            for (size_t j = 0; j < ns; j++)
                if (i != j && !memcmp(t.payload + 1 + 32 * i, t.payload + 1 + 32 * j, 32)) return XO_ERR_FORMAT;
        for (size_t i = 0; i < ns; i++) if (!memcmp(t.payload + 1 + 32 * i, t.source, 32)) return XO_ERR_FORMAT;
        out.tr.append_message("dom-sep", (const u8 *)"multisig-proof", 14);
        out.tr.append_u64("threshold", threshold);
        for (size_t i = 0; i < ns; i++) out.tr.append_message("signer", t.payload + 1 + 32 * i, 32);
        // lib.rs:181-193
        if (ns == 0) st.multisig.erase(mk32(t.source));
        else {
            std::vector<key32> s; for (size_t i = 0; i < ns; i++) s.push_back(mk32(t.payload + 1 + 32 * i));
            st.multisig[mk32(t.source)] = std::make_pair(s, threshold);
        }
        st.multisig_set_by[acc_it->second.index] = tx_index;
    }

    // value commitments for the range proof (verify.rs:434-478)
    size_t n_commit = t.n_commitments + (t.type == TX_TRANSFERS ? t.transfers.size() : 0);
    size_t m = 1; while (m < n_commit) m <<= 1;
    out.V_bytes.clear(); out.V_points.clear();
    for (size_t i = 0; i < t.n_commitments; i++) { out.V_bytes.push_back(t.commitments + 256 * i + 32); out.V_points.push_back(new_commitments[i]); }
    if (t.type == TX_TRANSFERS)
        for (size_t i = 0; i < t.transfers.size(); i++) { out.V_bytes.push_back(t.transfers[i] + 64); out.V_points.push_back(tp[i].C); }
    while (out.V_bytes.size() < m) { out.V_bytes.push_back(ZERO32); out.V_points.push_back(ge_identity()); }
    return XO_OK;
}

struct verdict { int32_t accept; int32_t err_code; int64_t first_bad_tx; };

// verify.rs:487-517.  The ledger is mutated incrementally, also on failure, as in the reference.
static inline verdict verify_batch(const std::vector<tx_view> &txs, ledger &st, const u8 seed[32]) {
    drbg rng; rng.init(seed);
    batch_collector bc;
    std::vector<prepared_tx> prepared(txs.size());
    for (size_t i = 0; i < txs.size(); i++) {
        int rc = pre_verify(txs[i], st, bc, rng, prepared[i], (int64_t)i);
        if (rc) { verdict v = {0, rc, (int64_t)i}; return v; }
    }
    if (!bc.verify()) { verdict v = {0, XO_ERR_GENERIC_PROOF, -1}; return v; }
    rp_batch rb;
    for (size_t i = 0; i < txs.size(); i++) {
        rp_terms T;
        int rc = range_proof_terms(T, txs[i].rp, txs[i].rp_len, prepared[i].V_bytes, prepared[i].V_points, prepared[i].tr, rng.next_scalar());
        if (rc) { verdict v = {0, rc, (int64_t)i}; return v; }
        rb.add(T, rng.next_scalar());
    }
    if (!txs.empty() && !rb.verify()) { verdict v = {0, XO_ERR_RANGE_PROOF, -1}; return v; }
    verdict v = {1, XO_OK, -1};
    return v;
}

// verify.rs:520-542
static inline verdict verify_single(const tx_view &tx, ledger &st, const u8 seed[32]) {
    drbg rng; rng.init(seed);
    batch_collector bc; prepared_tx p;
    int rc = pre_verify(tx, st, bc, rng, p, 0);
    if (rc) { verdict v = {0, rc, 0}; return v; }
    if (!bc.verify()) { verdict v = {0, XO_ERR_GENERIC_PROOF, 0}; return v; }
    rc = range_proof_verify_single(tx.rp, tx.rp_len, p.V_bytes, p.V_points, p.tr, rng);
    if (rc) { verdict v = {0, rc, 0}; return v; }
    verdict v = {1, XO_OK, -1};
    return v;
}

// verify.rs:545-619 (ledger mutation only).  Returns false where the reference would panic.
static inline bool apply_without_verify(const tx_view &t, ledger &st, int64_t tx_index) {
    auto acc_it = st.accounts.find(mk32(t.source));
    if (acc_it == st.accounts.end()) return false;
    std::vector<transfer_pts> tp(t.transfers.size());
    for (size_t i = 0; i < t.transfers.size(); i++) {
        const u8 *p = t.transfers[i];
        if (!ristretto_decode(tp[i].C, p + 64) || !ristretto_decode(tp[i].Ds, p + 96) || !ristretto_decode(tp[i].Dr, p + 128)) return false;
    }
    for (size_t i = 0; i < t.n_commitments; i++) {
        const u8 *cm = t.commitments + 256 * i;
        auto bal_it = acc_it->second.balances.find(mk32(cm));
        if (bal_it == acc_it->second.balances.end()) return false;
        ct cur; if (!ct_decode(cur, bal_it->second.data())) return false;
        bal_it->second = ct_encode(ct_sub(cur, sender_output_ct(t, cm, tp)));
    }
    if (t.type == TX_TRANSFERS) {
        for (size_t i = 0; i < t.transfers.size(); i++) {
            const u8 *p = t.transfers[i];
            auto r_it = st.accounts.find(mk32(p + 32));
            if (r_it == st.accounts.end()) return false;
            auto rb_it = r_it->second.balances.find(mk32(p));
            if (rb_it == r_it->second.balances.end()) return false;
            ct cur; if (!ct_decode(cur, rb_it->second.data())) return false;
            ct recv = {tp[i].C, tp[i].Dr};
            rb_it->second = ct_encode(ct_add(cur, recv));
        }
    } else if (t.type == TX_MULTISIG) {
        size_t ns = t.n_items;
        if (ns == 0) st.multisig.erase(mk32(t.source));
        else {
            std::vector<key32> s; for (size_t i = 0; i < ns; i++) s.push_back(mk32(t.payload + 1 + 32 * i));
            st.multisig[mk32(t.source)] = std::make_pair(s, t.payload[0]);
        }
        st.multisig_set_by[acc_it->second.index] = tx_index;
    }
    return true;
}

}  // namespace xo
