// Host-side parser of the flat transaction record (layout in include/xhe.h) and the structural checks
// of Transaction::pre_verify that need no arithmetic: verify_commitment_assets
// (/root/reference/src/tx/verify.rs:161-199) and the MultiSig payload rules (:401-418).
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "xhe_sc.h"

namespace xhe {

// Offsets (or lengths) of a record's transfers: transfer k lies at first + 320 k unless a transfer carries extra data, in
// which case the table is spelled out.  Read like a vector: size(), [k].
struct transfer_table {
    u32 n = 0, first = 0, step = 320;
    std::vector<u32> spelled;
    size_t size() const { return n; }
    u32 operator[](size_t k) const { return spelled.empty() ? first + step * (u32)k : spelled[k]; }
};

struct tx_record {
    u8 type, flags; u16 n_items, n_commitments, n_multisig, rp_len; u32 opaque_len;
    const u8 *base;            // start of the record
    u32 pre_off, pre_len;      // pre-image (without multisig entries), offset from base
    u32 ms_off, sig_off;       // multisig entries, signature
    u32 payload_off, payload_len, rp_off, cm_off;
    u8 version; u64 fee, nonce;
    transfer_table transfer_off, transfer_len;     // Transfers: each asset,dest,C,D_s,D_r,[extra],proof(160)
    bool has_multisig() const { return flags & 1; }
    const u8 *source() const { return base + pre_off + 1; }
    const u8 *pre() const { return base + pre_off; }
    const u8 *payload() const { return base + payload_off; }
    const u8 *commitment(size_t i) const { return base + cm_off + 256 * i; }        // asset, commitment, proof
    const u8 *transfer(size_t i) const { return base + transfer_off[i]; }
    const u8 *transfer_proof(size_t i) const { return base + transfer_off[i] + transfer_len[i] - 160; }
    const u8 *rp() const { return base + rp_off; }
    const u8 *ms_entry(size_t i) const { return base + ms_off + 65 * i; }
    const u8 *signature() const { return base + sig_off; }
    u32 lg() const { return (rp_len / 32 - 9) / 2; }
    u32 n_value_commitments() const { return n_commitments + (type == 0 ? n_items : 0); }
    u32 m() const { u32 k = n_value_commitments(), m = 1; while (m < k) m <<= 1; return m; }
};

static inline u64 rec_be64(const u8 *p) { u64 x = 0; for (int i = 0; i < 8; i++) x = (x << 8) | p[i]; return x; }
static inline u32 rec_le16(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8); }
static inline u32 rec_le32(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
static inline bool rec_scalar_canonical(const u8 *p) {
    u32 w[8];
    for (int i = 0; i < 8; i++) w[i] = rec_le32(p + 4 * i);
    return sc_words_canonical(w);
}
static inline bool rec_scalars_canonical(const u8 *p, int count) {
    for (int i = 0; i < count; i++) if (!rec_scalar_canonical(p + 32 * i)) return false;
    return true;
}

static inline bool record_scalars_canonical(const tx_record &t);

// false = the record would not have deserialised in the reference (wrong lengths, non-canonical scalars,
// RangeProof::from_bytes / InnerProductProof::from_bytes format errors).  check_scalars = false leaves the scalars' range
// check (record_scalars_canonical, which touches every part of the record) to the caller: the batch upload has the device do
// it (k_expand_descriptors), where the record's bytes are read anyway.
static inline bool record_parse(tx_record &t, const u8 *rec, u64 len, bool check_scalars = true) {
    if (len < 16 || len > 0x7fffffffull) return false;
    t.base = rec;
    t.type = rec[0]; t.flags = rec[1]; t.n_items = (u16)rec_le16(rec + 2); t.n_commitments = (u16)rec_le16(rec + 4);
    t.n_multisig = (u16)rec_le16(rec + 6); t.opaque_len = rec_le32(rec + 8); t.rp_len = (u16)rec_le16(rec + 12);
    if (t.type > 4 || (t.flags & ~1u) || rec_le16(rec + 14) != 0) return false;
    if (!t.has_multisig() && t.n_multisig) return false;
    u64 off = 16, payload = 0;
    t.transfer_off = transfer_table(); t.transfer_len = transfer_table();
    bool any_extra = false;
    const u8 *extra_table = rec + 16;
    if (t.type == 0) {
        if (len < off + 4ull * t.n_items) return false;
        for (u32 i = 0; i < t.n_items; i++) {
            u32 e = rec_le32(extra_table + 4 * i);
            if ((e != 0 && e < 64) || e > (1u << 20)) return false;
            any_extra |= e != 0; payload += 320 + e;
        }
        off += 4ull * t.n_items;
        if (t.opaque_len) return false;
    } else if (t.type == 1) { payload = 40; if (t.n_items || t.opaque_len) return false; }
    else if (t.type == 2) { payload = 1 + 32ull * t.n_items; if (t.opaque_len) return false; }
    else if (t.type == 3) { payload = 32 + 40ull * t.n_items + t.opaque_len; }
    else { payload = t.opaque_len; if (t.n_items) return false; }
    u64 pre_len = 49 + payload + t.rp_len + 256ull * t.n_commitments;
    u64 total = off + pre_len + 65ull * t.n_multisig + 64;
    if (len != total) return false;
    t.pre_off = (u32)off; t.pre_len = (u32)pre_len;
    const u8 *pre = rec + off;
    t.version = pre[0]; t.fee = rec_be64(pre + 33); t.nonce = rec_be64(pre + 41);
    t.payload_off = t.pre_off + 49; t.payload_len = (u32)payload;
    if (t.type == 0) {
        t.transfer_off.n = t.transfer_len.n = t.n_items;
        t.transfer_off.first = t.payload_off; t.transfer_off.step = 320;
        t.transfer_len.first = 320; t.transfer_len.step = 0;
        if (any_extra) {
            u32 p = t.payload_off;
            for (u32 i = 0; i < t.n_items; i++) { u32 e = rec_le32(extra_table + 4 * i); t.transfer_off.spelled.push_back(p); t.transfer_len.spelled.push_back(320 + e); p += 320 + e; }
        }
    }
    t.rp_off = t.payload_off + (u32)payload;
    t.cm_off = t.rp_off + t.rp_len;
    t.ms_off = t.pre_off + t.pre_len;
    t.sig_off = t.ms_off + 65u * t.n_multisig;
    if (t.rp_len % 32 != 0 || t.rp_len < 9 * 32 || ((t.rp_len / 32 - 9) & 1) || t.lg() >= 32) return false;
    return !check_scalars || record_scalars_canonical(t);
}
// every scalar of a parsed record is below the group order: the proofs' responses, the signatures, the range proof's
// t_x, t_x_blinding, e_blinding, a, b (Scalar::from_canonical_bytes in the reference's deserialisation)
static inline bool record_scalars_canonical(const tx_record &t) {
    for (size_t i = 0; i < t.transfer_off.size(); i++) if (!rec_scalars_canonical(t.transfer_proof(i) + 96, 2)) return false;
    for (u32 i = 0; i < t.n_commitments; i++) if (!rec_scalars_canonical(t.commitment(i) + 64 + 96, 3)) return false;
    if (!rec_scalars_canonical(t.signature(), 2)) return false;
    for (u32 i = 0; i < t.n_multisig; i++) if (!rec_scalars_canonical(t.ms_entry(i) + 1, 2)) return false;
    return rec_scalars_canonical(t.rp() + 128, 3) && rec_scalars_canonical(t.rp() + t.rp_len - 64, 2);
}

static inline bool rec_is_zero32(const u8 *p) { u8 a = 0; for (int i = 0; i < 32; i++) a |= p[i]; return a == 0; }

// verify.rs:161-199
static inline bool record_commitment_assets_ok(const tx_record &t) {
    auto has = [&](const u8 *asset) {
        for (u32 i = 0; i < t.n_commitments; i++) if (!memcmp(t.commitment(i), asset, 32)) return true;
        return false;
    };
    const u8 native[32] = {0};
    if (!has(native)) return false;
    for (u32 i = 0; i < t.n_commitments; i++)
        for (u32 j = i + 1; j < t.n_commitments; j++)
            if (!memcmp(t.commitment(i), t.commitment(j), 32)) return false;
    if (t.type == 0) { for (u32 i = 0; i < t.n_items; i++) if (!has(t.transfer(i))) return false; }
    else if (t.type == 1) { if (!has(t.payload())) return false; }
    else if (t.type == 3) { for (u32 i = 0; i < t.n_items; i++) if (!has(t.payload() + 32 + 40 * i)) return false; }
    return true;
}

// verify.rs:401-418: threshold/uniqueness/self rules of a MultiSig payload
static inline bool record_multisig_payload_ok(const tx_record &t) {
    const u8 *p = t.payload();
    u32 threshold = p[0], ns = t.n_items;
    if (threshold > ns || (ns != 0 && threshold == 0)) return false;
    for (u32 i = 0; i < ns; i++)
        for (u32 j = i + 1; j < ns; j++)
            if (!memcmp(p + 1 + 32 * i, p + 1 + 32 * j, 32)) return false;
    for (u32 i = 0; i < ns; i++) if (!memcmp(p + 1 + 32 * i, t.source(), 32)) return false;
    return true;
}

// amount subtracted in the clear from the sender's balance of `asset` (fee, burn, contract deposits):
// the scalar part of get_sender_output_ct, verify.rs:107-144
static inline u64 record_plain_output(const tx_record &t, const u8 *asset, bool *overflow) {
    unsigned __int128 s = 0;
    if (rec_is_zero32(asset)) s += t.fee;
    if (t.type == 1) { if (!memcmp(asset, t.payload(), 32)) s += rec_be64(t.payload() + 32); }
    else if (t.type == 3) { for (u32 i = 0; i < t.n_items; i++) if (!memcmp(asset, t.payload() + 32 + 40 * i, 32)) s += rec_be64(t.payload() + 64 + 40 * i); }
    if (overflow) *overflow = (s >> 64) != 0;
    return (u64)s;
}

}  // namespace xhe
