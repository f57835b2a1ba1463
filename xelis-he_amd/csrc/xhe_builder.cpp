// Transaction builder (prover) implementation; see xhe_builder.h.
#include "xhe_builder.h"
#include <string.h>
#include <algorithm>

namespace xhe {
namespace host {

static const bytes32 NATIVE_ASSET = {};

static void put_be64(std::vector<u8> &v, u64 x) { for (int i = 7; i >= 0; i--) v.push_back((u8)(x >> (8 * i))); }
static void put_le16(std::vector<u8> &v, u32 x) { v.push_back((u8)x); v.push_back((u8)(x >> 8)); }
static void put_le32(std::vector<u8> &v, u32 x) { for (int i = 0; i < 4; i++) v.push_back((u8)(x >> (8 * i))); }
static void put(std::vector<u8> &v, const u8 *p, size_t n) { v.insert(v.end(), p, p + n); }
static void put_sc(std::vector<u8> &v, const sc &x) { u8 b[32]; sc_tobytes(b, x); put(v, b, 32); }
static void put_pt(std::vector<u8> &v, const ge &p) { u8 b[32]; encode(b, p); put(v, b, 32); }

// elgamal.rs:53-65
static sc hash_and_point_to_scalar(const u8 pub[32], const u8 *msg, size_t n, const ge &point) {
    u8 enc[32]; encode(enc, point);
    sponge sp; sp.init(72, 0x06);
    sp.absorb(pub, 32); sp.absorb(msg, (u32)n); sp.absorb(enc, 32);
    u8 h[64]; sp.squeeze(h, 64);
    return sc_from_wide_bytes(h);
}

bytes64 sign(const keypair &kp, const u8 *msg, size_t n, drbg &rng) {
    sc k = rng.scalar();
    ge r = mul_H(k);
    sc e = hash_and_point_to_scalar(kp.pub_bytes.data(), msg, n, r);
    sc s = sc_add(sc_mul(sc_invert(kp.secret), e), k);
    bytes64 out; sc_tobytes(out.data(), s); sc_tobytes(out.data() + 32, e);
    return out;
}

bool signature_verify_host(const u8 sig[64], const u8 *msg, size_t n, const u8 pub[32]) {
    ge key; if (!ristretto_decode(key, pub)) return false;
    sc s = sc_frombytes(sig), e = sc_frombytes(sig + 32);
    ge r = ge_add(mul_H(s), scalarmul(sc_neg(e), key));
    return sc_eq(e, hash_and_point_to_scalar(pub, msg, n, r));
}

// ---- sigma proofs ----------------------------------------------------------------------------------
// proofs.rs:73-132.  Appends Y_0,Y_1,Y_2,z_s,z_x,z_r (192 bytes).
static void commitment_eq_prove(std::vector<u8> &out, const keypair &src, const ciphertext &src_ct, const sc &opening, u64 amount,
                                transcript &t, drbg &rng) {
    t.append_message("dom-sep", (const u8 *)"equality-proof", 14);
    sc x = sc_from_u64(amount);
    sc y_s = rng.scalar(), y_x = rng.scalar(), y_r = rng.scalar();
    ge Y0 = scalarmul(y_s, src.pub);
    ge Y1 = ge_add(mul_G(y_x), scalarmul(y_s, src_ct.D));
    ge Y2 = ge_add(mul_G(y_x), mul_H(y_r));
    u8 b0[32], b1[32], b2[32]; encode(b0, Y0); encode(b1, Y1); encode(b2, Y2);
    t.append_point("Y_0", b0); t.append_point("Y_1", b1); t.append_point("Y_2", b2);
    sc c = t.challenge_scalar("c");
    sc z_s = sc_add(sc_mul(c, src.secret), y_s), z_x = sc_add(sc_mul(c, x), y_x), z_r = sc_add(sc_mul(c, opening), y_r);
    t.append_scalar("z_s", z_s); t.append_scalar("z_x", z_x); t.append_scalar("z_r", z_r);
    (void)t.challenge_scalar("w");
    put(out, b0, 32); put(out, b1, 32); put(out, b2, 32); put_sc(out, z_s); put_sc(out, z_x); put_sc(out, z_r);
}

// proofs.rs:237-279.  Appends Y_0,Y_1,Y_2,z_r,z_x (160 bytes).
static void ct_validity_prove(std::vector<u8> &out, const ge &dest_pub, const ge &src_pub, u64 amount, const sc &opening, transcript &t, drbg &rng) {
    t.append_message("dom-sep", (const u8 *)"validity-proof", 14);
    sc x = sc_from_u64(amount);
    sc y_r = rng.scalar(), y_x = rng.scalar();
    ge Y0 = ge_add(mul_H(y_r), mul_G(y_x));
    ge Y1 = scalarmul(y_r, dest_pub), Y2 = scalarmul(y_r, src_pub);
    u8 b0[32], b1[32], b2[32]; encode(b0, Y0); encode(b1, Y1); encode(b2, Y2);
    t.append_point("Y_0", b0); t.append_point("Y_1", b1); t.append_point("Y_2", b2);
    sc c = t.challenge_scalar("c");
    sc z_r = sc_add(sc_mul(c, opening), y_r), z_x = sc_add(sc_mul(c, x), y_x);
    t.append_scalar("z_r", z_r); t.append_scalar("z_x", z_x);
    (void)t.challenge_scalar("w");
    put(out, b0, 32); put(out, b1, 32); put(out, b2, 32); put_sc(out, z_r); put_sc(out, z_x);
}

// ---- aggregated range proof (dalek-bulletproofs prove_multiple, party/dealer folded into one pass) ---
// Sums stay lazy for a few terms only: a scalar holds values below 2^257 (xhe_sc.h), a product is below 1.07 l, so every eighth
// addition the running sum is brought back below 1.07 l (a Montgomery product with one).
static inline sc sc_reduce(const sc &x) { return sc_mul(x, sc_one()); }
static sc inner(const std::vector<sc> &a, const std::vector<sc> &b, size_t off_a, size_t off_b, size_t n) {
    sc acc = sc_zero();
    for (size_t i = 0; i < n; i++) {
        acc = sc_add(acc, sc_mul(a[off_a + i], b[off_b + i]));
        if ((i & 7) == 7) acc = sc_reduce(acc);
    }
    return sc_reduce(acc);
}

void range_proof_begin(transcript &t, const std::vector<u64> &values, const std::vector<sc> &openings) {
    const size_t n = 64, m = values.size();
    t.append_message("dom-sep", (const u8 *)"rangeproof v1", 13);
    t.append_u64("n", n); t.append_u64("m", m);
    for (size_t j = 0; j < m; j++) { u8 vb[32]; encode(vb, commit(values[j], openings[j])); t.append_point("V", vb); }
}

void range_proof_prove(std::vector<u8> &out, transcript &t, const std::vector<u64> &values, const std::vector<sc> &openings, drbg &rng) {
    const size_t n = 64, m = values.size(), nm = n * m;
    generators &g = gens(); g.ensure(nm);
    range_proof_begin(t, values, openings);

    sc a_blinding = rng.scalar(), s_blinding = rng.scalar();
    std::vector<sc> s_L(nm), s_R(nm);
    for (size_t i = 0; i < nm; i++) { s_L[i] = rng.scalar(); s_R[i] = rng.scalar(); }
    // A = a_blinding*B_blinding + sum (bit ? G_i : -H_i)
    ge A = mul_H(a_blinding);
    for (size_t i = 0; i < nm; i++) {
        bool bit = (values[i / n] >> (i % n)) & 1;
        A = bit ? ge_add(A, g.G_vec[i]) : ge_sub(A, g.H_vec[i]);
    }
    // S = s_blinding*B_blinding + <s_L, G> + <s_R, H>
    ge S;
    {
        std::vector<sc> sc_v; std::vector<ge> pt_v;
        sc_v.reserve(2 * nm); pt_v.reserve(2 * nm);
        for (size_t i = 0; i < nm; i++) { sc_v.push_back(s_L[i]); pt_v.push_back(g.G_vec[i]); }
        for (size_t i = 0; i < nm; i++) { sc_v.push_back(s_R[i]); pt_v.push_back(g.H_vec[i]); }
        S = ge_add(mul_H(s_blinding), msm(sc_v, pt_v));
    }
    u8 Ab[32], Sb[32]; encode(Ab, A); encode(Sb, S);
    t.append_point("A", Ab); t.append_point("S", Sb);
    sc y = t.challenge_scalar("y"), z = t.challenge_scalar("z");
    sc zz = sc_mul(z, z), one = sc_one(), two = sc_from_u64(2), minus_one = sc_neg(one);

    // l(x) = l0 + l1 x, r(x) = r0 + r1 x
    std::vector<sc> l0(nm), l1(nm), r0(nm), r1(nm);
    sc exp_y = one, exp_z = one, t0_blinding = sc_zero();
    for (size_t j = 0; j < m; j++) {
        sc offset_zz = sc_mul(zz, exp_z), exp_2 = one;
        for (size_t k = 0; k < n; k++) {
            size_t i = j * n + k;
            bool bit = (values[j] >> k) & 1;
            sc a_L = bit ? one : sc_zero(), a_R = bit ? sc_zero() : minus_one;
            l0[i] = sc_sub(a_L, z); l1[i] = s_L[i];
            r0[i] = sc_add(sc_mul(exp_y, sc_add(a_R, z)), sc_mul(offset_zz, exp_2));
            r1[i] = sc_mul(exp_y, s_R[i]);
            exp_y = sc_mul(exp_y, y); exp_2 = sc_mul(exp_2, two);
        }
        t0_blinding = sc_reduce(sc_add(t0_blinding, sc_mul(offset_zz, openings[j])));
        exp_z = sc_mul(exp_z, z);
    }
    sc t0 = inner(l0, r0, 0, 0, nm), t2 = inner(l1, r1, 0, 0, nm);
    sc t1;
    {
        sc acc = sc_zero();
        for (size_t i = 0; i < nm; i++) {
            acc = sc_add(acc, sc_mul(sc_add(l0[i], l1[i]), sc_add(r0[i], r1[i])));
            if ((i & 7) == 7) acc = sc_reduce(acc);
        }
        t1 = sc_sub(sc_sub(sc_reduce(acc), t0), t2);
    }
    sc t1_blinding = rng.scalar(), t2_blinding = rng.scalar();
    ge T1 = ge_add(mul_G(t1), mul_H(t1_blinding)), T2 = ge_add(mul_G(t2), mul_H(t2_blinding));
    u8 T1b[32], T2b[32]; encode(T1b, T1); encode(T2b, T2);
    t.append_point("T_1", T1b); t.append_point("T_2", T2b);
    sc x = t.challenge_scalar("x"), xx = sc_mul(x, x);
    sc t_x = sc_add(sc_add(t0, sc_mul(t1, x)), sc_mul(t2, xx));
    sc t_x_blinding = sc_add(sc_add(t0_blinding, sc_mul(t1_blinding, x)), sc_mul(t2_blinding, xx));
    sc e_blinding = sc_add(a_blinding, sc_mul(s_blinding, x));
    t.append_scalar("t_x", t_x); t.append_scalar("t_x_blinding", t_x_blinding); t.append_scalar("e_blinding", e_blinding);
    sc w = t.challenge_scalar("w");
    ge Q = mul_G(w);

    std::vector<sc> a(nm), b(nm);
    for (size_t i = 0; i < nm; i++) { a[i] = sc_add(l0[i], sc_mul(l1[i], x)); b[i] = sc_add(r0[i], sc_mul(r1[i], x)); }

    put(out, Ab, 32); put(out, Sb, 32); put(out, T1b, 32); put(out, T2b, 32);
    put_sc(out, t_x); put_sc(out, t_x_blinding); put_sc(out, e_blinding);

    // inner-product argument with H_factors = y^-i folded in during the first round
    t.append_message("dom-sep", (const u8 *)"ipp v1", 6);
    t.append_u64("n", nm);
    std::vector<ge> G(g.G_vec.begin(), g.G_vec.begin() + nm), Hh(g.H_vec.begin(), g.H_vec.begin() + nm);
    std::vector<sc> h_factor(nm);
    { sc y_inv = sc_invert(y), p = one; for (size_t i = 0; i < nm; i++) { h_factor[i] = p; p = sc_mul(p, y_inv); } }
    size_t len = nm; bool first = true;
    while (len != 1) {
        len /= 2;
        sc c_L = inner(a, b, 0, len, len), c_R = inner(a, b, len, 0, len);
        std::vector<sc> sv; std::vector<ge> pv;
        sv.reserve(2 * len + 1); pv.reserve(2 * len + 1);
        for (size_t i = 0; i < len; i++) { sv.push_back(a[i]); pv.push_back(G[len + i]); }
        for (size_t i = 0; i < len; i++) { sv.push_back(first ? sc_mul(b[len + i], h_factor[i]) : b[len + i]); pv.push_back(Hh[i]); }
        sv.push_back(c_L); pv.push_back(Q);
        ge Lp = msm(sv, pv);
        sv.clear(); pv.clear();
        for (size_t i = 0; i < len; i++) { sv.push_back(a[len + i]); pv.push_back(G[i]); }
        for (size_t i = 0; i < len; i++) { sv.push_back(first ? sc_mul(b[i], h_factor[len + i]) : b[i]); pv.push_back(Hh[len + i]); }
        sv.push_back(c_R); pv.push_back(Q);
        ge Rp = msm(sv, pv);
        u8 Lb[32], Rb[32]; encode(Lb, Lp); encode(Rb, Rp);
        t.append_point("L", Lb); t.append_point("R", Rb);
        put(out, Lb, 32); put(out, Rb, 32);
        sc u = t.challenge_scalar("u"), u_inv = sc_invert(u);
        for (size_t i = 0; i < len; i++) {
            a[i] = sc_add(sc_mul(a[i], u), sc_mul(u_inv, a[len + i]));
            b[i] = sc_add(sc_mul(b[i], u_inv), sc_mul(u, b[len + i]));
            G[i] = mul2(u_inv, G[i], u, G[len + i]);
            if (first) Hh[i] = mul2(sc_mul(u, h_factor[i]), Hh[i], sc_mul(u_inv, h_factor[len + i]), Hh[len + i]);
            else Hh[i] = mul2(u, Hh[i], u_inv, Hh[len + i]);
        }
        first = false;
    }
    put_sc(out, a[0]); put_sc(out, b[0]);
}

// ---- transaction builder ---------------------------------------------------------------------------
bytes32 unsigned_tx::hash() const { bytes32 h; blake3_hash(h.data(), preimage.data(), preimage.size()); return h; }

std::vector<u8> unsigned_tx::sign(const keypair &kp, drbg &rng) const {
    // signature over to_bytes() including multisig entries (builder.rs:180-185,203-205)
    std::vector<u8> msg = preimage;
    for (auto &e : multisig) { msg.push_back(e.first); put(msg, e.second.data(), 64); }
    bytes64 sig = host::sign(kp, msg.data(), msg.size(), rng);
    std::vector<u8> rec = header;
    rec[1] = has_multisig ? 1 : 0;
    rec[6] = (u8)multisig.size(); rec[7] = (u8)(multisig.size() >> 8);
    put(rec, msg.data(), msg.size());
    put(rec, sig.data(), 64);
    return rec;
}

// builder.rs:261-294
static u64 transaction_cost(const builder_in &in, const bytes32 &asset) {
    u64 cost = 0;
    if (asset == NATIVE_ASSET) cost += in.fee;
    if (in.type == 0) { for (auto &tr : in.transfers) if (tr.asset == asset) cost += tr.amount; }
    else if (in.type == 1) { if (in.burn_asset == asset) cost += in.burn_amount; }
    else if (in.type == 3) { for (auto &ca : in.contract_assets) if (ca.first == asset) cost += ca.second; }
    return cost;
}
// builder.rs:296-318 (order: native first, then first appearance; the reference's HashSet order is arbitrary)
static std::vector<bytes32> used_assets(const builder_in &in) {
    std::vector<bytes32> v = {NATIVE_ASSET};
    auto add = [&](const bytes32 &a) { if (std::find(v.begin(), v.end(), a) == v.end()) v.push_back(a); };
    if (in.type == 0) for (auto &tr : in.transfers) add(tr.asset);
    else if (in.type == 1) add(in.burn_asset);
    else if (in.type == 3) for (auto &ca : in.contract_assets) add(ca.first);
    return v;
}

int build_unsigned(const builder_in &in, const keypair &source, drbg &rng, unsigned_tx &out, std::vector<asset_balance> *new_balances, pending_range_proof *defer) {
    struct transfer_c { ge C, Ds, Dr, dest; sc opening; u8 Cb[32], Dsb[32], Drb[32]; };
    std::vector<transfer_c> tc(in.type == 0 ? in.transfers.size() : 0);
    for (size_t i = 0; i < tc.size(); i++) {
        if (!ristretto_decode(tc[i].dest, in.transfers[i].dest.data())) return BUILD_ERR_DECOMPRESSION;
        tc[i].opening = rng.scalar();
        tc[i].C = commit(in.transfers[i].amount, tc[i].opening);
        tc[i].Ds = scalarmul(tc[i].opening, source.pub);
        tc[i].Dr = scalarmul(tc[i].opening, tc[i].dest);
        encode(tc[i].Cb, tc[i].C); encode(tc[i].Dsb, tc[i].Ds); encode(tc[i].Drb, tc[i].Dr);
    }
    // verify.rs:146-158
    transcript t; t.init("transaction-proof");
    t.append_u64("version", in.version);
    t.append_message("source_pubkey", source.pub_bytes.data(), 32);
    t.append_u64("fee", in.fee); t.append_u64("nonce", in.nonce);

    std::vector<bytes32> assets = used_assets(in);
    std::vector<u64> rp_values; std::vector<sc> rp_openings;
    std::vector<u8> commitments_bytes;
    if (new_balances) new_balances->clear();
    for (auto &asset : assets) {
        const asset_balance *bal = nullptr;
        for (auto &b : in.balances) if (b.asset == asset) bal = &b;
        if (!bal) return BUILD_ERR_STATE;
        u64 cost = transaction_cost(in, asset);
        if (bal->balance < cost) return BUILD_ERR_INSUFFICIENT_FUNDS;
        u64 new_balance = bal->balance - cost;
        sc new_opening = rng.scalar();
        ciphertext cur;
        if (!ct_from_bytes(cur, bal->ct.data())) return BUILD_ERR_DECOMPRESSION;
        ge new_commitment = commit(new_balance, new_opening);
        u8 ncb[32]; encode(ncb, new_commitment);
        // builder.rs:222-258: subtract fee / transfers / burn / contract deposits from the current ciphertext
        ciphertext nct = cur;
        if (asset == NATIVE_ASSET) nct.C = ge_sub(nct.C, mul_G(sc_from_u64(in.fee)));
        if (in.type == 0) {
            for (size_t i = 0; i < tc.size(); i++) if (in.transfers[i].asset == asset) { nct.C = ge_sub(nct.C, tc[i].C); nct.D = ge_sub(nct.D, tc[i].Ds); }
        } else if (in.type == 1) {
            if (in.burn_asset == asset) nct.C = ge_sub(nct.C, mul_G(sc_from_u64(in.burn_amount)));
        } else if (in.type == 3) {
            for (auto &ca : in.contract_assets) if (ca.first == asset) nct.C = ge_sub(nct.C, mul_G(sc_from_u64(ca.second)));
        }
        t.append_message("dom-sep", (const u8 *)"new-commitment-proof", 20);
        t.append_message("new_source_commitment_asset", asset.data(), 32);
        t.append_message("new_source_commitment", ncb, 32);
        put(commitments_bytes, asset.data(), 32); put(commitments_bytes, ncb, 32);
        commitment_eq_prove(commitments_bytes, source, nct, new_opening, new_balance, t, rng);
        rp_values.push_back(new_balance); rp_openings.push_back(new_opening);
        if (new_balances) { asset_balance nb; nb.asset = asset; nb.balance = new_balance; nb.ct = ct_bytes(nct); new_balances->push_back(nb); }
    }

    std::vector<u8> payload; std::vector<u32> extra_lens;
    u32 n_items = 0, opaque_len = 0;
    if (in.type == 0) {
        n_items = (u32)tc.size();
        for (size_t i = 0; i < tc.size(); i++) {
            const transfer_in &tr = in.transfers[i];
            t.append_message("dom-sep", (const u8 *)"transfer-proof", 14);
            t.append_message("dest_pubkey", tr.dest.data(), 32);
            t.append_message("amount_commitment", tc[i].Cb, 32);
            t.append_message("amount_sender_handle", tc[i].Dsb, 32);
            t.append_message("amount_receiver_handle", tc[i].Drb, 32);
            put(payload, tr.asset.data(), 32); put(payload, tr.dest.data(), 32);
            put(payload, tc[i].Cb, 32); put(payload, tc[i].Dsb, 32); put(payload, tc[i].Drb, 32);
            u32 extra = 0;
            if (!tr.extra_data.empty()) {
                // extra_data.rs:70-98: the bytes only enter the signed pre-image; the stream cipher lives in the
                // wallet layer, here the caller's bytes are carried with two fresh handles
                sc r = rng.scalar();
                put(payload, tr.extra_data.data(), tr.extra_data.size());
                put_pt(payload, scalarmul(r, source.pub)); put_pt(payload, scalarmul(r, tc[i].dest));
                extra = (u32)tr.extra_data.size() + 64;
            }
            extra_lens.push_back(extra);
            ct_validity_prove(payload, tc[i].dest, source.pub, tr.amount, tc[i].opening, t, rng);
            rp_values.push_back(tr.amount); rp_openings.push_back(tc[i].opening);
        }
    } else if (in.type == 1) {
        t.append_message("dom-sep", (const u8 *)"burn-proof", 10);
        t.append_message("asset", in.burn_asset.data(), 32);
        t.append_u64("amount", in.burn_amount);
        put(payload, in.burn_asset.data(), 32); put_be64(payload, in.burn_amount);
    } else if (in.type == 2) {
        // builder.rs:488-509
        if (in.threshold > in.signers.size() || (!in.signers.empty() && in.threshold == 0)) return BUILD_ERR_FORMAT;
        t.append_message("dom-sep", (const u8 *)"multisig-proof", 14);
        t.append_u64("threshold", in.threshold);
        for (size_t i = 0; i < in.signers.size(); i++) {
            if (in.signers[i] == source.pub_bytes) return BUILD_ERR_FORMAT;
            for (size_t j = 0; j < in.signers.size(); j++) if (i != j && in.signers[i] == in.signers[j]) return BUILD_ERR_FORMAT;
            t.append_message("signer", in.signers[i].data(), 32);
        }
        n_items = (u32)in.signers.size();
        payload.push_back(in.threshold);
        for (auto &s : in.signers) put(payload, s.data(), 32);
    } else if (in.type == 3) {
        n_items = (u32)in.contract_assets.size(); opaque_len = (u32)in.opaque.size();
        put(payload, in.contract.data(), 32);
        for (auto &ca : in.contract_assets) { put(payload, ca.first.data(), 32); put_be64(payload, ca.second); }
        put(payload, in.opaque.data(), in.opaque.size());
    } else {
        opaque_len = (u32)in.opaque.size();
        put(payload, in.opaque.data(), in.opaque.size());
    }

    // builder.rs:512-533: pad to a power of two with (0, 0) and prove
    size_t m = 1; while (m < rp_values.size()) m <<= 1;
    while (rp_values.size() < m) { rp_values.push_back(0); rp_openings.push_back(sc_zero()); }
    std::vector<u8> rp;
    if (defer) {
        // the proof is produced elsewhere (xhe_range_proofs_prove) from exactly the state the CPU prover would start from: the
        // transcript after the value commitments and the generator's counter; the counter then skips the proof's draws
        // (a_blinding, s_blinding, s_L and s_R per position, t1_blinding, t2_blinding)
        range_proof_begin(t, rp_values, rp_openings);
        defer->t = t; defer->values = rp_values; defer->openings = rp_openings;
        memcpy(defer->seed, rng.seed, 32); defer->ctr = rng.ctr;
        rng.ctr += 2 * 64 * (u64)m + 4;
        size_t lg = 6; while (((size_t)1 << lg) < 64 * m) lg++;
        rp.assign(32 * (9 + 2 * lg), 0);
    } else {
        range_proof_prove(rp, t, rp_values, rp_openings, rng);
    }

    out.header.clear();
    out.header.push_back(in.type); out.header.push_back(0);
    put_le16(out.header, n_items); put_le16(out.header, (u32)assets.size()); put_le16(out.header, 0);
    put_le32(out.header, opaque_len); put_le16(out.header, (u32)rp.size()); put_le16(out.header, 0);
    for (u32 e : extra_lens) put_le32(out.header, e);
    out.preimage.clear();
    out.preimage.push_back(in.version);
    put(out.preimage, source.pub_bytes.data(), 32);
    put_be64(out.preimage, in.fee); put_be64(out.preimage, in.nonce);
    put(out.preimage, payload.data(), payload.size());
    out.rp_off = out.preimage.size(); out.rp_len = rp.size();
    put(out.preimage, rp.data(), rp.size());
    put(out.preimage, commitments_bytes.data(), commitments_bytes.size());
    out.multisig.clear(); out.has_multisig = false;
    return BUILD_OK;
}

}  // namespace host
}  // namespace xhe
