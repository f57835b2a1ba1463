// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under xelis-he_amd/ may include, link or call this.
//
// Verifier side of the proofs on the batch-verification path:
//   * BatchCollector + sigma-proof pre_verify      /root/reference/src/proofs.rs:40-67,134-211,281-361
//   * Schnorr signature verify                      /root/reference/src/elgamal.rs:38-65
//   * Bulletproofs generators and aggregated range-proof verification (single and batched), as
//     called at /root/reference/src/tx/verify.rs:504-514,531-539 and /root/reference/src/proofs.rs:19-22.
// The range-proof code lives in the un-vendored bulletproofs 5.2.0 fork; this restates the published
// dalek-bulletproofs verifier (mega-check equation, transcript labels, generator chain).
#pragma once
#include <vector>
#include "xo_ge.h"
#include "xo_hash.h"

namespace xo {

enum {
    XO_OK = 0, XO_ERR_STATE = 1, XO_ERR_INVALID_NONCE = 2, XO_ERR_SIGNATURE = 3, XO_ERR_DECOMPRESSION = 4,
    XO_ERR_COMMITMENT_EQ = 5, XO_ERR_CT_VALIDITY = 6, XO_ERR_GENERIC_PROOF = 7, XO_ERR_RANGE_PROOF = 8,
    XO_ERR_TRANSCRIPT_IDENTITY = 9, XO_ERR_FORMAT = 10, XO_ERR_MALFORMED = 11
};

// deterministic stand-in for rand::rng(): SHAKE256(seed || counter) -> wide-reduced scalars
struct drbg {
    u8 seed[32]; u64 ctr;
    void init(const u8 s[32]) { memcpy(seed, s, 32); ctr = 0; }
    sc next_scalar() {
        sponge sp = shake256_init();
        sp.absorb(seed, 32);
        u8 c[8]; for (int i = 0; i < 8; i++) c[i] = (u8)(ctr >> (8 * i));
        ctr++;
        sp.absorb(c, 8);
        u8 buf[64]; sp.squeeze(buf, 64);
        return sc_from_bytes_mod_order_wide(buf);
    }
};

// proofs.rs:40-46
struct batch_collector {
    std::vector<sc> dynamic_scalars;
    std::vector<ge> dynamic_points;
    sc g_scalar = sc_zero(), h_scalar = sc_zero();
    // proofs.rs:49-67
    bool verify() const {
        std::vector<sc> s = dynamic_scalars; std::vector<ge> p = dynamic_points;
        s.push_back(g_scalar); p.push_back(ge_basepoint());
        s.push_back(h_scalar); p.push_back(ge_pedersen_h());
        return ge_ristretto_is_identity(ge_msm_vartime(s, p));
    }
};

// CommitmentEqProof wire: Y_0,Y_1,Y_2,z_s,z_x,z_r (proofs.rs:213-222). Returns XO_OK or an error code.
static inline int commitment_eq_pre_verify(const u8 proof[192], const ge &P_source, const ge &C_source, const ge &D_source,
                                           const ge &C_destination, transcript &t, batch_collector &bc, drbg &rng) {
    const u8 *Y0b = proof, *Y1b = proof + 32, *Y2b = proof + 64;
    sc z_s = sc_from_bytes_raw(proof + 96), z_x = sc_from_bytes_raw(proof + 128), z_r = sc_from_bytes_raw(proof + 160);
    t.append_message("dom-sep", (const u8 *)"equality-proof", 14);
    if (!t.validate_and_append_point("Y_0", Y0b)) return XO_ERR_TRANSCRIPT_IDENTITY;
    if (!t.validate_and_append_point("Y_1", Y1b)) return XO_ERR_TRANSCRIPT_IDENTITY;
    if (!t.validate_and_append_point("Y_2", Y2b)) return XO_ERR_TRANSCRIPT_IDENTITY;
    sc c = t.challenge_scalar("c");
    t.append_scalar("z_s", z_s); t.append_scalar("z_x", z_x); t.append_scalar("z_r", z_r);
    sc w = t.challenge_scalar("w");
    sc ww = sc_mul(w, w);
    sc w_neg = sc_neg(w), ww_neg = sc_neg(ww);
    ge Y0, Y1, Y2;
    if (!ristretto_decode(Y0, Y0b)) return XO_ERR_COMMITMENT_EQ;
    if (!ristretto_decode(Y1, Y1b)) return XO_ERR_COMMITMENT_EQ;
    if (!ristretto_decode(Y2, Y2b)) return XO_ERR_COMMITMENT_EQ;
    sc bf = rng.next_scalar();
    bc.g_scalar = sc_add(bc.g_scalar, sc_mul(sc_add(sc_mul(w, z_x), sc_mul(ww, z_x)), bf));
    bc.h_scalar = sc_add(bc.h_scalar, sc_mul(sc_add(sc_neg(c), sc_mul(ww, z_r)), bf));
    sc scal[7] = {z_s, sc_neg(sc_one()), sc_mul(w, z_s), sc_mul(w_neg, c), w_neg, sc_mul(ww_neg, c), ww_neg};
    const ge *pts[7] = {&P_source, &Y0, &D_source, &C_source, &Y1, &C_destination, &Y2};
    for (int i = 0; i < 7; i++) { bc.dynamic_scalars.push_back(sc_mul(scal[i], bf)); bc.dynamic_points.push_back(*pts[i]); }
    return XO_OK;
}

// CiphertextValidityProof wire: Y_0,Y_1,Y_2,z_r,z_x (proofs.rs:363-371)
static inline int ct_validity_pre_verify(const u8 proof[160], const ge &C, const ge &P_dest, const ge &P_source,
                                         const ge &D_dest, const ge &D_source, transcript &t, batch_collector &bc, drbg &rng) {
    const u8 *Y0b = proof, *Y1b = proof + 32, *Y2b = proof + 64;
    sc z_r = sc_from_bytes_raw(proof + 96), z_x = sc_from_bytes_raw(proof + 128);
    t.append_message("dom-sep", (const u8 *)"validity-proof", 14);
    if (!t.validate_and_append_point("Y_0", Y0b)) return XO_ERR_TRANSCRIPT_IDENTITY;
    if (!t.validate_and_append_point("Y_1", Y1b)) return XO_ERR_TRANSCRIPT_IDENTITY;
    if (!t.validate_and_append_point("Y_2", Y2b)) return XO_ERR_TRANSCRIPT_IDENTITY;
    sc c = t.challenge_scalar("c");
    t.append_scalar("z_r", z_r); t.append_scalar("z_x", z_x);
    sc w = t.challenge_scalar("w");
    sc w_neg = sc_neg(w);
    ge Y0, Y1, Y2;
    if (!ristretto_decode(Y0, Y0b)) return XO_ERR_CT_VALIDITY;
    if (!ristretto_decode(Y1, Y1b)) return XO_ERR_CT_VALIDITY;
    if (!ristretto_decode(Y2, Y2b)) return XO_ERR_CT_VALIDITY;
    sc bf = rng.next_scalar();
    bc.g_scalar = sc_add(bc.g_scalar, sc_mul(z_x, bf));
    bc.h_scalar = sc_add(bc.h_scalar, sc_mul(z_r, bf));
    sc w_z_r = sc_mul(w, z_r), w_neg_c = sc_mul(w_neg, c);
    sc scal[8] = {sc_neg(c), sc_neg(sc_one()), w_z_r, w_neg_c, w_neg, sc_mul(w, w_z_r), sc_mul(w, w_neg_c), sc_mul(w, w_neg)};
    const ge *pts[8] = {&C, &Y0, &P_dest, &D_dest, &Y1, &P_source, &D_source, &Y2};
    for (int i = 0; i < 8; i++) { bc.dynamic_scalars.push_back(sc_mul(scal[i], bf)); bc.dynamic_points.push_back(*pts[i]); }
    return XO_OK;
}

// elgamal.rs:38-65: r = s*H + (-e)*P ; e ?= SHA3-512(pk || msg || enc(r)) mod l
static inline bool signature_verify(const u8 sig[64], const u8 *msg, size_t msg_len, const ge &key, const u8 key_bytes[32]) {
    sc s = sc_from_bytes_raw(sig), e = sc_from_bytes_raw(sig + 32);
    ge r = ge_add(ge_scalarmul(s, ge_pedersen_h()), ge_scalarmul(sc_neg(e), key));
    u8 renc[32]; ristretto_encode(renc, r);
    sponge sp; sp.init(72, 0x06);
    sp.absorb(key_bytes, 32); sp.absorb(msg, msg_len); sp.absorb(renc, 32);
    u8 h[64]; sp.squeeze(h, 64);
    return sc_eq(e, sc_from_bytes_mod_order_wide(h));
}

// BulletproofGens::new(64, 512) (proofs.rs:20), materialised lazily per party share.
// share j, generator i: i-th 64-byte block of SHAKE256("GeneratorsChain" || 'G'|'H' || LE32(j)) -> one-way map
struct bp_gens_t {
    std::vector<ge> G, H;     // flat index 64*j + i
    void ensure(size_t n_points);
};
bp_gens_t &bp_gens();

static inline sc sc_sum_of_powers(const sc &x, size_t n) {   // 1 + x + ... + x^(n-1)
    sc acc = sc_zero(), p = sc_one();
    for (size_t i = 0; i < n; i++) { acc = sc_add(acc, p); p = sc_mul(p, x); }
    return acc;
}

// One aggregated range proof reduced to the terms of its mega-check (dalek-bulletproofs
// RangeProof::verify_multiple): dynamic (scalar, point) pairs plus scalars for the static generators.
struct rp_terms {
    std::vector<sc> dyn_scalars; std::vector<ge> dyn_points;   // A,S,T1,T2,L..,R..,V..
    std::vector<sc> g_scalars, h_scalars;                      // nm each
    sc b_scalar, b_blinding_scalar;                            // for B = basepoint, B_blinding = H
};

// proof bytes: A,S,T_1,T_2,t_x,t_x_blinding,e_blinding,(L_i,R_i)*lg,a,b.  commitments: m 32-byte encodings
// with their decoded points (identity allowed).  `weight` is the random factor applied to the T-check
// (the verifier's `c`).  Returns XO_OK or XO_ERR_RANGE_PROOF.
static inline int range_proof_terms(rp_terms &out, const u8 *proof, size_t proof_len, const std::vector<const u8 *> &V_bytes,
                                    const std::vector<ge> &V_points, transcript &t, const sc &weight) {
    const size_t n = 64, m = V_bytes.size();
    if (proof_len % 32 != 0 || proof_len < 9 * 32) return XO_ERR_RANGE_PROOF;
    size_t lg = (proof_len / 32 - 9) / 2;
    if ((proof_len / 32 - 9) % 2 != 0 || lg >= 32) return XO_ERR_RANGE_PROOF;
    if (m == 0 || m > 512 || (m & (m - 1))) return XO_ERR_RANGE_PROOF;   // party capacity / power of two
    const u8 *Ab = proof, *Sb = proof + 32, *T1b = proof + 64, *T2b = proof + 96;
    sc t_x = sc_from_bytes_raw(proof + 128), t_x_bl = sc_from_bytes_raw(proof + 160), e_bl = sc_from_bytes_raw(proof + 192);
    const u8 *LR = proof + 224;
    sc a = sc_from_bytes_raw(LR + 64 * lg), b = sc_from_bytes_raw(LR + 64 * lg + 32);

    t.append_message("dom-sep", (const u8 *)"rangeproof v1", 13);
    t.append_u64("n", n); t.append_u64("m", m);
    for (size_t j = 0; j < m; j++) t.append_point("V", V_bytes[j]);
    if (!t.validate_and_append_point("A", Ab)) return XO_ERR_RANGE_PROOF;
    if (!t.validate_and_append_point("S", Sb)) return XO_ERR_RANGE_PROOF;
    sc y = t.challenge_scalar("y"), z = t.challenge_scalar("z");
    sc zz = sc_mul(z, z), minus_z = sc_neg(z);
    if (!t.validate_and_append_point("T_1", T1b)) return XO_ERR_RANGE_PROOF;
    if (!t.validate_and_append_point("T_2", T2b)) return XO_ERR_RANGE_PROOF;
    sc x = t.challenge_scalar("x");
    t.append_scalar("t_x", t_x); t.append_scalar("t_x_blinding", t_x_bl); t.append_scalar("e_blinding", e_bl);
    sc w = t.challenge_scalar("w");
    const sc &c = weight;

    // inner-product verification scalars
    size_t nm = n * m;
    if (nm != ((size_t)1 << lg)) return XO_ERR_RANGE_PROOF;
    t.append_message("dom-sep", (const u8 *)"ipp v1", 6);
    t.append_u64("n", nm);
    std::vector<sc> u(lg), u_inv(lg);
    for (size_t i = 0; i < lg; i++) {
        if (!t.validate_and_append_point("L", LR + 64 * i)) return XO_ERR_RANGE_PROOF;
        if (!t.validate_and_append_point("R", LR + 64 * i + 32)) return XO_ERR_RANGE_PROOF;
        u[i] = t.challenge_scalar("u");
    }
    // Scalar::batch_invert: one inversion for all challenges (Montgomery's trick)
    sc allinv;
    {
        std::vector<sc> pre(lg + 1);
        pre[0] = sc_one();
        for (size_t i = 0; i < lg; i++) pre[i + 1] = sc_mul(pre[i], u[i]);
        sc inv = sc_invert(pre[lg]);
        allinv = inv;
        for (size_t i = lg; i-- > 0;) { u_inv[i] = sc_mul(inv, pre[i]); inv = sc_mul(inv, u[i]); }
    }
    std::vector<sc> u_sq(lg), u_inv_sq(lg);
    for (size_t i = 0; i < lg; i++) { u_sq[i] = sc_mul(u[i], u[i]); u_inv_sq[i] = sc_mul(u_inv[i], u_inv[i]); }
    std::vector<sc> s(nm);
    s[0] = allinv;
    for (size_t i = 1; i < nm; i++) {
        int lg_i = 63 - __builtin_clzll((u64)i);
        size_t k = (size_t)1 << lg_i;
        s[i] = sc_mul(s[i - k], u_sq[(lg - 1) - lg_i]);
    }

    // decode dynamic points
    ge A, S, T1, T2;
    if (!ristretto_decode(A, Ab) || !ristretto_decode(S, Sb) || !ristretto_decode(T1, T1b) || !ristretto_decode(T2, T2b))
        return XO_ERR_RANGE_PROOF;
    std::vector<ge> Ls(lg), Rs(lg);
    for (size_t i = 0; i < lg; i++)
        if (!ristretto_decode(Ls[i], LR + 64 * i) || !ristretto_decode(Rs[i], LR + 64 * i + 32)) return XO_ERR_RANGE_PROOF;

    out.g_scalars.resize(nm); out.h_scalars.resize(nm);
    sc y_inv = sc_invert(y), exp_y_inv = sc_one();
    sc two = sc_from_u64(2);
    sc exp_z = sc_one();
    size_t idx = 0;
    for (size_t j = 0; j < m; j++) {
        sc exp_2 = sc_one();
        for (size_t k = 0; k < n; k++, idx++) {
            out.g_scalars[idx] = sc_sub(minus_z, sc_mul(a, s[idx]));
            sc z_and_2 = sc_mul(exp_2, exp_z);
            out.h_scalars[idx] = sc_add(z, sc_mul(exp_y_inv, sc_sub(sc_mul(zz, z_and_2), sc_mul(b, s[nm - 1 - idx]))));
            exp_2 = sc_mul(exp_2, two);
            exp_y_inv = sc_mul(exp_y_inv, y_inv);
        }
        exp_z = sc_mul(exp_z, z);
    }
    // delta(n, m, y, z) = (z - z^2) * sum y^i - z^3 * sum 2^i * sum z^j
    sc sum_y = sc_sum_of_powers(y, nm), sum_2 = sc_sum_of_powers(two, n), sum_z = sc_sum_of_powers(z, m);
    sc delta = sc_sub(sc_mul(sc_sub(z, zz), sum_y), sc_mul(sc_mul(sc_mul(zz, z), sum_2), sum_z));
    out.b_scalar = sc_add(sc_mul(w, sc_sub(t_x, sc_mul(a, b))), sc_mul(c, sc_sub(delta, t_x)));
    out.b_blinding_scalar = sc_sub(sc_neg(e_bl), sc_mul(c, t_x_bl));

    out.dyn_scalars.clear(); out.dyn_points.clear();
    out.dyn_scalars.push_back(sc_one()); out.dyn_points.push_back(A);
    out.dyn_scalars.push_back(x); out.dyn_points.push_back(S);
    out.dyn_scalars.push_back(sc_mul(c, x)); out.dyn_points.push_back(T1);
    out.dyn_scalars.push_back(sc_mul(sc_mul(c, x), x)); out.dyn_points.push_back(T2);
    for (size_t i = 0; i < lg; i++) { out.dyn_scalars.push_back(u_sq[i]); out.dyn_points.push_back(Ls[i]); }
    for (size_t i = 0; i < lg; i++) { out.dyn_scalars.push_back(u_inv_sq[i]); out.dyn_points.push_back(Rs[i]); }
    exp_z = sc_one();
    for (size_t j = 0; j < m; j++) {
        out.dyn_scalars.push_back(sc_mul(sc_mul(c, zz), exp_z)); out.dyn_points.push_back(V_points[j]);
        exp_z = sc_mul(exp_z, z);
    }
    return XO_OK;
}

// RangeProof::verify_multiple (verify.rs:531-539): one proof, one mega-check
static inline int range_proof_verify_single(const u8 *proof, size_t proof_len, const std::vector<const u8 *> &V_bytes,
                                            const std::vector<ge> &V_points, transcript &t, drbg &rng) {
    rp_terms T;
    int rc = range_proof_terms(T, proof, proof_len, V_bytes, V_points, t, rng.next_scalar());
    if (rc) return rc;
    size_t nm = T.g_scalars.size();
    bp_gens_t &gens = bp_gens(); gens.ensure(nm);
    std::vector<sc> s = T.dyn_scalars; std::vector<ge> p = T.dyn_points;
    s.push_back(T.b_blinding_scalar); p.push_back(ge_pedersen_h());
    s.push_back(T.b_scalar); p.push_back(ge_basepoint());
    for (size_t i = 0; i < nm; i++) { s.push_back(T.g_scalars[i]); p.push_back(gens.G[i]); }
    for (size_t i = 0; i < nm; i++) { s.push_back(T.h_scalars[i]); p.push_back(gens.H[i]); }
    return ge_ristretto_is_identity(ge_msm_vartime(s, p)) ? XO_OK : XO_ERR_RANGE_PROOF;
}

// RangeProof::verify_batch over verification views (verify.rs:504-513): every proof's equation is
// multiplied by a fresh random weight; scalars of the shared static generators are summed.
struct rp_batch {
    std::vector<sc> dyn_scalars; std::vector<ge> dyn_points;
    std::vector<sc> g_scalars, h_scalars;
    sc b_scalar = sc_zero(), b_blinding_scalar = sc_zero();
    void add(const rp_terms &T, const sc &rho) {
        for (size_t i = 0; i < T.dyn_scalars.size(); i++) { dyn_scalars.push_back(sc_mul(T.dyn_scalars[i], rho)); dyn_points.push_back(T.dyn_points[i]); }
        if (g_scalars.size() < T.g_scalars.size()) { g_scalars.resize(T.g_scalars.size(), sc_zero()); h_scalars.resize(T.h_scalars.size(), sc_zero()); }
        for (size_t i = 0; i < T.g_scalars.size(); i++) {
            g_scalars[i] = sc_add(g_scalars[i], sc_mul(T.g_scalars[i], rho));
            h_scalars[i] = sc_add(h_scalars[i], sc_mul(T.h_scalars[i], rho));
        }
        b_scalar = sc_add(b_scalar, sc_mul(T.b_scalar, rho));
        b_blinding_scalar = sc_add(b_blinding_scalar, sc_mul(T.b_blinding_scalar, rho));
    }
    bool verify() const {
        size_t nm = g_scalars.size();
        bp_gens_t &gens = bp_gens(); gens.ensure(nm);
        std::vector<sc> s = dyn_scalars; std::vector<ge> p = dyn_points;
        s.push_back(b_blinding_scalar); p.push_back(ge_pedersen_h());
        s.push_back(b_scalar); p.push_back(ge_basepoint());
        for (size_t i = 0; i < nm; i++) { s.push_back(g_scalars[i]); p.push_back(gens.G[i]); }
        for (size_t i = 0; i < nm; i++) { s.push_back(h_scalars[i]); p.push_back(gens.H[i]); }
        return ge_ristretto_is_identity(ge_msm_vartime(s, p));
    }
};

}  // namespace xo
