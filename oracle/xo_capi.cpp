// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under xelis-he_amd/ may include, link or call this.
//
// C entry points of the CPU oracle (liboracle.so), loaded with ctypes by tests/, by
// __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  The oracle restates the reference's
// batch-verification path (/root/reference/src/tx/verify.rs:487-517 and what it calls); the
// reference itself is Rust with three un-vendored git dependencies and cannot be built in this
// image, so parity with it is UNPINNED at byte level; the oracle is pinned by RFC 9496 vectors,
// the published Pedersen blinding generator, hashlib SHA3/SHAKE, Merlin and BLAKE3 vectors and a
// big-integer Python spec (oracle/spec_ristretto.py).
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <thread>
#include <mutex>
#include <string>
#include <unordered_map>
#include "xo_tx.h"

namespace xo {

const fe_consts &fe_constants() {
    static fe_consts K;
    static bool init = false;
    if (!init) {
        // two-phase: sqrt_m1 first (fe_sqrt_ratio_m1 reads it)
        // d = -121665/121666
        K.d = fe_mul(fe_neg(fe_small(121665)), fe_invert(fe_small(121666)));
        K.d2 = fe_add(K.d, K.d);
        // sqrt(-1) = 2^((p-1)/4): 2^(2^253 - 5) = (2^(2^252-3))^2 * 2
        fe two = fe_small(2);
        K.sqrt_m1 = fe_mul(fe_sq(fe_pow22523(two)), two);
        init = true;   // the calls below use K.sqrt_m1 through fe_constants()
        fe minus_one = fe_neg(fe_one());
        fe_sqrt_ratio_m1(K.invsqrt_a_minus_d, fe_one(), fe_sub(minus_one, K.d));     // 1/sqrt(a-d), a=-1
        fe r; fe_sqrt_ratio_m1(r, fe_sub(fe_neg(K.d), fe_one()), fe_one());           // sqrt(a*d-1)
        K.sqrt_ad_minus_one = fe_neg(r);                                              // RFC 9496 lists the odd root
        K.one_minus_d_sq = fe_sub(fe_one(), fe_sq(K.d));
        K.d_minus_one_sq = fe_sq(fe_sub(K.d, fe_one()));
    }
    return K;
}

const sc_consts &sc_constants() {
    static sc_consts K;
    static bool init = false;
    if (!init) {
        sc r = sc_one();
        for (int i = 0; i < 256; i++) r = sc_add(r, r);
        K.R = r;
        for (int i = 0; i < 256; i++) r = sc_add(r, r);
        K.RR = r;
        for (int i = 0; i < 256; i++) r = sc_add(r, r);
        K.RRR = r;
        init = true;
    }
    return K;
}

static const u8 BASEPOINT_BYTES[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                       0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
const ge &ge_basepoint() {
    static ge B; static bool init = false;
    if (!init) { if (!ristretto_decode(B, BASEPOINT_BYTES)) abort(); init = true; }
    return B;
}
const ge &ge_pedersen_h() {
    static ge H; static bool init = false;
    if (!init) { u8 h[64]; sha3_512(h, BASEPOINT_BYTES, 32); H = ristretto_from_uniform_bytes(h); init = true; }
    return H;
}

void bp_gens_t::ensure(size_t n_points) {
    size_t have = G.size();
    if (have >= n_points) return;
    size_t parties = (n_points + 63) / 64;
    G.resize(parties * 64); H.resize(parties * 64);
    for (size_t j = have / 64; j < parties; j++) {
        for (int which = 0; which < 2; which++) {
            sponge sp = shake256_init();
            sp.absorb((const u8 *)"GeneratorsChain", 15);
            u8 label[5] = {(u8)(which ? 'H' : 'G'), (u8)j, (u8)(j >> 8), (u8)(j >> 16), (u8)(j >> 24)};
            sp.absorb(label, 5);
            for (int i = 0; i < 64; i++) {
                u8 ub[64]; sp.squeeze(ub, 64);
                (which ? H : G)[64 * j + i] = ristretto_from_uniform_bytes(ub);
            }
        }
    }
}
bp_gens_t &bp_gens() { static bp_gens_t g; return g; }

}  // namespace xo

using namespace xo;

static void warm() { fe_constants(); sc_constants(); ge_basepoint(); ge_pedersen_h(); }

extern "C" {

// ---- primitives, for parity tests ---------------------------------------------------------------
void xo_init(void) { warm(); }

// out_xyzt: 4 x 32 canonical field bytes (X, Y, Z=1, T); ok[i] = 1 if the encoding is valid
void xo_ristretto_decompress_batch(uint64_t n, const uint8_t *in32, uint8_t *out_xyzt128, uint8_t *ok) {
    warm();
    for (uint64_t i = 0; i < n; i++) {
        ge p; bool good = ristretto_decode(p, in32 + 32 * i);
        ok[i] = good;
        if (!good) { memset(out_xyzt128 + 128 * i, 0, 128); continue; }
        fe_tobytes(out_xyzt128 + 128 * i, p.X); fe_tobytes(out_xyzt128 + 128 * i + 32, p.Y);
        fe_tobytes(out_xyzt128 + 128 * i + 64, p.Z); fe_tobytes(out_xyzt128 + 128 * i + 96, p.T);
    }
}
void xo_ristretto_compress_batch(uint64_t n, const uint8_t *in_xyzt128, uint8_t *out32) {
    warm();
    for (uint64_t i = 0; i < n; i++) {
        ge p = {fe_frombytes(in_xyzt128 + 128 * i), fe_frombytes(in_xyzt128 + 128 * i + 32),
                fe_frombytes(in_xyzt128 + 128 * i + 64), fe_frombytes(in_xyzt128 + 128 * i + 96)};
        ristretto_encode(out32 + 32 * i, p);
    }
}
// vartime multiscalar multiplication over compressed points; returns -1 if a point fails to decode
int32_t xo_msm_vartime(uint64_t n, const uint8_t *scalars32, const uint8_t *points32, uint8_t out32[32], uint8_t *is_identity) {
    warm();
    std::vector<sc> s(n); std::vector<ge> p(n);
    for (uint64_t i = 0; i < n; i++) {
        s[i] = sc_from_bytes_mod_order(scalars32 + 32 * i);
        if (!ristretto_decode(p[i], points32 + 32 * i)) return -1;
    }
    ge r = ge_msm_vartime(s, p);
    ristretto_encode(out32, r);
    if (is_identity) *is_identity = ge_ristretto_is_identity(r);
    return 0;
}
void xo_from_uniform_bytes(const uint8_t in64[64], uint8_t out32[32]) { warm(); ristretto_encode(out32, ristretto_from_uniform_bytes(in64)); }
void xo_pedersen_h(uint8_t out32[32]) { warm(); ristretto_encode(out32, ge_pedersen_h()); }
void xo_bp_generators(uint64_t n, uint8_t *g_out32, uint8_t *h_out32) {
    warm(); bp_gens().ensure(n);
    for (uint64_t i = 0; i < n; i++) { ristretto_encode(g_out32 + 32 * i, bp_gens().G[i]); ristretto_encode(h_out32 + 32 * i, bp_gens().H[i]); }
}
void xo_sc_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { sc r = sc_mul(sc_from_bytes_mod_order(a), sc_from_bytes_mod_order(b)); sc_tobytes(out, r); }
void xo_sc_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { sc r = sc_add(sc_from_bytes_mod_order(a), sc_from_bytes_mod_order(b)); sc_tobytes(out, r); }
void xo_sc_invert(const uint8_t a[32], uint8_t out[32]) { sc r = sc_invert(sc_from_bytes_mod_order(a)); sc_tobytes(out, r); }
void xo_sc_from_wide(const uint8_t a[64], uint8_t out[32]) { sc r = sc_from_bytes_mod_order_wide(a); sc_tobytes(out, r); }
void xo_fe_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { fe_tobytes(out, fe_mul(fe_frombytes(a), fe_frombytes(b))); }
void xo_fe_invert(const uint8_t a[32], uint8_t out[32]) { fe_tobytes(out, fe_invert(fe_frombytes(a))); }
void xo_sha3_512(const uint8_t *d, uint64_t n, uint8_t out[64]) { sha3_512(out, d, n); }
void xo_sha3_256(const uint8_t *d, uint64_t n, uint8_t out[32]) { sha3_256(out, d, n); }
void xo_shake256(const uint8_t *d, uint64_t n, uint8_t *out, uint64_t outlen) { sponge s = shake256_init(); s.absorb(d, n); s.squeeze(out, outlen); }
void xo_blake3(const uint8_t *d, uint64_t n, uint8_t out[32]) { blake3_hash(out, d, n); }
void xo_keccak_f1600(uint64_t st[25]) { keccak_f1600(st); }

// Merlin transcript driven by a tiny script, for vector tests: ops is a sequence of
//   'A' label\0 u32 len, bytes     (append_message)
//   'C' label\0 u32 len            (challenge_bytes -> appended to out)
uint64_t xo_merlin_script(const char *proto_label, const uint8_t *ops, uint64_t ops_len, uint8_t *out, uint64_t out_cap) {
    transcript t; t.init(proto_label);
    uint64_t pos = 0, w = 0;
    while (pos < ops_len) {
        char op = (char)ops[pos++];
        const char *label = (const char *)(ops + pos);
        pos += strlen(label) + 1;
        u32 len = load_le32(ops + pos); pos += 4;
        if (op == 'A') { t.append_message(label, ops + pos, len); pos += len; }
        else { if (w + len > out_cap) return 0; t.challenge_bytes(label, out + w, len); w += len; }
    }
    return w;
}
int32_t xo_signature_verify(const uint8_t sig[64], const uint8_t *msg, uint64_t n, const uint8_t key[32]) {
    warm(); ge k; if (!ristretto_decode(k, key)) return -1;
    return signature_verify(sig, msg, n, k, key) ? 1 : 0;
}

// ---- the path itself -----------------------------------------------------------------------------
static bool parse_all(std::vector<tx_view> &txs, const uint8_t *blob, const uint64_t *offsets, uint64_t n, verdict *v) {
    txs.resize(n);
    for (uint64_t i = 0; i < n; i++)
        if (!tx_parse(txs[i], blob + offsets[i], offsets[i + 1] - offsets[i])) { v->accept = 0; v->err_code = XO_ERR_MALFORMED; v->first_bad_tx = (int64_t)i; return false; }
    return true;
}

// Transaction::verify_batch.  out may be NULL.  The output ledger is written whatever the verdict
// (the reference mutates its state incrementally, also when it later fails).
int32_t xo_verify_batch(const uint8_t *blob, const uint64_t *offsets, uint64_t n, const ledger_view *in, const ledger_out *out,
                        verdict *v, const uint8_t seed[32]) {
    warm();
    std::vector<tx_view> txs;
    if (!parse_all(txs, blob, offsets, n, v)) return 0;
    ledger st; st.load(*in);
    *v = verify_batch(txs, st, seed);
    if (out) st.store(*in, *out);
    return 0;
}
// Transaction::verify
int32_t xo_verify_single(const uint8_t *rec, uint64_t len, const ledger_view *in, const ledger_out *out, verdict *v, const uint8_t seed[32]) {
    warm();
    tx_view t;
    if (!tx_parse(t, rec, len)) { v->accept = 0; v->err_code = XO_ERR_MALFORMED; v->first_bad_tx = 0; return 0; }
    ledger st; st.load(*in);
    *v = verify_single(t, st, seed);
    if (out) st.store(*in, *out);
    return 0;
}
// Transaction::apply_without_verify over a batch, in order
int32_t xo_apply_batch(const uint8_t *blob, const uint64_t *offsets, uint64_t n, const ledger_view *in, const ledger_out *out) {
    warm();
    std::vector<tx_view> txs; verdict v;
    if (!parse_all(txs, blob, offsets, n, &v)) return -1;
    ledger st; st.load(*in);
    for (uint64_t i = 0; i < n; i++) if (!apply_without_verify(txs[i], st, (int64_t)i)) return -2;
    if (out) st.store(*in, *out);
    return 0;
}

// ElGamalSecretKey::decrypt (/root/reference/src/elgamal.rs:140-145): the encoding of C - s * D; -1 when a point does not decode
int32_t xo_decrypt_point(const uint8_t secret32[32], const uint8_t ct64[64], uint8_t out32[32]) {
    warm();
    ge c, d;
    if (!ristretto_decode(c, ct64) || !ristretto_decode(d, ct64 + 32)) return -1;
    ristretto_encode(out32, ge_sub(c, ge_scalarmul(sc_from_bytes_mod_order(secret32), d)));
    return 0;
}
// ECDLPInstance::decode (/root/reference/src/elgamal.rs:68-92) restated as the plainest baby-step/giant-step: the m in
// [0, 2^max_bits) with m * G = point, or -1.  The reference's own decoder (the ecdlp module of its curve25519-dalek fork) is not
// among its sources; what is restated is its contract.  Keys are canonical encodings, so every look-up is exact.
// max_bits <= 40 (2^16 baby steps, 2^(max_bits-16) giant steps of one encoding each).
int64_t xo_ecdlp_decode(const uint8_t point32[32], uint32_t max_bits) {
    warm();
    if (max_bits < 1 || max_bits > 40) return -2;
    ge p;
    if (!ristretto_decode(p, point32)) return -1;
    static std::mutex mu;
    static std::unordered_map<std::string, uint32_t> baby;
    static const uint32_t B = 16;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (baby.empty()) {
            ge q = ge_identity();
            for (uint32_t j = 0; j < (1u << B); j++) {
                u8 e[32]; ristretto_encode(e, q);
                baby.emplace(std::string((const char *)e, 32), j);
                q = ge_add(q, ge_basepoint());
            }
        }
    }
    const ge stride = ge_neg(ge_scalarmul(sc_from_u64(1ull << B), ge_basepoint()));
    const uint64_t giant = max_bits > B ? 1ull << (max_bits - B) : 1;
    ge q = p;
    for (uint64_t i = 0; i < giant; i++) {
        u8 e[32]; ristretto_encode(e, q);
        auto it = baby.find(std::string((const char *)e, 32));
        if (it != baby.end()) {
            const uint64_t m = (i << B) + it->second;
            return (m >> max_bits) ? -1 : (int64_t)m;
        }
        q = ge_add(q, stride);
    }
    return -1;
}

// The Fiat-Shamir challenges of one transaction in transcript order, for comparing the device
// transcript kernel with the oracle: per commitment (c, w), per transfer (c, w), then the range
// proof's y, z, x, w, u_1..u_lg.  The balances come from `in` (sender side only needs the ledger
// for points, which do not enter the transcript, so no ledger is needed here).  Returns the count.
int64_t xo_tx_challenges(const uint8_t *rec, uint64_t len, uint8_t *out32, uint64_t cap) {
    warm();
    tx_view t;
    if (!tx_parse(t, rec, len)) return -1;
    transcript tr;
    tr.init("transaction-proof");
    tr.append_u64("version", t.version); tr.append_message("source_pubkey", t.source, 32);
    tr.append_u64("fee", t.fee); tr.append_u64("nonce", t.nonce);
    std::vector<sc> ch;
    for (size_t i = 0; i < t.n_commitments; i++) {
        const u8 *cm = t.commitments + 256 * i;
        tr.append_message("dom-sep", (const u8 *)"new-commitment-proof", 20);
        tr.append_message("new_source_commitment_asset", cm, 32);
        tr.append_message("new_source_commitment", cm + 32, 32);
        tr.append_message("dom-sep", (const u8 *)"equality-proof", 14);
        tr.append_point("Y_0", cm + 64); tr.append_point("Y_1", cm + 96); tr.append_point("Y_2", cm + 128);
        ch.push_back(tr.challenge_scalar("c"));
        tr.append_message("z_s", cm + 160, 32); tr.append_message("z_x", cm + 192, 32); tr.append_message("z_r", cm + 224, 32);
        ch.push_back(tr.challenge_scalar("w"));
    }
    if (t.type == TX_TRANSFERS) {
        for (size_t i = 0; i < t.transfers.size(); i++) {
            const u8 *p = t.transfers[i], *pr = t.transfer_proof(i);
            tr.append_message("dom-sep", (const u8 *)"transfer-proof", 14);
            tr.append_message("dest_pubkey", p + 32, 32); tr.append_message("amount_commitment", p + 64, 32);
            tr.append_message("amount_sender_handle", p + 96, 32); tr.append_message("amount_receiver_handle", p + 128, 32);
            tr.append_message("dom-sep", (const u8 *)"validity-proof", 14);
            tr.append_point("Y_0", pr); tr.append_point("Y_1", pr + 32); tr.append_point("Y_2", pr + 64);
            ch.push_back(tr.challenge_scalar("c"));
            tr.append_message("z_r", pr + 96, 32); tr.append_message("z_x", pr + 128, 32);
            ch.push_back(tr.challenge_scalar("w"));
        }
    } else if (t.type == TX_BURN) {
        tr.append_message("dom-sep", (const u8 *)"burn-proof", 10);
        tr.append_message("asset", t.payload, 32); tr.append_u64("amount", load_be64(t.payload + 32));
    } else if (t.type == TX_MULTISIG) {
        tr.append_message("dom-sep", (const u8 *)"multisig-proof", 14);
        tr.append_u64("threshold", t.payload[0]);
        for (size_t i = 0; i < t.n_items; i++) tr.append_message("signer", t.payload + 1 + 32 * i, 32);
    }
    size_t n_commit = t.n_commitments + (t.type == TX_TRANSFERS ? t.transfers.size() : 0);
    size_t m = 1; while (m < n_commit) m <<= 1;
    size_t lg = (t.rp_len / 32 - 9) / 2;
    tr.append_message("dom-sep", (const u8 *)"rangeproof v1", 13);
    tr.append_u64("n", 64); tr.append_u64("m", m);
    for (size_t i = 0; i < t.n_commitments; i++) tr.append_point("V", t.commitments + 256 * i + 32);
    if (t.type == TX_TRANSFERS) for (size_t i = 0; i < t.transfers.size(); i++) tr.append_point("V", t.transfers[i] + 64);
    for (size_t i = n_commit; i < m; i++) tr.append_point("V", ZERO32);
    tr.append_point("A", t.rp); tr.append_point("S", t.rp + 32);
    ch.push_back(tr.challenge_scalar("y")); ch.push_back(tr.challenge_scalar("z"));
    tr.append_point("T_1", t.rp + 64); tr.append_point("T_2", t.rp + 96);
    ch.push_back(tr.challenge_scalar("x"));
    tr.append_message("t_x", t.rp + 128, 32); tr.append_message("t_x_blinding", t.rp + 160, 32); tr.append_message("e_blinding", t.rp + 192, 32);
    ch.push_back(tr.challenge_scalar("w"));
    tr.append_message("dom-sep", (const u8 *)"ipp v1", 6);
    tr.append_u64("n", 64 * m);
    for (size_t i = 0; i < lg; i++) {
        tr.append_point("L", t.rp + 224 + 64 * i); tr.append_point("R", t.rp + 224 + 64 * i + 32);
        ch.push_back(tr.challenge_scalar("u"));
    }
    if (ch.size() > cap) return -2;
    for (size_t i = 0; i < ch.size(); i++) sc_tobytes(out32 + 32 * i, ch[i]);
    return (int64_t)ch.size();
}

// CPU baseline: `threads` independent copies of the same batch, one ledger clone each, exactly
// like the reference's 8-thread bench (/root/reference/benches/tx.rs:307-325).  Returns seconds.
double xo_bench_verify_batch(const uint8_t *blob, const uint64_t *offsets, uint64_t n, const ledger_view *in, int threads, int32_t *all_ok) {
    warm();
    std::vector<tx_view> txs; verdict v0;
    if (!parse_all(txs, blob, offsets, n, &v0)) { *all_ok = 0; return 0.0; }
    { ledger st; st.load(*in); size_t nm = 0; for (auto &t : txs) { size_t lg = (t.rp_len / 32 - 9) / 2; nm = std::max(nm, (size_t)1 << lg); } bp_gens().ensure(nm); }
    std::vector<int> ok(threads, 0);
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    std::vector<std::thread> th;
    for (int k = 0; k < threads; k++)
        th.emplace_back([&, k]() {
            ledger st; st.load(*in);
            u8 seed[32]; memset(seed, k + 1, 32);
            verdict v = verify_batch(txs, st, seed);
            ok[k] = v.accept;
        });
    for (auto &t : th) t.join();
    clock_gettime(CLOCK_MONOTONIC, &b);
    *all_ok = 1; for (int k = 0; k < threads; k++) if (!ok[k]) *all_ok = 0;
    return (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
}

}  // extern "C"
