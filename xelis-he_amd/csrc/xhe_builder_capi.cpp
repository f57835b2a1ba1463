// C entry points of the transaction builder and of the synthetic-batch generator (include/xhe_builder.h).
// CPU code: it produces inputs for the verification path (keys, ciphertexts, proofs, signatures,
// ledgers) and never runs inside it.
#include <string.h>
#include "xhe_settings.h"
#include <map>
#include <thread>
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include "../../include/xhe_builder.h"
#include "xhe_builder.h"
#include "xhe_record.h"

using namespace xhe;
using namespace xhe::host;

static bytes32 b32(const u8 *p) { bytes32 r; memcpy(r.data(), p, 32); return r; }
static bytes64 b64(const u8 *p) { bytes64 r; memcpy(r.data(), p, 64); return r; }

static int fill_builder(builder_in &in, const xhe_tx_request *rq) {
    in.version = rq->version; in.type = rq->type; in.fee = rq->fee; in.nonce = rq->nonce;
    for (u32 i = 0; i < rq->n_transfers; i++) {
        transfer_in t; t.asset = b32(rq->transfer_assets + 32 * i); t.amount = rq->transfer_amounts[i]; t.dest = b32(rq->transfer_dests + 32 * i);
        if (rq->transfer_extra && rq->transfer_extra_len && rq->transfer_extra_len[i]) t.extra_data.assign(rq->transfer_extra[i], rq->transfer_extra[i] + rq->transfer_extra_len[i]);
        in.transfers.push_back(t);
    }
    in.burn_asset = b32(rq->burn_asset); in.burn_amount = rq->burn_amount;
    for (u32 i = 0; i < rq->n_signers; i++) in.signers.push_back(b32(rq->signers + 32 * i));
    in.threshold = rq->threshold;
    in.contract = b32(rq->contract);
    for (u32 i = 0; i < rq->n_contract_assets; i++) in.contract_assets.push_back({b32(rq->contract_assets + 32 * i), rq->contract_amounts[i]});
    if (rq->opaque_len) in.opaque.assign(rq->opaque, rq->opaque + rq->opaque_len);
    for (u32 i = 0; i < rq->n_balances; i++) { asset_balance b; b.asset = b32(rq->balance_assets + 32 * i); b.balance = rq->balance_amounts[i]; b.ct = b64(rq->balance_cts + 64 * i); in.balances.push_back(b); }
    return 0;
}

extern "C" {

int32_t xhe_keygen(const uint8_t secret32[32], uint8_t pub32[32]) {
    if (!secret32 || !pub32 || !rec_scalar_canonical(secret32) || rec_is_zero32(secret32)) return -1;
    keypair k = keypair::from_secret(sc_frombytes(secret32));
    memcpy(pub32, k.pub_bytes.data(), 32);
    return 0;
}
int32_t xhe_keygen_from_seed(const uint8_t seed32[32], uint8_t secret32[32], uint8_t pub32[32]) {
    drbg rng(seed32);
    sc s = rng.scalar();
    sc_tobytes(secret32, s);
    return xhe_keygen(secret32, pub32);
}
int32_t xhe_encrypt(const uint8_t pub32[32], uint64_t amount, const uint8_t seed32[32], uint8_t ct64[64]) {
    ge pub; if (!ristretto_decode(pub, pub32)) return -1;
    drbg rng(seed32);
    bytes64 c = ct_bytes(encrypt(pub, amount, rng.scalar()));
    memcpy(ct64, c.data(), 64);
    return 0;
}
// decrypts to the point amount*G and compares (ElGamalSecretKey::decrypt, elgamal.rs:140-145): returns 1 on match
int32_t xhe_ct_decrypts_to(const uint8_t secret32[32], const uint8_t ct64[64], uint64_t amount) {
    ciphertext c; if (!ct_from_bytes(c, ct64)) return -1;
    sc s = sc_frombytes(secret32);
    ge pt = ge_sub(c.C, scalarmul(s, c.D));
    return ge_ristretto_eq(pt, mul_G(sc_from_u64(amount))) ? 1 : 0;
}
// ElGamalSecretKey::decrypt (elgamal.rs:140-145): the encoding of the point amount*G = C - s*D (what ECDLPInstance holds)
int32_t xhe_decrypt_to_point(const uint8_t secret32[32], const uint8_t ct64[64], uint8_t out32[32]) {
    ciphertext c; if (!ct_from_bytes(c, ct64)) return -1;
    ristretto_encode(out32, ge_sub(c.C, scalarmul(sc_frombytes(secret32), c.D)));
    return 0;
}
int32_t xhe_sign(const uint8_t secret32[32], const uint8_t *msg, uint64_t n, const uint8_t seed32[32], uint8_t sig64[64]) {
    keypair k = keypair::from_secret(sc_frombytes(secret32));
    drbg rng(seed32);
    bytes64 s = sign(k, msg, n, rng);
    memcpy(sig64, s.data(), 64);
    return 0;
}
int32_t xhe_signature_verify_host(const uint8_t sig64[64], const uint8_t *msg, uint64_t n, const uint8_t pub32[32]) {
    return signature_verify_host(sig64, msg, n, pub32) ? 1 : 0;
}

// TransactionBuilder::build_unsigned.  The unsigned transaction is returned as a record with no
// multisig entries and an all-zero signature.  new_balances (nullable): per emitted commitment
// 32-byte asset || 8-byte LE balance || 64-byte ciphertext of the source after this transaction.
int32_t xhe_tx_build_unsigned(const xhe_tx_request *rq, const uint8_t source_secret[32], const uint8_t seed32[32], uint8_t *out, uint64_t cap,
                              uint64_t *out_len, uint8_t *new_balances, uint32_t *n_new_balances) {
    if (!rq || !source_secret || !seed32 || !out_len) return -1;
    builder_in in; fill_builder(in, rq);
    keypair src = keypair::from_secret(sc_frombytes(source_secret));
    drbg rng(seed32);
    unsigned_tx u; std::vector<asset_balance> nb;
    int rc = build_unsigned(in, src, rng, u, &nb);
    if (rc) return 100 + rc;
    std::vector<u8> rec = u.header;
    rec.insert(rec.end(), u.preimage.begin(), u.preimage.end());
    rec.resize(rec.size() + 64, 0);
    *out_len = rec.size();
    if (rec.size() > cap || !out) return -2;
    memcpy(out, rec.data(), rec.size());
    if (n_new_balances) *n_new_balances = (uint32_t)nb.size();
    if (new_balances) for (size_t i = 0; i < nb.size(); i++) {
        u8 *p = new_balances + 104 * i;
        memcpy(p, nb[i].asset.data(), 32);
        for (int k = 0; k < 8; k++) p[32 + k] = (u8)(nb[i].balance >> (8 * k));
        memcpy(p + 40, nb[i].ct.data(), 64);
    }
    return 0;
}
// TransactionUnsigned::hash: BLAKE3 of the pre-image (without multisig entries and signature)
int32_t xhe_tx_hash(const uint8_t *record, uint64_t len, uint8_t hash32[32]) {
    tx_record t; if (!record_parse(t, record, len)) return -1;
    blake3_hash(hash32, t.pre(), t.pre_len);
    return 0;
}
// set_multisig + sign: rebuilds the record with the given multisig entries (ids[i], sigs[64*i]) and the
// source's signature over pre-image || entries.  multisig_present = 0 leaves the field None.
int32_t xhe_tx_finalize(const uint8_t *unsigned_record, uint64_t len, int32_t multisig_present, uint32_t n_multisig, const uint8_t *ids,
                        const uint8_t *sigs, const uint8_t source_secret[32], const uint8_t seed32[32], uint8_t *out, uint64_t cap, uint64_t *out_len) {
    tx_record t; if (!record_parse(t, unsigned_record, len) || t.n_multisig != 0) return -1;
    unsigned_tx u;
    u.header.assign(unsigned_record, unsigned_record + t.pre_off);
    u.preimage.assign(t.pre(), t.pre() + t.pre_len);
    u.has_multisig = multisig_present != 0;
    for (u32 i = 0; i < n_multisig; i++) u.multisig.push_back({ids[i], b64(sigs + 64 * i)});
    keypair src = keypair::from_secret(sc_frombytes(source_secret));
    drbg rng(seed32);
    std::vector<u8> rec = u.sign(src, rng);
    *out_len = rec.size();
    if (rec.size() > cap || !out) return -2;
    memcpy(out, rec.data(), rec.size());
    return 0;
}

// Transaction::apply_without_verify on the host (verify.rs:545-619): the prover's way to advance its
// own copy of the ledger.  balance_cts is updated in place.  Returns 0, or -1 where the reference
// would panic (unknown account/asset, ill-formed ciphertext).
int32_t xhe_tx_apply_host(const uint8_t *record, uint64_t len, uint64_t n_accounts, const uint8_t *account_keys, uint64_t n_balances,
                          const uint32_t *balance_account, const uint8_t *balance_assets, uint8_t *balance_cts) {
    tx_record t; if (!record_parse(t, record, len)) return -1;
    auto find_slot = [&](const u8 *account, const u8 *asset) -> int64_t {
        for (u64 i = 0; i < n_balances; i++)
            if (balance_account[i] < n_accounts && !memcmp(account_keys + 32 * balance_account[i], account, 32) && !memcmp(balance_assets + 32 * i, asset, 32)) return (int64_t)i;
        return -1;
    };
    struct tp { ge C, Ds, Dr; };
    std::vector<tp> pts(t.transfer_off.size());
    for (size_t i = 0; i < pts.size(); i++) {
        const u8 *p = t.transfer(i);
        if (!ristretto_decode(pts[i].C, p + 64) || !ristretto_decode(pts[i].Ds, p + 96) || !ristretto_decode(pts[i].Dr, p + 128)) return -1;
    }
    for (u32 i = 0; i < t.n_commitments; i++) {
        const u8 *asset = t.commitment(i);
        int64_t slot = find_slot(t.source(), asset);
        if (slot < 0) return -1;
        ciphertext cur; if (!ct_from_bytes(cur, balance_cts + 64 * slot)) return -1;
        bool of; u64 plain = record_plain_output(t, asset, &of);
        if (of) return -1;
        cur.C = ge_sub(cur.C, mul_G(sc_from_u64(plain)));
        for (size_t k = 0; k < pts.size(); k++) if (!memcmp(asset, t.transfer(k), 32)) { cur.C = ge_sub(cur.C, pts[k].C); cur.D = ge_sub(cur.D, pts[k].Ds); }
        bytes64 nb = ct_bytes(cur); memcpy(balance_cts + 64 * slot, nb.data(), 64);
    }
    for (size_t k = 0; k < pts.size(); k++) {
        const u8 *p = t.transfer(k);
        int64_t slot = find_slot(p + 32, p);
        if (slot < 0) return -1;
        ciphertext cur; if (!ct_from_bytes(cur, balance_cts + 64 * slot)) return -1;
        cur.C = ge_add(cur.C, pts[k].C); cur.D = ge_add(cur.D, pts[k].Dr);
        bytes64 nb = ct_bytes(cur); memcpy(balance_cts + 64 * slot, nb.data(), 64);
    }
    return 0;
}

// ---- synthetic batches (benches/tx.rs:129-189 generalised) -------------------------------------------
struct xhe_synth {
    std::vector<u8> records; std::vector<u64> offsets;
    std::vector<u8> account_keys; std::vector<u64> nonces;
    std::vector<u32> balance_account; std::vector<u8> balance_assets, balance_cts;
    std::vector<u8> secrets;
    xhe_ledger_view view;
};

// n_txs transfers-only transactions of n_transfers transfers each (amount 1, fee 3, nonce 0, native
// asset), spread round-robin over n_senders sender accounts funded with 10^7, each with its own
// receiver; every transaction is built against the ledger left by the previous transaction of its
// sender.  Commitments and ciphertext chains are laid out sequentially per sender (cheap), the proofs
// are generated by `threads` workers.
xhe_synth *xhe_synth_transfers_gpu(xhe_ctx *ctx, uint64_t n_txs, uint32_t n_transfers, uint32_t n_senders, const uint8_t seed32[32], int32_t threads) {
    if (!n_senders || !n_transfers || n_transfers > 255 || !seed32) return nullptr;
    if (threads < 1) threads = 1;
    u32 m = 1; while (m < n_transfers + 1) m <<= 1;
    const bool on_gpu = ctx != nullptr && m <= XHE_RANGE_PROOF_GPU_MAX_VALUES && n_txs > 0;
    xhe_synth *S = new xhe_synth();
    drbg root(seed32);
    const u64 FUNDS = 10000000;
    std::vector<keypair> senders(n_senders), receivers(n_senders);
    std::vector<bytes64> sender_ct(n_senders);
    std::vector<u64> sender_bal(n_senders, FUNDS);
    S->account_keys.resize(64ull * n_senders); S->nonces.assign(2ull * n_senders, 0);
    S->balance_account.resize(2ull * n_senders); S->balance_assets.assign(64ull * n_senders, 0); S->balance_cts.resize(128ull * n_senders);
    S->secrets.resize(64ull * n_senders);
    for (u32 i = 0; i < n_senders; i++) {
        senders[i] = keypair::from_secret(root.scalar()); receivers[i] = keypair::from_secret(root.scalar());
        sc_tobytes(&S->secrets[64 * i], senders[i].secret); sc_tobytes(&S->secrets[64 * i + 32], receivers[i].secret);
        memcpy(&S->account_keys[64 * i], senders[i].pub_bytes.data(), 32);
        memcpy(&S->account_keys[64 * i + 32], receivers[i].pub_bytes.data(), 32);
        sender_ct[i] = ct_bytes(encrypt(senders[i].pub, FUNDS, root.scalar()));
        bytes64 rc = ct_bytes(encrypt(receivers[i].pub, 0, root.scalar()));
        S->balance_account[2 * i] = 2 * i; S->balance_account[2 * i + 1] = 2 * i + 1;
        memcpy(&S->balance_cts[128 * i], sender_ct[i].data(), 64); memcpy(&S->balance_cts[128 * i + 64], rc.data(), 64);
    }
    // Per-transaction seeds and per-sender chains: a sender's transactions are built in order by one worker (each needs
    // the ciphertext its predecessor left), different senders in parallel.  With a GPU the aggregated range proofs of the
    // whole batch are left out of that pass, proved together on the device, and patched in before signing; the records
    // are the same bytes either way.
    std::vector<std::vector<u8>> recs(n_txs);
    std::vector<bytes32> seeds(n_txs);
    for (u64 k = 0; k < n_txs; k++) { u8 s[32]; root.bytes(s, 32); seeds[k] = b32(s); }
    std::vector<unsigned_tx> unsig(on_gpu ? n_txs : 0);
    std::vector<u64> sign_ctr(on_gpu ? n_txs : 0);
    std::vector<u8> g_transcripts, g_openings, g_seeds, g_proofs; std::vector<u64> g_values, g_ctr;
    if (on_gpu) {
        g_transcripts.assign((size_t)XHE_RANGE_PROOF_TRANSCRIPT_BYTES * n_txs, 0); g_openings.resize(32ull * m * n_txs); g_seeds.resize(32 * n_txs);
        g_values.resize((u64)m * n_txs); g_ctr.resize(n_txs);
    }
    std::atomic<u32> next_sender(0); std::atomic<int> failed(0);
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {           // XHE_SYNTH_TIMES=1: phase times on stderr
        if (!setting("XHE_SYNTH_TIMES")) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[xhe synth] %s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    auto worker = [&]() {
        for (;;) {
            u32 s = next_sender.fetch_add(1);
            if (s >= n_senders) break;
            for (u64 k = s; k < n_txs; k += n_senders) {
                builder_in in; in.version = 1; in.type = 0; in.fee = 3; in.nonce = 0;
                for (u32 j = 0; j < n_transfers; j++) { transfer_in tr; tr.asset = bytes32{}; tr.amount = 1; tr.dest = receivers[s].pub_bytes; in.transfers.push_back(tr); }
                asset_balance b; b.asset = bytes32{}; b.balance = sender_bal[s]; b.ct = sender_ct[s]; in.balances.push_back(b);
                drbg rng(seeds[k].data());
                std::vector<asset_balance> nb;
                if (on_gpu) {
                    pending_range_proof pend;
                    if (build_unsigned(in, senders[s], rng, unsig[k], &nb, &pend) != BUILD_OK || pend.values.size() != m) { failed = 1; return; }
                    u8 *ts = &g_transcripts[(size_t)XHE_RANGE_PROOF_TRANSCRIPT_BYTES * k];
                    memcpy(ts, pend.t.s.st.bytes, 200); ts[200] = pend.t.s.pos; ts[201] = pend.t.s.pos_begin;
                    for (u32 j = 0; j < m; j++) { g_values[(u64)m * k + j] = pend.values[j]; sc_tobytes(&g_openings[32ull * (m * k + j)], pend.openings[j]); }
                    memcpy(&g_seeds[32 * k], pend.seed, 32); g_ctr[k] = pend.ctr;
                    sign_ctr[k] = rng.ctr;
                } else {
                    unsigned_tx u;
                    if (build_unsigned(in, senders[s], rng, u, &nb) != BUILD_OK) { failed = 1; return; }
                    recs[k] = u.sign(senders[s], rng);
                }
                sender_bal[s] = nb[0].balance; sender_ct[s] = nb[0].ct;
            }
        }
    };
    gens().ensure(64 * 2);   // touch the lazily built tables before the workers start
    gens().ensure(64ull * m);
    {
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; i++) pool.emplace_back(worker);
        for (auto &th : pool) th.join();
    }
    lap(on_gpu ? "transactions without range proofs" : "transactions");
    if (!failed && on_gpu) {
        u32 lg = 6; while ((1u << lg) < 64 * m) lg++;
        const size_t proof_bytes = 32 * (9 + 2 * lg);
        g_proofs.resize(proof_bytes * n_txs);
        if (xhe_range_proofs_prove(ctx, n_txs, m, g_transcripts.data(), g_values.data(), g_openings.data(), g_seeds.data(), g_ctr.data(), g_proofs.data()) != 0) failed = 1;
        lap("range proofs on the GPU");
        std::atomic<u64> next_tx(0);
        auto signer = [&]() {
            for (;;) {
                u64 k0 = next_tx.fetch_add(64);
                if (k0 >= n_txs) break;
                for (u64 k = k0; k < std::min<u64>(k0 + 64, n_txs); k++) {
                    unsigned_tx &u = unsig[k];
                    if (u.rp_len != proof_bytes) { failed = 1; return; }
                    memcpy(&u.preimage[u.rp_off], &g_proofs[proof_bytes * k], proof_bytes);
                    drbg rng(seeds[k].data()); rng.ctr = sign_ctr[k];
                    recs[k] = u.sign(senders[k % n_senders], rng);
                    u = unsigned_tx();
                }
            }
        };
        if (!failed) {
            std::vector<std::thread> pool;
            for (int i = 0; i < threads; i++) pool.emplace_back(signer);
            for (auto &th : pool) th.join();
        }
        lap("signatures");
    }
    if (failed) { delete S; return nullptr; }
    S->offsets.push_back(0);
    {
        size_t total = 0;
        for (auto &r : recs) total += r.size();
        S->records.reserve(total);
    }
    for (u64 k = 0; k < n_txs; k++) { S->records.insert(S->records.end(), recs[k].begin(), recs[k].end()); S->offsets.push_back(S->records.size()); std::vector<u8>().swap(recs[k]); }
    xhe_ledger_view &v = S->view;
    memset(&v, 0, sizeof v);
    v.n_accounts = 2ull * n_senders; v.account_keys = S->account_keys.data(); v.account_nonces = S->nonces.data();
    v.n_balances = 2ull * n_senders; v.balance_account = S->balance_account.data(); v.balance_assets = S->balance_assets.data(); v.balance_cts = S->balance_cts.data();
    return S;
}
xhe_synth *xhe_synth_transfers(uint64_t n_txs, uint32_t n_transfers, uint32_t n_senders, const uint8_t seed32[32], int32_t threads) {
    return xhe_synth_transfers_gpu(nullptr, n_txs, n_transfers, n_senders, seed32, threads);
}
// n_groups groups of 16 transactions over disjoint account sets: two multisig set-ups (2 signers, threshold 2), four
// one-transfer transactions, four with a native and a second-asset transfer (two source commitments, four value
// commitments), two native and two second-asset burns, and a co-signed transfer from each of the two multisig
// accounts configured earlier in the same group.  Accounts per group: sender (both assets), receiver (both assets),
// two multisig accounts, two co-signers.  Every transaction is built on the ledger left by its predecessors.
xhe_synth *xhe_synth_mixed_gpu(xhe_ctx *ctx, uint64_t n_groups, const uint8_t seed32[32], int32_t threads) {
    if (!n_groups || !seed32) return nullptr;
    if (threads < 1) threads = 1;
    const bool on_gpu = ctx != nullptr;
    xhe_synth *S = new xhe_synth();
    drbg root(seed32);
    const u64 FUNDS = 1000000, ACC = 6, SLOTS = 8, TXS = 16;
    bytes32 asset2; asset2.fill(55);
    struct group_keys { keypair k[6]; };
    std::vector<group_keys> keys(n_groups);
    S->account_keys.resize(32 * ACC * n_groups); S->nonces.assign(ACC * n_groups, 0);
    S->balance_account.resize(SLOTS * n_groups); S->balance_assets.assign(32 * SLOTS * n_groups, 0); S->balance_cts.resize(64 * SLOTS * n_groups);
    S->secrets.resize(32 * ACC * n_groups);
    // slot layout per group: 0 sender/native, 1 sender/asset2, 2 receiver/native, 3 receiver/asset2, 4 multisig0, 5 multisig1, 6 cosigner0, 7 cosigner1
    const u32 slot_acc[8] = {0, 0, 1, 1, 2, 3, 4, 5};
    const bool slot_second[8] = {false, true, false, true, false, false, false, false};
    const u64 slot_funds[8] = {FUNDS, FUNDS, 0, 0, FUNDS, FUNDS, 0, 0};
    for (u64 g = 0; g < n_groups; g++) {
        for (u32 a = 0; a < ACC; a++) {
            keys[g].k[a] = keypair::from_secret(root.scalar());
            sc_tobytes(&S->secrets[32 * (ACC * g + a)], keys[g].k[a].secret);
            memcpy(&S->account_keys[32 * (ACC * g + a)], keys[g].k[a].pub_bytes.data(), 32);
        }
        for (u32 sl = 0; sl < SLOTS; sl++) {
            u64 i = SLOTS * g + sl;
            S->balance_account[i] = (u32)(ACC * g + slot_acc[sl]);
            if (slot_second[sl]) memcpy(&S->balance_assets[32 * i], asset2.data(), 32);
            bytes64 ct = ct_bytes(encrypt(keys[g].k[slot_acc[sl]].pub, slot_funds[sl], root.scalar()));
            memcpy(&S->balance_cts[64 * i], ct.data(), 64);
        }
    }
    std::vector<std::vector<u8>> recs(TXS * n_groups);
    std::vector<bytes32> seeds(n_groups);
    for (u64 g = 0; g < n_groups; g++) { u8 t[32]; root.bytes(t, 32); seeds[g] = b32(t); }
    // With a GPU the range proofs are left out of the builder's pass (the builder's random generator skips their draws),
    // proved together per shape on the device -- the sixteen transactions of a group carry 1, 2 or 4 value commitments --
    // and patched in before the co-signatures and signatures, which cover them; the generator's counter at each
    // transaction's first signing draw is kept, so the records are the same bytes as without a GPU.
    // position in the group -> value commitments (a power of two), and its rank among the group's proofs of that shape
    static const u32 shape_m[16] = {1, 1, 2, 2, 2, 2, 4, 4, 4, 4, 1, 1, 2, 2, 2, 2};
    static const u32 shape_rank[16] = {0, 1, 0, 1, 2, 3, 0, 1, 2, 3, 2, 3, 4, 5, 6, 7};
    static const u32 shape_per_group[3] = {4, 8, 4};            // m = 1, 2, 4
    struct shape_inputs { std::vector<u8> transcripts, openings, seeds, proofs; std::vector<u64> values, ctr; };
    shape_inputs sh[3];
    std::vector<unsigned_tx> unsig(on_gpu ? TXS * n_groups : 0);
    std::vector<u64> sign_ctr(on_gpu ? TXS * n_groups : 0);
    if (on_gpu) for (u32 q = 0; q < 3; q++) {
        const u64 cnt = shape_per_group[q] * n_groups, m = 1ull << q;
        sh[q].transcripts.assign((size_t)XHE_RANGE_PROOF_TRANSCRIPT_BYTES * cnt, 0); sh[q].openings.resize(32 * m * cnt); sh[q].seeds.resize(32 * cnt);
        sh[q].values.resize(m * cnt); sh[q].ctr.resize(cnt);
    }
    std::atomic<u64> next_group(0); std::atomic<int> failed(0);
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {           // XHE_SYNTH_TIMES=1: phase times on stderr
        if (!setting("XHE_SYNTH_TIMES")) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[xhe synth mixed] %s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    auto worker = [&]() {
        for (;;) {
            u64 g = next_group.fetch_add(1);
            if (g >= n_groups) break;
            drbg rng(seeds[g].data());
            const keypair *k = keys[g].k;
            // the prover's view of the balances it spends from: sender native / second asset, multisig 0 / 1
            asset_balance sn{bytes32{}, FUNDS, {}}, s2{asset2, FUNDS, {}}, m0{bytes32{}, FUNDS, {}}, m1{bytes32{}, FUNDS, {}};
            memcpy(sn.ct.data(), &S->balance_cts[64 * (SLOTS * g + 0)], 64); memcpy(s2.ct.data(), &S->balance_cts[64 * (SLOTS * g + 1)], 64);
            memcpy(m0.ct.data(), &S->balance_cts[64 * (SLOTS * g + 4)], 64); memcpy(m1.ct.data(), &S->balance_cts[64 * (SLOTS * g + 5)], 64);
            u64 base = TXS * g;
            auto build = [&](u32 pos, builder_in &in, const keypair &src, std::vector<asset_balance *> bal, bool cosign) {
                for (auto *b : bal) in.balances.push_back(*b);
                std::vector<asset_balance> nb;
                if (on_gpu) {
                    pending_range_proof pend;
                    unsigned_tx &u = unsig[base + pos];
                    const u32 m = shape_m[pos], q = m == 1 ? 0 : m == 2 ? 1 : 2;
                    if (build_unsigned(in, src, rng, u, &nb, &pend) != BUILD_OK || pend.values.size() != m) { failed = 1; return; }
                    const u64 r = (u64)shape_per_group[q] * g + shape_rank[pos];
                    u8 *ts = &sh[q].transcripts[(size_t)XHE_RANGE_PROOF_TRANSCRIPT_BYTES * r];
                    memcpy(ts, pend.t.s.st.bytes, 200); ts[200] = pend.t.s.pos; ts[201] = pend.t.s.pos_begin;
                    for (u32 j = 0; j < m; j++) { sh[q].values[(u64)m * r + j] = pend.values[j]; sc_tobytes(&sh[q].openings[32ull * (m * r + j)], pend.openings[j]); }
                    memcpy(&sh[q].seeds[32 * r], pend.seed, 32); sh[q].ctr[r] = pend.ctr;
                    sign_ctr[base + pos] = rng.ctr;
                    rng.ctr += cosign ? 3 : 1;                    // the co-signers' nonces and the source's, drawn when the proof is in
                } else {
                    unsigned_tx u;
                    if (build_unsigned(in, src, rng, u, &nb) != BUILD_OK) { failed = 1; return; }
                    if (cosign) {
                        bytes32 h = u.hash();
                        u.has_multisig = true;
                        for (u8 c = 0; c < 2; c++) u.multisig.push_back({c, host::sign(k[4 + c], h.data(), 32, rng)});
                    }
                    recs[base + pos] = u.sign(src, rng);
                }
                for (auto &n : nb) for (auto *b : bal) if (b->asset == n.asset) *b = n;
            };
            auto transfer = [&](const bytes32 &asset) { transfer_in tr; tr.asset = asset; tr.amount = 1; tr.dest = k[1].pub_bytes; return tr; };
            for (u32 j = 0; j < 2; j++) {             // multisig set-ups
                builder_in in; in.type = 2; in.fee = 3; in.signers = {k[4].pub_bytes, k[5].pub_bytes}; in.threshold = 2;
                build(j, in, k[2 + j], {j ? &m1 : &m0}, false);
            }
            for (u32 j = 0; j < 4; j++) {             // one transfer, one asset
                builder_in in; in.type = 0; in.fee = 3; in.transfers = {transfer(bytes32{})};
                build(2 + j, in, k[0], {&sn}, false);
            }
            for (u32 j = 0; j < 4; j++) {             // two transfers, two assets
                builder_in in; in.type = 0; in.fee = 3; in.transfers = {transfer(bytes32{}), transfer(asset2)};
                build(6 + j, in, k[0], {&sn, &s2}, false);
            }
            for (u32 j = 0; j < 4; j++) {             // burns: native, native, second asset, second asset
                builder_in in; in.type = 1; in.fee = 3; in.burn_asset = j < 2 ? bytes32{} : asset2; in.burn_amount = 1;
                if (j < 2) build(10 + j, in, k[0], {&sn}, false); else build(10 + j, in, k[0], {&sn, &s2}, false);
            }
            for (u32 j = 0; j < 2; j++) {             // co-signed transfers from the multisig accounts
                builder_in in; in.type = 0; in.fee = 3; in.transfers = {transfer(bytes32{})};
                build(14 + j, in, k[2 + j], {j ? &m1 : &m0}, true);
            }
            if (failed) return;
        }
    };
    gens().ensure(64 * 4);
    {
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; i++) pool.emplace_back(worker);
        for (auto &th : pool) th.join();
    }
    lap(on_gpu ? "transactions without range proofs" : "transactions");
    if (!failed && on_gpu) {
        for (u32 q = 0; q < 3 && !failed; q++) {
            const u64 cnt = shape_per_group[q] * n_groups;
            const u32 m = 1u << q, lg = 6 + q;
            sh[q].proofs.resize((size_t)32 * (9 + 2 * lg) * cnt);
            if (xhe_range_proofs_prove(ctx, cnt, m, sh[q].transcripts.data(), sh[q].values.data(), sh[q].openings.data(), sh[q].seeds.data(), sh[q].ctr.data(),
                                       sh[q].proofs.data()) != 0) failed = 1;
        }
        lap("range proofs on the GPU");
        std::atomic<u64> next_sign(0);
        auto signer = [&]() {
            for (;;) {
                u64 g = next_sign.fetch_add(1);
                if (g >= n_groups) break;
                const keypair *k = keys[g].k;
                for (u32 pos = 0; pos < TXS; pos++) {
                    unsigned_tx &u = unsig[TXS * g + pos];
                    const u32 m = shape_m[pos], q = m == 1 ? 0 : m == 2 ? 1 : 2;
                    const size_t proof_bytes = 32 * (9 + 2 * (6 + q));
                    if (u.rp_len != proof_bytes) { failed = 1; return; }
                    memcpy(&u.preimage[u.rp_off], &sh[q].proofs[proof_bytes * ((u64)shape_per_group[q] * g + shape_rank[pos])], proof_bytes);
                    drbg rng(seeds[g].data()); rng.ctr = sign_ctr[TXS * g + pos];
                    const bool cosign = pos >= 14;
                    if (cosign) {
                        bytes32 h = u.hash();
                        u.has_multisig = true;
                        for (u8 c = 0; c < 2; c++) u.multisig.push_back({c, host::sign(k[4 + c], h.data(), 32, rng)});
                    }
                    const keypair &src = pos < 2 ? k[2 + pos] : pos >= 14 ? k[2 + (pos - 14)] : k[0];
                    recs[TXS * g + pos] = u.sign(src, rng);
                    u = unsigned_tx();
                }
            }
        };
        if (!failed) {
            std::vector<std::thread> pool;
            for (int i = 0; i < threads; i++) pool.emplace_back(signer);
            for (auto &th : pool) th.join();
        }
        lap("signatures");
    }
    if (failed) { delete S; return nullptr; }
    S->offsets.push_back(0);
    {
        size_t total = 0;
        for (auto &r : recs) total += r.size();
        S->records.reserve(total);
    }
    for (auto &r : recs) { S->records.insert(S->records.end(), r.begin(), r.end()); S->offsets.push_back(S->records.size()); std::vector<u8>().swap(r); }
    xhe_ledger_view &v = S->view;
    memset(&v, 0, sizeof v);
    v.n_accounts = ACC * n_groups; v.account_keys = S->account_keys.data(); v.account_nonces = S->nonces.data();
    v.n_balances = SLOTS * n_groups; v.balance_account = S->balance_account.data(); v.balance_assets = S->balance_assets.data(); v.balance_cts = S->balance_cts.data();
    return S;
}
xhe_synth *xhe_synth_mixed(uint64_t n_groups, const uint8_t seed32[32], int32_t threads) { return xhe_synth_mixed_gpu(nullptr, n_groups, seed32, threads); }
void xhe_synth_free(xhe_synth *s) { delete s; }
const uint8_t *xhe_synth_records(const xhe_synth *s, uint64_t *len) { if (len) *len = s->records.size(); return s->records.data(); }
const uint64_t *xhe_synth_offsets(const xhe_synth *s, uint64_t *n_txs) { if (n_txs) *n_txs = s->offsets.size() - 1; return s->offsets.data(); }
const xhe_ledger_view *xhe_synth_ledger(const xhe_synth *s) { return &s->view; }
const uint8_t *xhe_synth_secrets(const xhe_synth *s) { return s->secrets.data(); }

}  // extern "C"
