// Multi-GPU Transaction::verify_batch behind the C ABI (include/xhe.h: xhe_partition_batch, xhe_verify_batch_sharded).  Host code only.
//
// A transaction's checks read the sender's balance as every earlier transaction touching that (account, asset) left it
// (/root/reference/src/tx/verify.rs:301-341), so the unit of sharding is a connected component of the graph of source and
// destination accounts, not a transaction.  Components go, heaviest first, onto the lightest shard (weight = multiscalar
// terms); inside a shard the batch order is kept.  Every shard is verified against the same ledger snapshot by its own
// context; the verdicts combine by the reference's precedence -- the per-transaction error of the lowest transaction index
// (verify.rs:492-498), then GenericProof, then RangeProof (verify.rs:500-514) -- which is a MIN over one 64-bit key per shard,
// the same key the process-per-GPU driver all-reduces over RCCL.  The ledger merges by ownership: a slot belongs to the shard
// that holds its account's component.
#include <string.h>
#include <algorithm>
#include <atomic>
#include <numeric>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../include/xhe.h"
#include "xhe_record.h"

using namespace xhe;

namespace {
struct key32 { u8 b[32]; bool operator==(const key32 &o) const { return !memcmp(b, o.b, 32); } };
struct key32_hash { size_t operator()(const key32 &k) const { u64 h, t; memcpy(&h, k.b, 8); memcpy(&t, k.b + 24, 8); return (size_t)(h ^ (t * 0x9e3779b97f4a7c15ull)); } };
static key32 make_key(const u8 *p) { key32 k; memcpy(k.b, p, 32); return k; }

// dynamic multiscalar terms of a transaction: 7 per source commitment, 8 per transfer, 4 + 2 lg + m of the range proof
static u64 tx_weight(const tx_record &t) { return 7ull * t.n_commitments + 8ull * t.transfer_off.size() + 4 + 2ull * t.lg() + t.m(); }

struct partition_result {
    std::vector<u32> shard_of_tx;
    std::vector<u32> account_ids;                      // per transaction: the id of its source account (component look-ups)
    std::unordered_map<key32, u32, key32_hash> id_of;  // account key -> id
    std::vector<u32> shard_of_account;                 // account id -> shard
    int64_t first_malformed = -1;
};

static bool partition_batch(partition_result &pr, const u8 *records, const u64 *offsets, u64 n, u32 n_shards) {
    std::vector<tx_record> recs(n);
    {
        // parse on several threads; the smallest malformed index wins, as in the single-context path
        std::atomic<u64> first_bad(~0ull);
        unsigned threads = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
        if (n < 4096) threads = 1;
        auto work = [&](u64 lo, u64 hi) {
            for (u64 i = lo; i < hi; i++)
                if (offsets[i + 1] < offsets[i] || !record_parse(recs[i], records + offsets[i], offsets[i + 1] - offsets[i])) {
                    u64 seen = first_bad.load();
                    while (i < seen && !first_bad.compare_exchange_weak(seen, i)) {}
                    break;
                }
        };
        std::vector<std::thread> pool;
        const u64 per = (n + threads - 1) / threads;
        for (unsigned t = 1; t < threads; t++) { u64 lo = t * per, hi = std::min(n, lo + per); if (lo < hi) pool.emplace_back(work, lo, hi); }
        if (n) work(0, std::min(n, per));
        for (auto &th : pool) th.join();
        if (first_bad.load() != ~0ull) { pr.first_malformed = (int64_t)first_bad.load(); return false; }
    }
    // union-find over the account keys the batch names
    std::vector<u32> parent;
    auto id = [&](const u8 *key) {
        auto it = pr.id_of.find(make_key(key));
        if (it != pr.id_of.end()) return it->second;
        u32 v = (u32)parent.size();
        parent.push_back(v); pr.id_of.emplace(make_key(key), v);
        return v;
    };
    auto find = [&](u32 x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    pr.account_ids.resize(n);
    pr.id_of.reserve(2 * n + 16);
    for (u64 i = 0; i < n; i++) {
        const tx_record &t = recs[i];
        u32 a = find(id(t.source()));
        pr.account_ids[i] = a;
        for (size_t k = 0; k < t.transfer_off.size(); k++) {
            u32 b = find(id(t.transfer(k) + 32));
            if (a != b) { parent[b] = a; }
        }
    }
    // components: weight and first transaction
    const u32 n_ids = (u32)parent.size();
    std::vector<u64> weight(n_ids, 0); std::vector<u64> first_tx(n_ids, ~0ull);
    for (u64 i = 0; i < n; i++) { u32 c = find(pr.account_ids[i]); weight[c] += tx_weight(recs[i]); if (first_tx[c] == ~0ull) first_tx[c] = i; }
    std::vector<u32> comps;
    for (u32 c = 0; c < n_ids; c++) if (first_tx[c] != ~0ull) comps.push_back(c);
    std::sort(comps.begin(), comps.end(), [&](u32 x, u32 y) { return weight[x] != weight[y] ? weight[x] > weight[y] : first_tx[x] < first_tx[y]; });
    std::vector<u64> load(n_shards, 0);
    std::vector<u32> shard_of_comp(n_ids, 0);
    for (u32 c : comps) {
        u32 best = 0;
        for (u32 s = 1; s < n_shards; s++) if (load[s] < load[best]) best = s;
        shard_of_comp[c] = best; load[best] += weight[c];
    }
    pr.shard_of_tx.resize(n);
    for (u64 i = 0; i < n; i++) pr.shard_of_tx[i] = shard_of_comp[find(pr.account_ids[i])];
    pr.shard_of_account.resize(n_ids);
    for (u32 a = 0; a < n_ids; a++) pr.shard_of_account[a] = first_tx[find(a)] != ~0ull ? shard_of_comp[find(a)] : 0xffffffffu;
    return true;
}

// the key of the MIN reduction: accept = all ones; per-transaction errors order by global index, then the aggregate checks
static const u64 ACCEPT_KEY = ~0ull >> 1;
static u64 verdict_key(const xhe_verdict &v, const std::vector<u64> &global_index) {
    if (v.accept) return ACCEPT_KEY;
    const u64 stage = v.err_code == XHE_ERR_GENERIC_PROOF ? 1 : v.err_code == XHE_ERR_RANGE_PROOF ? 2 : 0;
    u64 index = (1ull << 52) - 1;                       // "no index" sorts behind every index of its stage
    if (v.first_bad_tx >= 0 && (u64)v.first_bad_tx < global_index.size()) index = global_index[(size_t)v.first_bad_tx];
    return (stage << 60) | (index << 8) | ((u64)v.err_code & 0xff);
}
}  // namespace

extern "C" {

int32_t xhe_partition_batch(const uint8_t *records, const uint64_t *offsets, uint64_t n_txs, uint32_t n_shards, uint32_t *shard_of_tx, int64_t *first_malformed) {
    if (!offsets || !shard_of_tx || !n_shards || (n_txs && !records)) return XHE_E_ARG;
    partition_result pr;
    bool ok = partition_batch(pr, records, offsets, n_txs, n_shards);
    if (first_malformed) *first_malformed = pr.first_malformed;
    if (!ok) return 1;
    memcpy(shard_of_tx, pr.shard_of_tx.data(), 4 * n_txs);
    return 0;
}

uint64_t xhe_verdict_key(const xhe_verdict *v, const uint64_t *global_index, uint64_t n_local) {
    std::vector<u64> g(global_index, global_index + (global_index ? n_local : 0));
    return verdict_key(*v, g);
}
void xhe_verdict_from_key(uint64_t key, xhe_verdict *v) {
    if (key == ACCEPT_KEY) { *v = {1, XHE_OK, -1}; return; }
    const u64 index = (key >> 8) & ((1ull << 52) - 1);
    *v = {0, (int32_t)(key & 0xff), index == (1ull << 52) - 1 ? (int64_t)-1 : (int64_t)index};
}

int32_t xhe_verify_batch_sharded(xhe_ctx *const *ctxs, uint32_t n_ctx, const uint8_t *records, const uint64_t *offsets, uint64_t n_txs, const xhe_ledger_view *in,
                                 const xhe_ledger_out *out, xhe_verdict *verdict, const uint8_t rng_seed[32]) {
    if (!ctxs || !n_ctx || !offsets || !in || !verdict || !rng_seed || (n_txs && !records)) return XHE_E_ARG;
    for (u32 s = 0; s < n_ctx; s++) if (!ctxs[s]) return XHE_E_ARG;
    if (n_ctx == 1) return xhe_verify_batch(ctxs[0], records, offsets, n_txs, in, out, verdict, rng_seed);
    partition_result pr;
    if (!partition_batch(pr, records, offsets, n_txs, n_ctx)) { *verdict = {0, XHE_ERR_MALFORMED, pr.first_malformed}; return 0; }
    // gather every shard's records (batch order kept)
    struct shard { std::vector<u8> blob; std::vector<u64> offs, global; std::vector<u8> cts; std::vector<u64> nonces; std::vector<int64_t> ms; xhe_verdict v{0, 0, -1}; int32_t rc = 0; };
    std::vector<shard> sh(n_ctx);
    {
        std::vector<u64> bytes(n_ctx, 0);
        for (u64 i = 0; i < n_txs; i++) bytes[pr.shard_of_tx[i]] += offsets[i + 1] - offsets[i];
        for (u32 s = 0; s < n_ctx; s++) { sh[s].blob.reserve(bytes[s]); sh[s].offs.push_back(0); }
        for (u64 i = 0; i < n_txs; i++) {
            shard &S = sh[pr.shard_of_tx[i]];
            S.blob.insert(S.blob.end(), records + offsets[i], records + offsets[i + 1]);
            S.offs.push_back(S.blob.size()); S.global.push_back(i);
        }
    }
    std::vector<std::thread> pool;
    for (u32 s = 0; s < n_ctx; s++)
        pool.emplace_back([&, s]() {
            shard &S = sh[s];
            S.cts.resize(64 * in->n_balances + 64); S.nonces.resize(in->n_accounts + 1); S.ms.assign(in->n_accounts + 1, -1);
            xhe_ledger_out o{S.cts.data(), S.nonces.data(), S.ms.data()};
            // a different seed per shard: the shards' batch factors are independent
            u8 seed[32]; memcpy(seed, rng_seed, 32); seed[31] ^= (u8)s; seed[30] ^= (u8)(s >> 8);
            S.rc = xhe_verify_batch(ctxs[s], S.blob.data(), S.offs.data(), S.global.size(), in, &o, &S.v, seed);
        });
    for (auto &th : pool) th.join();
    u64 key = ACCEPT_KEY;
    for (u32 s = 0; s < n_ctx; s++) { if (sh[s].rc < 0) return sh[s].rc; key = std::min(key, verdict_key(sh[s].v, sh[s].global)); }
    xhe_verdict_from_key(key, verdict);
    if (!verdict->accept || !out) return 0;
    // ledger: a slot is written by the shard that holds its account's component; untouched accounts keep their input
    if (out->balance_cts) {
        if (out->balance_cts != in->balance_cts) memmove(out->balance_cts, in->balance_cts, 64 * in->n_balances);
        for (u64 i = 0; i < in->n_balances; i++) {
            auto it = pr.id_of.find(make_key(in->account_keys + 32ull * in->balance_account[i]));
            if (it == pr.id_of.end()) continue;
            const u32 s = pr.shard_of_account[it->second];
            if (s < n_ctx) memcpy(out->balance_cts + 64 * i, sh[s].cts.data() + 64 * i, 64);
        }
    }
    if (out->account_nonces) memcpy(out->account_nonces, in->account_nonces, 8 * in->n_accounts);
    if (out->multisig_set_by)
        for (u64 a = 0; a < in->n_accounts; a++) {
            int64_t best = -1;
            for (u32 s = 0; s < n_ctx; s++) { int64_t l = sh[s].ms[a]; if (l >= 0 && (int64_t)sh[s].global[(size_t)l] > best) best = (int64_t)sh[s].global[(size_t)l]; }
            out->multisig_set_by[a] = best;
        }
    return 0;
}

}  // extern "C"
