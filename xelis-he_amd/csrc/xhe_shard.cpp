// Multi-GPU Transaction::verify_batch behind the C ABI (include/xhe.h: xhe_partition_batch, xhe_plan_shards, xhe_slice_shard,
// xhe_verify_batch_sharded).  Host code only.
//
// A transaction's checks read the sender's balance as every earlier transaction touching that (account, asset) left it
// (/root/reference/src/tx/verify.rs:301-341), so the unit of sharding is a connected component of the graph of source and
// destination accounts, not a transaction.  Components go, heaviest first, onto the lightest shard (weight = multiscalar
// terms); inside a shard the batch order is kept.  Every shard is verified by its own context against the part of the ledger
// snapshot its transactions name; the verdicts combine by the reference's precedence -- the per-transaction error of the lowest
// transaction index (verify.rs:492-498), then GenericProof, then RangeProof (verify.rs:500-514) -- which is a MIN over one
// 64-bit key per shard, the same key the process-per-GPU driver all-reduces over RCCL.
//
// When the components do not balance (one account group holds most of the batch: the reference's own bench has ONE sender for
// every transaction, /root/reference/benches/tx.rs:129-189), the batch is cut into consecutive ranges instead.  Range s must
// start from the ledger the ranges in front of it leave; that state is the Twisted-ElGamal arithmetic alone
// (Transaction::apply_without_verify, verify.rs:545-619: xhe_batch_apply), so the ranges are uploaded side by side, the ledger
// runs through them once, range after range, and every range is verified as soon as its starting state is known.
#include <string.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <numeric>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../include/xhe.h"
#include "xhe_record.h"

using namespace xhe;

namespace {
constexpr u32 NONE = 0xffffffffu;
struct key32 { u8 b[32]; bool operator==(const key32 &o) const { return !memcmp(b, o.b, 32); } };
struct key32_hash { size_t operator()(const key32 &k) const { u64 h, t; memcpy(&h, k.b, 8); memcpy(&t, k.b + 24, 8); return (size_t)(h ^ (t * 0x9e3779b97f4a7c15ull)); } };
static key32 make_key(const u8 *p) { key32 k; memcpy(k.b, p, 32); return k; }

// dynamic multiscalar terms of a transaction: 7 per source commitment, 8 per transfer, 4 + 2 lg + m of the range proof
static u64 tx_weight(const tx_record &t) { return 7ull * t.n_commitments + 8ull * t.transfer_off.size() + 4 + 2ull * t.lg() + t.m(); }

// A shard's load may exceed the even share by this much before the batch is cut into consecutive ranges instead (per cent)
constexpr u64 IMBALANCE_PCT = 125;

struct shard_plan {
    std::vector<u32> shard_of_tx;
    bool ranges = false;                               // consecutive ranges (shard_of_tx ascends) instead of account groups
    std::vector<u32> account_ids;                      // per transaction: the id of its source account
    std::unordered_map<key32, u32, key32_hash> id_of;  // account key -> id (every source and destination key of the batch)
    std::vector<std::vector<u32>> ids_of_shard;        // the account ids each shard's transactions name
    int64_t first_malformed = -1;
};

static bool plan_shards(shard_plan &pr, const u8 *records, const u64 *offsets, u64 n, u32 n_shards, bool allow_ranges) {
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
    std::vector<u32> dest_first(n + 1, 0), dest_ids;   // every transaction's destination ids (for the shards' account lists)
    bool any_multisig_setup = false;
    for (u64 i = 0; i < n; i++) {
        const tx_record &t = recs[i];
        const u32 own = id(t.source());
        u32 a = find(own);
        pr.account_ids[i] = own;
        dest_first[i] = (u32)dest_ids.size();
        for (size_t k = 0; k < t.transfer_off.size(); k++) {
            const u32 d = id(t.transfer(k) + 32);
            dest_ids.push_back(d);
            u32 b = find(d);
            if (a != b) { parent[b] = a; }
        }
        any_multisig_setup |= t.type == 2;
    }
    dest_first[n] = (u32)dest_ids.size();
    // components: weight and first transaction
    const u32 n_ids = (u32)parent.size();
    std::vector<u64> weight(n_ids, 0); std::vector<u64> first_tx(n_ids, ~0ull);
    u64 total = 0;
    for (u64 i = 0; i < n; i++) { u32 c = find(pr.account_ids[i]); const u64 w = tx_weight(recs[i]); weight[c] += w; total += w; if (first_tx[c] == ~0ull) first_tx[c] = i; }
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
    const u64 heaviest = *std::max_element(load.begin(), load.end());
    // A transaction that changes an account's multisig configuration decides which keys later transactions are checked
    // against; ranges would have to hand that on as well, so such batches keep to their account groups.
    pr.ranges = allow_ranges && n_shards > 1 && n >= n_shards && !any_multisig_setup && heaviest * 100 * n_shards > total * IMBALANCE_PCT;
    if (pr.ranges) {
        u64 run = 0, s = 0;
        for (u64 i = 0; i < n; i++) {
            // range s ends where the running weight passes (s + 1) / n_shards of the total; every range gets a transaction
            while (s + 1 < n_shards && run * n_shards >= total * (s + 1) && i >= s + 1) s++;
            if (n - i <= n_shards - 1 - s) s = n_shards - (n - i);           // the last ranges are not left empty
            pr.shard_of_tx[i] = (u32)s;
            run += tx_weight(recs[i]);
        }
    } else {
        for (u64 i = 0; i < n; i++) pr.shard_of_tx[i] = shard_of_comp[find(pr.account_ids[i])];
    }
    // the accounts every shard names
    pr.ids_of_shard.assign(n_shards, {});
    {
        std::vector<u32> last_shard(n_ids, NONE);        // (ranges ascend, components are disjoint: a repeated id of a shard is the last one seen)
        auto add = [&](u32 s, u32 a) { if (last_shard[a] != s) { last_shard[a] = s; pr.ids_of_shard[s].push_back(a); } };
        for (u64 i = 0; i < n; i++) {
            const u32 s = pr.shard_of_tx[i];
            add(s, pr.account_ids[i]);
            for (u32 k = dest_first[i]; k < dest_first[i + 1]; k++) add(s, dest_ids[k]);
        }
    }
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

// The part of a ledger snapshot a shard's transactions name: its accounts (with their nonces and multisig configurations) and
// every balance of those accounts, with the maps back to the whole view.
struct sub_view {
    std::vector<u32> acc_global, bal_global;
    std::vector<u8> keys, assets, cts, ms_thr, ms_signers;
    std::vector<u64> nonces;
    std::vector<u32> bal_acc, ms_acc, ms_off;
    xhe_ledger_view view;
    void finish() {
        memset(&view, 0, sizeof view);
        ms_off.insert(ms_off.begin(), 0u);
        view.n_accounts = acc_global.size(); view.account_keys = keys.data(); view.account_nonces = nonces.data();
        view.n_balances = bal_global.size(); view.balance_account = bal_acc.data(); view.balance_assets = assets.data(); view.balance_cts = cts.data();
        view.n_multisig = ms_acc.size(); view.multisig_account = ms_acc.data(); view.multisig_threshold = ms_thr.data(); view.multisig_offsets = ms_off.data();
        view.multisig_signers = ms_signers.data();
    }
};

// false: the view's own indices are inconsistent (the single-context call reports that)
static bool view_consistent(const xhe_ledger_view &in) {
    for (u64 i = 0; i < in.n_balances; i++) if (in.balance_account[i] >= in.n_accounts) return false;
    for (u64 i = 0; i < in.n_multisig; i++) if (in.multisig_account[i] >= in.n_accounts || in.multisig_offsets[i + 1] < in.multisig_offsets[i]) return false;
    return !(in.n_multisig && in.multisig_offsets[0] != 0);
}

static void build_sub_views(std::vector<sub_view> &sv, const shard_plan &pr, const xhe_ledger_view &in) {
    const u32 n_shards = (u32)pr.ids_of_shard.size();
    sv.assign(n_shards, sub_view());
    // ledger account of every id the batch names (an id without one: the verification reports the unknown account)
    std::vector<u32> ledger_acc_of_id(pr.id_of.size(), NONE);
    for (u64 a = 0; a < in.n_accounts; a++) {
        auto it = pr.id_of.find(make_key(in.account_keys + 32 * a));
        if (it != pr.id_of.end() && ledger_acc_of_id[it->second] == NONE) ledger_acc_of_id[it->second] = (u32)a;
    }
    // balances and multisig configurations by account
    std::vector<u32> bal_off(in.n_accounts + 1, 0), bal_idx(in.n_balances), ms_of_acc(in.n_accounts, NONE);
    for (u64 i = 0; i < in.n_balances; i++) bal_off[in.balance_account[i] + 1]++;
    for (u64 a = 0; a < in.n_accounts; a++) bal_off[a + 1] += bal_off[a];
    { std::vector<u32> fill(bal_off.begin(), bal_off.end() - 1); for (u64 i = 0; i < in.n_balances; i++) bal_idx[fill[in.balance_account[i]]++] = (u32)i; }
    for (u64 i = 0; i < in.n_multisig; i++) ms_of_acc[in.multisig_account[i]] = (u32)i;       // (the last entry of an account wins, as in the single-context walk)
    for (u32 s = 0; s < n_shards; s++) {
        sub_view &v = sv[s];
        for (u32 id : pr.ids_of_shard[s]) {
            const u32 a = ledger_acc_of_id[id];
            if (a == NONE) continue;
            const u32 local = (u32)v.acc_global.size();
            v.acc_global.push_back(a);
            v.keys.insert(v.keys.end(), in.account_keys + 32ull * a, in.account_keys + 32ull * a + 32);
            v.nonces.push_back(in.account_nonces[a]);
            for (u32 k = bal_off[a]; k < bal_off[a + 1]; k++) {
                const u32 b = bal_idx[k];
                v.bal_global.push_back(b); v.bal_acc.push_back(local);
                v.assets.insert(v.assets.end(), in.balance_assets + 32ull * b, in.balance_assets + 32ull * b + 32);
                v.cts.insert(v.cts.end(), in.balance_cts + 64ull * b, in.balance_cts + 64ull * b + 64);
            }
            if (ms_of_acc[a] != NONE) {
                const u32 m = ms_of_acc[a];
                v.ms_acc.push_back(local); v.ms_thr.push_back(in.multisig_threshold[m]);
                v.ms_signers.insert(v.ms_signers.end(), in.multisig_signers + 32ull * in.multisig_offsets[m], in.multisig_signers + 32ull * in.multisig_offsets[m + 1]);
                v.ms_off.push_back((u32)(v.ms_signers.size() / 32));
            }
        }
        v.finish();
    }
}

struct shard_records { std::vector<u8> blob; std::vector<u64> offs, global; };
static void gather_records(std::vector<shard_records> &sh, const shard_plan &pr, const u8 *records, const u64 *offsets, u64 n_txs, u32 n_shards) {
    sh.assign(n_shards, shard_records());
    std::vector<u64> bytes(n_shards, 0), count(n_shards, 0);
    for (u64 i = 0; i < n_txs; i++) { bytes[pr.shard_of_tx[i]] += offsets[i + 1] - offsets[i]; count[pr.shard_of_tx[i]]++; }
    for (u32 s = 0; s < n_shards; s++) { sh[s].blob.reserve(bytes[s]); sh[s].offs.reserve(count[s] + 1); sh[s].global.reserve(count[s]); sh[s].offs.push_back(0); }
    for (u64 i = 0; i < n_txs; i++) {
        shard_records &S = sh[pr.shard_of_tx[i]];
        S.blob.insert(S.blob.end(), records + offsets[i], records + offsets[i + 1]);
        S.offs.push_back(S.blob.size()); S.global.push_back(i);
    }
}
}  // namespace

extern "C" {

int32_t xhe_partition_batch(const uint8_t *records, const uint64_t *offsets, uint64_t n_txs, uint32_t n_shards, uint32_t *shard_of_tx, int64_t *first_malformed) {
    if (!offsets || !shard_of_tx || !n_shards || (n_txs && !records)) return XHE_E_ARG;
    shard_plan pr;
    bool ok = plan_shards(pr, records, offsets, n_txs, n_shards, false);
    if (first_malformed) *first_malformed = pr.first_malformed;
    if (!ok) return 1;
    memcpy(shard_of_tx, pr.shard_of_tx.data(), 4 * n_txs);
    return 0;
}

int32_t xhe_plan_shards(const uint8_t *records, const uint64_t *offsets, uint64_t n_txs, uint32_t n_shards, uint32_t *shard_of_tx, int32_t *consecutive_ranges,
                        int64_t *first_malformed) {
    if (!offsets || !shard_of_tx || !consecutive_ranges || !n_shards || (n_txs && !records)) return XHE_E_ARG;
    shard_plan pr;
    bool ok = plan_shards(pr, records, offsets, n_txs, n_shards, true);
    if (first_malformed) *first_malformed = pr.first_malformed;
    if (!ok) return 1;
    memcpy(shard_of_tx, pr.shard_of_tx.data(), 4 * n_txs);
    *consecutive_ranges = pr.ranges ? 1 : 0;
    return 0;
}

// One shard's records, offsets and global indices out of a planned batch (shard_of_tx from xhe_plan_shards / xhe_partition_batch).
// Call with out_records = NULL to learn the sizes.
int32_t xhe_slice_shard(const uint8_t *records, const uint64_t *offsets, uint64_t n_txs, const uint32_t *shard_of_tx, uint32_t shard, uint8_t *out_records,
                        uint64_t *out_offsets, uint64_t *out_global_index, uint64_t *n_local, uint64_t *n_bytes) {
    if (!offsets || !shard_of_tx || !n_local || !n_bytes || (n_txs && !records)) return XHE_E_ARG;
    u64 count = 0, bytes = 0;
    for (u64 i = 0; i < n_txs; i++) if (shard_of_tx[i] == shard) { count++; bytes += offsets[i + 1] - offsets[i]; }
    *n_local = count; *n_bytes = bytes;
    if (!out_records && !out_offsets && !out_global_index) return 0;
    if ((bytes && !out_records) || !out_offsets || !out_global_index) return XHE_E_ARG;
    u64 k = 0, at = 0;
    out_offsets[0] = 0;
    for (u64 i = 0; i < n_txs; i++)
        if (shard_of_tx[i] == shard) {
            memcpy(out_records + at, records + offsets[i], offsets[i + 1] - offsets[i]);
            at += offsets[i + 1] - offsets[i];
            out_global_index[k] = i; out_offsets[++k] = at;
        }
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
    if (n_ctx == 1 || !view_consistent(*in)) return xhe_verify_batch(ctxs[0], records, offsets, n_txs, in, out, verdict, rng_seed);
    shard_plan pr;
    if (!plan_shards(pr, records, offsets, n_txs, n_ctx, true)) { *verdict = {0, XHE_ERR_MALFORMED, pr.first_malformed}; return 0; }
    std::vector<shard_records> sh;
    gather_records(sh, pr, records, offsets, n_txs, n_ctx);
    std::vector<sub_view> sv;
    build_sub_views(sv, pr, *in);
    struct result { std::vector<u8> cts; std::vector<u64> nonces; std::vector<int64_t> ms; xhe_verdict v{0, XHE_ERR_STATE, -1}; int32_t rc = 0; bool ran = false; };
    std::vector<result> res(n_ctx);
    auto seed_of = [&](u32 s, u8 seed[32]) { memcpy(seed, rng_seed, 32); seed[31] ^= (u8)s; seed[30] ^= (u8)(s >> 8); };      // independent batch factors per shard
    std::vector<u8> final_cts;          // (ranges) the whole ledger as the ranges leave it

    if (!pr.ranges) {
        std::vector<std::thread> pool;
        for (u32 s = 0; s < n_ctx; s++)
            pool.emplace_back([&, s]() {
                result &R = res[s];
                R.cts.resize(64 * sv[s].view.n_balances + 64); R.nonces.resize(sv[s].view.n_accounts + 1); R.ms.assign(sv[s].view.n_accounts + 1, -1);
                xhe_ledger_out o{R.cts.data(), R.nonces.data(), R.ms.data()};
                u8 seed[32]; seed_of(s, seed);
                R.rc = xhe_verify_batch(ctxs[s], sh[s].blob.data(), sh[s].offs.data(), sh[s].global.size(), &sv[s].view, &o, &R.v, seed);
                R.ran = true;
            });
        for (auto &th : pool) th.join();
    } else {
        // Consecutive ranges.  Every range is uploaded by a thread of its own (against the snapshot: its tables depend on the
        // ledger's keys only); the ledger then runs through the ranges in order -- set the range's starting ciphertexts, apply,
        // read the result -- and a range is verified, on its own context, as soon as it has been applied.  A range whose apply
        // fails holds a transaction that fails verification too: the ranges behind it are not verified (nothing they hold can
        // come first).
        std::vector<xhe_batch *> up(n_ctx, nullptr);
        std::vector<xhe_verdict> structural(n_ctx, xhe_verdict{1, XHE_OK, -1});
        std::vector<int32_t> up_rc(n_ctx, 0);
        {
            std::vector<std::thread> pool;
            for (u32 s = 0; s < n_ctx; s++)
                pool.emplace_back([&, s]() { up_rc[s] = xhe_batch_upload(ctxs[s], sh[s].blob.data(), sh[s].offs.data(), sh[s].global.size(), &sv[s].view, &up[s], &structural[s]); });
            for (auto &th : pool) th.join();
        }
        std::vector<u8> cur(in->balance_cts, in->balance_cts + 64 * in->n_balances);
        std::mutex mu; std::condition_variable cv;
        std::vector<char> may_verify(n_ctx, 0);
        bool chain_broken = false;
        std::vector<std::thread> pool;
        for (u32 s = 0; s < n_ctx; s++)
            pool.emplace_back([&, s]() {
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return may_verify[s] != 0; }); if (may_verify[s] == 2) return; }
                result &R = res[s];
                u8 seed[32]; seed_of(s, seed);
                R.rc = xhe_batch_verify(ctxs[s], up[s], &R.v, seed);
                R.ran = true;
            });
        int32_t chain_rc = 0;
        for (u32 s = 0; s < n_ctx; s++) {
            char go = 2;                                // 1: verify, 2: nothing to verify (the chain broke in front of this range, or its upload failed)
            if (!chain_broken && chain_rc == 0) {
                if (up_rc[s] != 0) { chain_rc = up_rc[s]; chain_broken = true; }
                else if (!structural[s].accept) { res[s].v = structural[s]; res[s].ran = true; chain_broken = true; }      // (the plan parsed every record: not reached)
                else {
                    sub_view &v = sv[s];
                    for (u64 k = 0; k < v.bal_global.size(); k++) memcpy(&v.cts[64 * k], &cur[64ull * v.bal_global[k]], 64);
                    int32_t rc = s ? xhe_batch_set_ledger(ctxs[s], up[s], v.cts.data()) : 0;
                    xhe_verdict av{1, XHE_OK, -1};
                    if (rc == 0) rc = xhe_batch_apply(ctxs[s], up[s], &av);
                    if (rc == 0 && av.accept) {
                        std::vector<u8> after(64 * v.bal_global.size() + 64);
                        xhe_ledger_out o{after.data(), nullptr, nullptr};
                        rc = xhe_batch_read_ledger(ctxs[s], up[s], &o);
                        if (rc == 0) for (u64 k = 0; k < v.bal_global.size(); k++) memcpy(&cur[64ull * v.bal_global[k]], &after[64 * k], 64);
                    } else if (rc == 0) chain_broken = true;       // this range still goes through verification: it names the error
                    if (rc != 0) { chain_rc = rc; chain_broken = true; } else go = 1;
                }
            }
            { std::lock_guard<std::mutex> lk(mu); may_verify[s] = go; }
            cv.notify_all();
        }
        for (auto &th : pool) th.join();
        for (u32 s = 0; s < n_ctx; s++) if (up[s]) xhe_batch_free(ctxs[s], up[s]);
        if (chain_rc < 0) return chain_rc;
        final_cts.swap(cur);
    }
    u64 key = ACCEPT_KEY;
    for (u32 s = 0; s < n_ctx; s++) {
        if (res[s].rc < 0) return res[s].rc;
        if (res[s].ran) key = std::min(key, verdict_key(res[s].v, sh[s].global));
    }
    // (ranges) a chain that broke leaves ranges unverified: the range it broke in has reported an error, or the verdict is not accept
    if (pr.ranges) for (u32 s = 0; s < n_ctx; s++) if (!res[s].ran && key == ACCEPT_KEY) key = verdict_key(xhe_verdict{0, XHE_ERR_STATE, -1}, sh[s].global);
    xhe_verdict_from_key(key, verdict);
    if (!verdict->accept || !out) return 0;
    if (out->balance_cts) {
        if (pr.ranges) memcpy(out->balance_cts, final_cts.data(), 64 * in->n_balances);
        else {
            // a balance is written by the shard that holds its account's component; untouched accounts keep their input
            if (out->balance_cts != in->balance_cts) memmove(out->balance_cts, in->balance_cts, 64 * in->n_balances);
            for (u32 s = 0; s < n_ctx; s++)
                for (u64 k = 0; k < sv[s].bal_global.size(); k++) memcpy(out->balance_cts + 64ull * sv[s].bal_global[k], &res[s].cts[64 * k], 64);
        }
    }
    if (out->account_nonces) memmove(out->account_nonces, in->account_nonces, 8 * in->n_accounts);
    if (out->multisig_set_by) {
        for (u64 a = 0; a < in->n_accounts; a++) out->multisig_set_by[a] = -1;
        if (!pr.ranges)
            for (u32 s = 0; s < n_ctx; s++)
                for (u64 k = 0; k < sv[s].acc_global.size(); k++) {
                    const int64_t l = res[s].ms[k];
                    if (l >= 0) { int64_t &dst = out->multisig_set_by[sv[s].acc_global[k]]; dst = std::max(dst, (int64_t)sh[s].global[(size_t)l]); }
                }
    }
    return 0;
}

}  // extern "C"
