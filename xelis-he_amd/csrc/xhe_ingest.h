// Host ingest of a batch: parses the records, runs the checks of Transaction::pre_verify that are pure
// control logic (nonce, asset/commitment consistency, multisig structure, ledger look-ups;
// /root/reference/src/tx/verify.rs:210-225,259-292,301-303,354-360,401-418) and lays out the flat tables
// of xhe_verify_types.h for the kernels.  No group or field arithmetic happens here.
//
// The work is split by what depends on the order of the transactions.  Parsing a record, looking its accounts and
// (account, asset) slots up in the ledger and writing its descriptors depend on that record alone and run on all host
// threads; between them one light sequential walk over the batch hands out everything the order decides: which
// transaction touches an account key or a ledger slot first (its point is decoded once, the ledger scan starts from
// it), the position of every ledger operation inside its slot's run, the multisig configuration a transaction sees (a
// MultiSig transaction updates the configuration seen by the transactions after it, as in the reference's sequential
// pre_verify), the distinct signing keys, the transcript templates.  Each transaction's checks assume the transactions
// before it passed: the verdict takes the failure of the lowest transaction index, for which that assumption holds.
#pragma once
#include <stdio.h>
#include "xhe_settings.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory_resource>
#include <mutex>
#include <string>
#include <thread>
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

// A growable array of plain data that is not cleared when it is sized (the ingest writes every element it sizes; clearing
// hundreds of megabytes of descriptors first costs as much as writing them).
template <typename T> struct pod_vec {
    T *p = nullptr; size_t n = 0, cap = 0;
    pod_vec() {}
    pod_vec(const pod_vec &) = delete;
    pod_vec &operator=(const pod_vec &) = delete;
    ~pod_vec() { free(p); }
    void reserve(size_t c) { if (c > cap) { p = (T *)realloc(p, c * sizeof(T)); cap = c; } }
    void resize(size_t k) { reserve(k); n = k; }
    void assign(size_t k, const T &v) { resize(k); for (size_t i = 0; i < k; i++) p[i] = v; }
    void push_back(const T &v) { if (n == cap) reserve(cap ? 2 * cap : 64); p[n++] = v; }
    T *data() { return p; } const T *data() const { return p; }
    size_t size() const { return n; } bool empty() const { return n == 0; }
    T &operator[](size_t i) { return p[i]; } const T &operator[](size_t i) const { return p[i]; }
    T *begin() { return p; } T *end() { return p + n; } const T *begin() const { return p; } const T *end() const { return p + n; }
    // takes the other array's memory when that is the larger one (the other keeps this one's); both end up empty
    void trade_storage(pod_vec &o) { if (o.cap > cap) { std::swap(p, o.p); std::swap(cap, o.cap); } n = 0; o.n = 0; }
};

struct host_batch {
    u64 n_txs = 0;
    u64 bytes_len = 0, rec_bytes = 0, n_signer_keys = 0;      // device byte buffer: records || account keys || balance cts || multisig signers || spare
    u64 keys_base = 0, cts_base = 0, signers_base = 0;
    // What the ordered walk decides, per transaction, transfer and commitment: the device expands these records into the
    // descriptors, decode lists, ledger operations and signature descriptors (k_expand_descriptors, xhe_k_expand.hip).
    pod_vec<expand_tx> ex_txs;
    pod_vec<expand_xf> ex_xfs;
    pod_vec<expand_cm> ex_cms;
    std::vector<expand_ms> ex_mss;         // co-signatures, in transaction order
    std::vector<dev_dec> shared_decs;      // points shared between transactions (account keys, ledger ciphertexts, signer keys): decs[] behind the blocks
    std::vector<u32> shared_early, shared_key;     // their positions, by decode list
    std::vector<u32> slot_pos, slot_base;  // per used slot: position of its initial C (D = +1); first of its operations (n_slots + 1 entries)
    std::vector<u32> emit_slot;            // per commitment: the used slot its emitted ciphertext belongs to
    std::vector<u32> sig_keys;             // point index of every distinct key a signature of the batch is checked against
    // sizes of what the device writes
    u32 n_cm = 0, n_xf = 0, n_sigs = 0, n_ops = 0, n_decs = 0, n_early = 0, n_key = 0, n_bulk = 0, n_segs = 0, n_key_terms = 0;
    u32 blk_end = 0, early_end = 0, key_end = 0;       // where the shared points start in decs[] and in the two lists
    std::vector<u32> slot_of_used;         // used slot -> ledger balance index
    std::vector<u32> ms_hash_tx;           // transactions whose pre-image BLAKE3 digest is needed
    std::vector<std::vector<u32>> lg_groups;   // transactions by lg (index = lg)
    std::vector<int64_t> multisig_set_by;  // per account
    bool multisig_updated = false;         // a transaction of the batch changed an account's multisig configuration
    std::vector<u32> tmpl;                 // transcript templates of the batch's shapes
    // Terms of the multiscalar check that share a point are merged before the bucket sort (their scalars add): pairs (a value
    // commitment's sigma-proof term and its range-proof term, one per commitment and transfer) and runs (the terms on each
    // account key: source and destination keys recur in every proof of an account's transactions), group g of the runs being
    // key_terms[key_group_off[g] .. key_group_off[g + 1]) on the device
    std::vector<u32> key_group_off;
    std::vector<u32> serial_txs;           // transactions whose transcript runs in the serial kernel
    std::vector<u32> tr_list;              // the others, sorted by template, each template padded to whole tiles of 64 (0xffffffff)
    std::vector<transcript_tile> tr_tiles; // one per 64 entries of tr_list
    transcript_plan tr_plan;
    u32 n_static = 0, nm_cap = 0;
    u32 n_points = 0;                      // static + decoded + emitted
    u32 emit_base = 0, n_emit = 0;         // emitted ciphertext points (2 per commitment)
    u32 n_chal = 0;
    u32 n_terms = 0, sigma_terms = 0, static_term_off = 0;
    bool bad_ledger = false;               // the ledger view itself is inconsistent (indices out of range): a usage error, not a verdict
    xhe_verdict structural{1, XHE_OK, -1};
    // The large tables are written once per batch and their memory is worth keeping: a freed batch leaves it to the next one
    // (fresh pages cost a fault each, more than writing them).
    void trade_storage(host_batch &o) { ex_txs.trade_storage(o.ex_txs); ex_xfs.trade_storage(o.ex_xfs); ex_cms.trade_storage(o.ex_cms); }
};

// Memory of the ingest's temporary tables (some forty arrays of one entry per transaction, transfer or commitment), kept by
// the context between uploads: fresh memory from the allocator costs a page fault per 4 KiB on first touch, on the one thread
// that sizes the arrays -- more than the passes that fill them.  A bump allocator over blocks that are kept; reset() makes one
// block of what the last use needed.
struct ingest_arena : std::pmr::memory_resource {
    struct block { char *p; size_t cap; };
    std::vector<block> blocks;
    size_t cur = 0, off = 0;
    ~ingest_arena() override { for (auto &b : blocks) free(b.p); }
    void reset() {
        if (blocks.size() > 1) {
            size_t total = 0;
            for (auto &b : blocks) { total += b.cap; free(b.p); }
            blocks.clear();
            char *p = (char *)malloc(total);
            if (p) blocks.push_back({p, total});
        }
        cur = 0; off = 0;
    }
    void *do_allocate(size_t bytes, size_t align) override {
        if (align < 64) align = 64;
        for (;; cur++, off = 0) {
            if (cur == blocks.size()) {
                const size_t cap = std::max<size_t>(bytes + align, std::max<size_t>(4u << 20, blocks.empty() ? 0 : 2 * blocks.back().cap));
                char *p = (char *)malloc(cap);
                if (!p) throw std::bad_alloc();
                blocks.push_back({p, cap});
            }
            const size_t at = (((size_t)blocks[cur].p + off + align - 1) & ~(align - 1)) - (size_t)blocks[cur].p;
            if (at + bytes <= blocks[cur].cap) { off = at + bytes; return blocks[cur].p + at; }
        }
    }
    void do_deallocate(void *, size_t, size_t) override {}
    bool do_is_equal(const std::pmr::memory_resource &o) const noexcept override { return this == &o; }
};

// Host threads that stay with a context between uploads: starting fifteen threads costs more than a pass of the ingest over a
// few thousand records takes.  run(parts, fn) calls fn(1) .. fn(parts - 1) on the workers and fn(0) on the calling thread and
// returns when all are done; one caller at a time.
struct ingest_workers {
    std::vector<std::thread> threads;
    std::mutex mu; std::condition_variable cv_work, cv_done;
    const std::function<void(unsigned)> *job = nullptr;
    u64 generation = 0; unsigned parts = 0, pending = 0; bool quit = false;
    void start(unsigned n) {
        for (unsigned t = 0; t < n; t++)
            threads.emplace_back([this, t]() {
                u64 seen = 0;
                for (;;) {
                    const std::function<void(unsigned)> *fn;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_work.wait(lk, [&]() { return quit || generation != seen; });
                        if (quit) return;
                        seen = generation; fn = job;
                        if (t + 1 >= parts) continue;          // fewer parts than workers this time
                    }
                    (*fn)(t + 1);
                    { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) cv_done.notify_one(); }
                }
            });
    }
    void run(unsigned n_parts, const std::function<void(unsigned)> &fn) {
        if (n_parts > threads.size() + 1) n_parts = (unsigned)threads.size() + 1;
        if (n_parts > 1) { { std::lock_guard<std::mutex> lk(mu); job = &fn; parts = n_parts; pending = n_parts - 1; generation++; } cv_work.notify_all(); }
        fn(0);
        if (n_parts > 1) { std::unique_lock<std::mutex> lk(mu); cv_done.wait(lk, [&]() { return pending == 0; }); }
    }
    unsigned size() const { return (unsigned)threads.size(); }
    ~ingest_workers() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_work.notify_all();
        for (auto &th : threads) th.join();
    }
};

// fn(lo, hi) over [0, count) on up to `threads` host threads (the calling thread takes the first range); with workers, on them
template <typename F>
static inline void ingest_parallel(u64 count, unsigned threads, F fn, ingest_workers *workers = nullptr) {
    if (threads <= 1 || count < 2048) { if (count) fn((u64)0, count); return; }
    if (workers && workers->size()) {
        threads = std::min(threads, workers->size() + 1);
        const u64 per = (count + threads - 1) / threads;
        const std::function<void(unsigned)> part = [&](unsigned t) { const u64 lo = t * per, hi = std::min(count, lo + per); if (lo < hi) fn(lo, hi); };
        workers->run(threads, part);
        return;
    }
    const u64 per = (count + threads - 1) / threads;
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; t++) {
        const u64 lo = t * per, hi = std::min(count, lo + per);
        if (lo < hi) pool.emplace_back([=]() { fn(lo, hi); });
    }
    fn((u64)0, std::min(count, per));
    for (auto &th : pool) th.join();
}
// host threads of the ingest: XHE_INGEST_THREADS, else the machine's threads, at most 16
static inline unsigned ingest_thread_count() {
    if (const char *e = setting("XHE_INGEST_THREADS")) { unsigned v = (unsigned)strtoul(e, nullptr, 10); if (v >= 1 && v <= 256) return v; }
    unsigned hc = std::thread::hardware_concurrency();
    return hc == 0 ? 1 : (hc > 16 ? 16 : hc);
}

// The ledger view's keys as hash tables: account key -> account index, (asset, account index) -> balance index.  They depend on
// the view alone, so the parts of a batch that is verified in parts share one.  Open addressing over indices into the view's own
// arrays (a node-based map costs an allocation per account and per balance: 8 ms for 37 500 accounts with 50 000 balances, on the
// one thread every part waits for).  The first of two equal keys counts.
struct ledger_index {
    static constexpr u32 ABSENT = 0xffffffffu;
    const xhe_ledger_view *lv = nullptr;
    std::vector<u32> acc_tab, bal_tab;
    u64 acc_mask = 0, bal_mask = 0;
    static u64 mix(u64 h) { h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32; return h; }
    static u64 hash_key(const u8 *k) { u64 a, b; memcpy(&a, k, 8); memcpy(&b, k + 24, 8); return mix(a ^ (b * 0x9e3779b97f4a7c15ull)); }
    static u64 hash_slot(const u8 *asset, u32 account) { u64 a, b; memcpy(&a, asset, 8); memcpy(&b, asset + 24, 8); return mix(a ^ (b * 0x9e3779b97f4a7c15ull) ^ ((u64)(account + 1) * 0xd6e8feb86659fd93ull)); }
    // false: the view's own indices are inconsistent (an account index out of range, multisig offsets that do not ascend from 0)
    static bool view_consistent(const xhe_ledger_view &lv) {
        for (u64 i = 0; i < lv.n_balances; i++) if (lv.balance_account[i] >= lv.n_accounts) return false;
        for (u64 i = 0; i < lv.n_multisig; i++)
            if (lv.multisig_account[i] >= lv.n_accounts || lv.multisig_offsets[i + 1] < lv.multisig_offsets[i]) return false;
        return !(lv.n_multisig && lv.multisig_offsets[0] != 0);
    }
    u32 find_account(const u8 *key) const {
        for (u64 h = hash_key(key) & acc_mask;; h = (h + 1) & acc_mask) {
            const u32 i = acc_tab[h];
            if (i == ABSENT || !memcmp(lv->account_keys + 32ull * i, key, 32)) return i;
        }
    }
    u32 find_slot(const u8 *asset, u32 account) const {
        for (u64 h = hash_slot(asset, account) & bal_mask;; h = (h + 1) & bal_mask) {
            const u32 i = bal_tab[h];
            if (i == ABSENT || (lv->balance_account[i] == account && !memcmp(lv->balance_assets + 32ull * i, asset, 32))) return i;
        }
    }
    void build(const xhe_ledger_view &view) {
        lv = &view;
        u64 cap = 16; while (cap < 2 * view.n_accounts + 2) cap <<= 1;
        acc_tab.assign(cap, ABSENT); acc_mask = cap - 1;
        for (u64 i = 0; i < view.n_accounts; i++) {
            const u8 *key = view.account_keys + 32 * i;
            u64 h = hash_key(key) & acc_mask;
            while (acc_tab[h] != ABSENT && memcmp(view.account_keys + 32ull * acc_tab[h], key, 32)) h = (h + 1) & acc_mask;
            if (acc_tab[h] == ABSENT) acc_tab[h] = (u32)i;
        }
        cap = 16; while (cap < 2 * view.n_balances + 2) cap <<= 1;
        bal_tab.assign(cap, ABSENT); bal_mask = cap - 1;
        for (u64 i = 0; i < view.n_balances; i++) {
            const u8 *asset = view.balance_assets + 32 * i; const u32 account = view.balance_account[i];
            u64 h = hash_slot(asset, account) & bal_mask;
            while (bal_tab[h] != ABSENT && !(view.balance_account[bal_tab[h]] == account && !memcmp(view.balance_assets + 32ull * bal_tab[h], asset, 32))) h = (h + 1) & bal_mask;
            if (bal_tab[h] == ABSENT) bal_tab[h] = (u32)i;
        }
    }
};

// returns false when a record is malformed (hb.structural says which) or the ledger view is inconsistent (hb.bad_ledger).
// shared_index (optional): the maps of this very ledger view, built by the caller.
static inline bool ingest_batch(host_batch &hb, const u8 *records, const u64 *offsets, u64 n, const xhe_ledger_view &lv, ingest_workers *workers = nullptr,
                                ingest_arena *arena = nullptr, const ledger_index *shared_index = nullptr) {
    std::pmr::memory_resource *mr = arena ? static_cast<std::pmr::memory_resource *>(arena) : std::pmr::new_delete_resource();
    if (arena) arena->reset();
    constexpr u32 NONE = 0xffffffffu;
    const unsigned threads = ingest_thread_count();
    const bool show_times = setting("XHE_UPLOAD_TIMES") != nullptr;      // phase times on stderr
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!show_times) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[xhe ingest] %-28s %.3f ms\n", what, std::chrono::duration<double>(now - t_last).count() * 1e3);
        t_last = now;
    };
    hb.n_txs = n;
    // ---- the ledger view's own indices ----
    if (!shared_index && !ledger_index::view_consistent(lv)) { hb.bad_ledger = true; return false; }

    // ---- pass A (parallel): parse every record ----
    std::pmr::vector<tx_record> recs(n, mr);
    {
        std::atomic<u64> first_bad(~0ull);
        ingest_parallel(n, threads, [&](u64 lo, u64 hi) {
            for (u64 i = lo; i < hi; i++) {
                // lengths and layout only: the scalars' range check runs on the device with the descriptor expansion
                bool ok = offsets[i + 1] >= offsets[i] && record_parse(recs[i], records + offsets[i], offsets[i + 1] - offsets[i], false);
                if (ok && recs[i].type == 3) {         // a HashMap cannot hold one asset twice (src/tx/mod.rs:78)
                    const tx_record &t = recs[i];
                    for (u32 a = 0; a < t.n_items && ok; a++)
                        for (u32 b = a + 1; b < t.n_items; b++)
                            if (!memcmp(t.payload() + 32 + 40 * a, t.payload() + 32 + 40 * b, 32)) { ok = false; break; }
                }
                if (!ok) {                             // the first bad record of a range is enough: the verdict names the smallest index
                    u64 seen = first_bad.load();
                    while (i < seen && !first_bad.compare_exchange_weak(seen, i)) {}
                    break;
                }
            }
        }, workers);
        if (first_bad.load() != ~0ull) {
            // nothing of this batch reaches the device: a record in front of the bad one with a scalar that is not canonical is
            // malformed too and comes first (the records in front of the first bad one all parsed)
            const u64 bad = first_bad.load();
            ingest_parallel(bad, threads, [&](u64 lo, u64 hi) {
                for (u64 i = lo; i < hi; i++)
                    if (!record_scalars_canonical(recs[i])) { u64 seen = first_bad.load(); while (i < seen && !first_bad.compare_exchange_weak(seen, i)) {} break; }
            }, workers);
            hb.structural = {0, XHE_ERR_MALFORMED, (int64_t)first_bad.load()};
            return false;
        }
    }
    lap("records parsed");
    u32 max_lg = 6;
    for (u64 i = 0; i < n; i++) if (recs[i].lg() > max_lg) max_lg = recs[i].lg();
    if (max_lg > 15) max_lg = 15;      // party capacity 512 (src/proofs.rs:20); larger proofs fail in the range-proof stage
    hb.nm_cap = 1u << max_lg;
    hb.n_static = 2 + 2 * hb.nm_cap;

    // ---- byte buffer ----
    const u64 rec_bytes = offsets[n];
    hb.keys_base = (rec_bytes + 63) & ~63ull;
    hb.cts_base = hb.keys_base + 32 * lv.n_accounts;
    const u64 n_signers = lv.n_multisig ? lv.multisig_offsets[lv.n_multisig] : 0;
    hb.signers_base = hb.cts_base + 64 * lv.n_balances;
    // the device's byte buffer: records | account keys | ledger ciphertexts | multisig signer keys | 64 spare bytes.  It is
    // assembled from the caller's arrays when the batch is uploaded; no host copy is kept.
    hb.rec_bytes = rec_bytes; hb.n_signer_keys = n_signers;
    hb.bytes_len = hb.signers_base + 32 * n_signers + 64;

    // ---- ledger key structure (read-only from here on) ----
    ledger_index own_index;
    if (!shared_index) own_index.build(lv);
    const ledger_index &index = shared_index ? *shared_index : own_index;
    hb.multisig_set_by.assign(lv.n_accounts, -1);

    // ---- prefix sums over the shapes: transfers, commitments, terms, challenges, segments ----
    std::pmr::vector<u32> xf_base(n + 1, mr), cm_base(n + 1, mr), cm_term(n, mr), xf_term(n, mr), seg_base(n + 1, mr);
    {
        u32 x = 0, c = 0, term = 0, sg = 0;
        for (u64 i = 0; i < n; i++) {
            const u32 T = (u32)recs[i].transfer_off.size(), A = recs[i].n_commitments;
            xf_base[i] = x; cm_base[i] = c; seg_base[i] = sg;
            cm_term[i] = term; term += 7 * A; xf_term[i] = term; term += 8 * T;      // sigma terms first and contiguous: the sigma-only check uses them
            x += T; c += A; sg += 2 + A + 2 * T;
        }
        xf_base[n] = x; cm_base[n] = c; seg_base[n] = sg;
        hb.sigma_terms = term;
    }
    const u32 n_xf = xf_base[n], n_cm = cm_base[n];

    // ---- pass B (parallel): ledger look-ups of every transaction ----
    std::pmr::vector<u32> src_acc(n, mr), xf_acc(n_xf, mr), xf_bal(n_xf, mr), cm_bal(n_cm, mr);
    std::pmr::vector<u16> xf_cm(n_xf, mr);                      // the source commitment with the transfer's asset (0xffff: none), so that the ordered walk reads no record bytes
    std::pmr::vector<u64> cm_amount(n_cm, mr);                  // plain amount leaving the slot besides the fee: burn, contract deposit
    ingest_parallel(n, threads, [&](u64 lo, u64 hi) {
        for (u64 i = lo; i < hi; i++) {
            const tx_record &t = recs[i];
            // the keys this pass reads lie one to a cache line all over the records: ask for those of the record after next now
            if (i + 2 < hi) {
                const tx_record &f = recs[i + 2];
                __builtin_prefetch(f.source());
                for (u32 k = 0; k < (u32)f.transfer_off.size() && k < 16; k++) __builtin_prefetch(f.transfer(k) + 32);
                for (u32 k = 0; k < f.n_commitments && k < 4; k++) __builtin_prefetch(f.commitment(k));
            }
            const u32 sa = index.find_account(t.source());
            src_acc[i] = sa;
            for (u32 k = 0; k < t.n_commitments; k++) {
                const u8 *asset = t.commitment(k);
                u32 b = NONE;
                if (sa != NONE) b = index.find_slot(asset, sa);
                cm_bal[cm_base[i] + k] = b;
                u64 amount = 0;
                if (t.type == 1) { if (!memcmp(asset, t.payload(), 32)) amount = rec_be64(t.payload() + 32); }
                else if (t.type == 3) { for (u32 x = 0; x < t.n_items; x++) if (!memcmp(asset, t.payload() + 32 + 40 * x, 32)) amount = rec_be64(t.payload() + 64 + 40 * x); }
                cm_amount[cm_base[i] + k] = amount;
            }
            const u8 *last_dest = nullptr; u32 last_acc = NONE;
            for (u32 k = 0; k < (u32)t.transfer_off.size(); k++) {
                const u8 *asset = t.transfer(k), *dest = asset + 32;
                u32 da;
                if (last_dest && !memcmp(last_dest, dest, 32)) da = last_acc;
                else { da = index.find_account(dest); last_dest = dest; last_acc = da; }
                u32 b = NONE;
                if (da != NONE) b = index.find_slot(asset, da);
                xf_acc[xf_base[i] + k] = da; xf_bal[xf_base[i] + k] = b;
                u16 same = 0xffff;
                for (u32 c = 0; c < t.n_commitments && c < 0xffff; c++) if (!memcmp(asset, t.commitment(c), 32)) { same = (u16)c; break; }
                xf_cm[xf_base[i] + k] = same;
            }
        }
    }, workers);

    lap("ledger look-ups");
    // ---- decode positions: one block per transaction, then the points shared between transactions ----
    // block: range proof (4 + 2 lg) | source key if the ledger does not know it | per transfer C, D_s, D_r, Y_0, Y_1, Y_2 |
    // per commitment C_new, Y_0, Y_1, Y_2 | destination keys the ledger does not know
    std::pmr::vector<u32> blk(n + 1, mr), early_base(n + 1, mr), key_base(n + 1, mr), bulk_base(n + 1, mr);
    {
        u32 p = 0, e = 0, k = 0, b = 0;
        for (u64 i = 0; i < n; i++) {
            const u32 T = (u32)recs[i].transfer_off.size(), A = recs[i].n_commitments, lg = recs[i].lg();
            u32 unknown_dests = 0;
            for (u32 x = 0; x < T; x++) if (xf_acc[xf_base[i] + x] == NONE) unknown_dests++;
            const u32 own_key = src_acc[i] == NONE ? 1u : 0u;
            blk[i] = p; early_base[i] = e; key_base[i] = k; bulk_base[i] = b;
            p += 4 + 2 * lg + own_key + 6 * T + 4 * A + unknown_dests;
            e += 3 * T; k += own_key; b += 4 + 2 * lg + 3 * T + 4 * A + unknown_dests;
        }
        blk[n] = p; early_base[n] = e; key_base[n] = k; bulk_base[n] = b;
    }

    lap("prefix sums");
    // ---- pass C (sequential): everything the order of the transactions decides ----
    std::vector<dev_dec> &shared_decs = hb.shared_decs;    // positions blk[n] + index
    std::vector<u32> &shared_early = hb.shared_early, &shared_key = hb.shared_key;
    shared_decs.clear(); shared_early.clear(); shared_key.clear();
    auto new_shared = [&](u64 off, u32 early) -> u32 {
        const u32 pos = blk[n] + (u32)shared_decs.size();
        shared_decs.push_back({off});
        (early == 1 ? shared_early : shared_key).push_back(pos);
        return pos;
    };
    std::pmr::vector<u32> acc_pos(lv.n_accounts, NONE, mr), acc_keyslot(lv.n_accounts, NONE, mr), used_slot(lv.n_balances, NONE, mr);
    std::vector<u32> &slot_pos = hb.slot_pos; slot_pos.clear();
    std::pmr::vector<u32> ops_count(mr);                        // per used slot: operations so far (slot_pos: position of its initial C, D = +1)
    std::unordered_map<u64, u32> signer_pos;           // byte offset of a signer key -> position
    std::unordered_map<u32, u32> pos_keyslot;          // position of a signer / unknown-source key -> position in sig_keys
    struct ms_cfg { u64 first_off; u32 n_signers, threshold; };       // signer k's key: 32 bytes at first_off + 32 k (the ledger view's and a record's are both contiguous)
    std::vector<ms_cfg> cfgs;
    cfgs.reserve(lv.n_multisig + 16);
    std::pmr::vector<int32_t> cfg_of(lv.n_accounts, -1, mr);
    for (u64 i = 0; i < lv.n_multisig; i++) {
        cfg_of[lv.multisig_account[i]] = (int32_t)cfgs.size();
        cfgs.push_back({hb.signers_base + 32ull * lv.multisig_offsets[i], lv.multisig_offsets[i + 1] - lv.multisig_offsets[i], lv.multisig_threshold[i]});
    }
    auto account_position = [&](u32 a) { if (acc_pos[a] == NONE) acc_pos[a] = new_shared(hb.keys_base + 32ull * a, 2); return acc_pos[a]; };
    auto touch_slot = [&](u32 bal, bool *fresh) {
        if (used_slot[bal] != NONE) { *fresh = false; return used_slot[bal]; }
        const u32 u = (u32)slot_pos.size();
        used_slot[bal] = u;
        const u32 p = new_shared(hb.cts_base + 64ull * bal, 1);
        new_shared(hb.cts_base + 64ull * bal + 32, 1);
        slot_pos.push_back(p); hb.slot_of_used.push_back(bal); ops_count.push_back(1);       // operation 0: the initial balance
        *fresh = true;
        return u;
    };
    auto keyslot_of_pos = [&](u32 pos) {
        auto it = pos_keyslot.find(pos);
        if (it != pos_keyslot.end()) return it->second;
        const u32 s = (u32)hb.sig_keys.size();
        hb.sig_keys.push_back(hb.n_static + pos); pos_keyslot.emplace(pos, s);
        return s;
    };
    // what pass C leaves for pass D
    std::pmr::vector<u32> src_pos(n, mr), src_keyslot(n, mr), sig_base(n + 1, mr), step_err(n, ERRKEY_NONE, mr), tmpl_of_tx(n, mr), rp_row(n, 0, mr), hash_idx(n, NONE, mr);
    std::pmr::vector<u32> cm_slot(n_cm, NONE, mr), cm_local(n_cm, 0, mr), xf_slot(n_xf, NONE, mr), xf_local(n_xf, 0, mr), xf_dest_pos(n_xf, NONE, mr);
    std::pmr::vector<u8> cm_fresh(n_cm, 0, mr), xf_fresh(n_xf, 0, mr);
    std::pmr::vector<u32> key_cnt(lv.n_accounts, 0, mr), src_keylocal(n, mr), xf_keylocal(n_xf, NONE, mr);      // terms on each account's key, and where a transaction's come in its run
    std::vector<expand_ms> &ms_entries = hb.ex_mss; ms_entries.clear();       // co-signatures to verify, in transaction order
    std::unordered_map<transcript_shape, u32, transcript_shape_hash> tmpl_map;
    transcript_shape last_shape{255, 0, 0, 0, 0, 0}; u32 last_tmpl = TT_NONE;
    // a batch of mixed transactions changes its shape with nearly every transaction and knows a handful of shapes: the ones
    // seen last are found in a small direct-mapped table before the map is asked
    struct recent_shape { transcript_shape shape{255, 0, 0, 0, 0, 0}; u32 tmpl = TT_NONE; } recent[16];
    hb.lg_groups.assign(16, {});
    {
        u32 sigs = 0;
        for (u64 i = 0; i < n; i++) {
            const tx_record &t = recs[i];
            const u32 T = (u32)t.transfer_off.size(), A = t.n_commitments, lg = t.lg();
            const u64 rec = offsets[i];
            auto raise = [&](u32 key) { if (key < step_err[i]) step_err[i] = key; };
            if (lg <= 15) { rp_row[i] = (u32)hb.lg_groups[lg].size(); hb.lg_groups[lg].push_back((u32)i); }
            {
                transcript_shape sh{t.type, A, T, t.n_items, t.m(), lg};
                if (!(sh == last_shape)) {
                    last_shape = sh;
                    recent_shape &slot = recent[(sh.type * 5u + sh.n_commitments * 3u + sh.n_transfers + sh.n_items * 7u) & 15u];
                    if (slot.shape == sh) last_tmpl = slot.tmpl;
                    else {
                        if (!transcript_shape_supported(sh)) last_tmpl = TT_NONE;
                        else { auto it = tmpl_map.find(sh); last_tmpl = it != tmpl_map.end() ? it->second : (tmpl_map[sh] = build_transcript_template(hb.tmpl, sh)); }
                        slot.shape = sh; slot.tmpl = last_tmpl;
                    }
                }
                tmpl_of_tx[i] = last_tmpl;
                if (last_tmpl == TT_NONE) hb.serial_txs.push_back((u32)i);
            }
            // source key and its place among the distinct signing keys
            const u32 sa = src_acc[i];
            if (sa != NONE) { src_keylocal[i] = key_cnt[sa]; key_cnt[sa] += A + T; }
            sig_base[i] = sigs; sigs++;
            if (sa != NONE) {
                src_pos[i] = account_position(sa);
                if (acc_keyslot[sa] == NONE) { acc_keyslot[sa] = (u32)hb.sig_keys.size(); hb.sig_keys.push_back(hb.n_static + src_pos[i]); }
                src_keyslot[i] = acc_keyslot[sa];
            } else {
                src_pos[i] = blk[i] + 4 + 2 * lg;      // the key inside the record: a point of its own
                src_keyslot[i] = (u32)hb.sig_keys.size(); hb.sig_keys.push_back(hb.n_static + src_pos[i]);
            }
            // multisig (verify.rs:259-292)
            const int32_t ci = sa != NONE ? cfg_of[sa] : -1;
            if (ci >= 0) {
                const ms_cfg &cfg = cfgs[(size_t)ci];
                if (!t.has_multisig() || t.n_multisig == 0 || t.n_multisig != cfg.threshold) raise(errkey(ST_MULTISIG, 0, 0, XHE_ERR_FORMAT));
                else {
                    hash_idx[i] = (u32)hb.ms_hash_tx.size(); hb.ms_hash_tx.push_back((u32)i);
                    for (u32 k = 0; k < t.n_multisig; k++) {
                        const u8 idx = t.ms_entry(k)[0];
                        for (u32 j = 0; j < t.n_multisig; j++) if (j != k && t.ms_entry(j)[0] == idx) raise(errkey(ST_MULTISIG, k, 1, XHE_ERR_FORMAT));
                        if (idx < cfg.n_signers) {
                            const u64 koff = cfg.first_off + 32ull * idx;
                            auto sp = signer_pos.find(koff);
                            const u32 kp = sp != signer_pos.end() ? sp->second : (signer_pos[koff] = new_shared(koff, 2));
                            ms_entries.push_back({koff, (u32)i, k, kp, keyslot_of_pos(kp)});
                            sigs++;
                        }
                    }
                }
            } else if (t.has_multisig()) raise(errkey(ST_MULTISIG, 0, 0, XHE_ERR_FORMAT));
            // commitments (verify.rs:296-341): the slot, and its operations: minus every transfer of the asset, minus the plain
            // amount (burn, contract deposit), minus the fee with the emission of the sender's new ciphertext
            for (u32 k = 0; k < A; k++) {
                const u32 c = cm_base[i] + k, bal = cm_bal[c];
                if (bal == NONE) continue;
                bool fresh; const u32 u = touch_slot(bal, &fresh);
                cm_slot[c] = u; cm_fresh[c] = fresh ? 1 : 0; cm_local[c] = ops_count[u];
                u32 cnt = 1;
                for (u32 x = 0; x < T; x++) if (xf_cm[xf_base[i] + x] == k) cnt++;
                if (cm_amount[c]) cnt++;
                ops_count[u] += cnt;
            }
            // transfers (verify.rs:345-395)
            for (u32 k = 0; k < T; k++) {
                const u32 x = xf_base[i] + k, da = xf_acc[x];
                if (da == NONE) continue;
                xf_dest_pos[x] = account_position(da);
                xf_keylocal[x] = key_cnt[da]++;
                if (xf_bal[x] == NONE) continue;
                bool fresh; const u32 u = touch_slot(xf_bal[x], &fresh);
                xf_slot[x] = u; xf_fresh[x] = fresh ? 1 : 0; xf_local[x] = ops_count[u]++;
            }
            // MultiSig payload rules and state update (verify.rs:401-428)
            if (t.type == 2) {
                if (!record_multisig_payload_ok(t)) raise(errkey(ST_TRANSFER, 0, 0, XHE_ERR_FORMAT));
                else if (sa != NONE) {
                    if (t.n_items == 0) cfg_of[sa] = -1;
                    else {
                        cfg_of[sa] = (int32_t)cfgs.size(); cfgs.push_back({rec + t.payload_off + 1, t.n_items, t.payload()[0]});
                    }
                    hb.multisig_set_by[sa] = (int64_t)i; hb.multisig_updated = true;
                }
            }
        }
        sig_base[n] = sigs;
    }
    lap("ordered walk");
    const u32 n_slots = (u32)slot_pos.size();
    std::vector<u32> &slot_base = hb.slot_base; slot_base.resize(n_slots + 1);
    { u32 o = 0; for (u32 u = 0; u < n_slots; u++) { slot_base[u] = o; o += ops_count[u]; } slot_base[n_slots] = o; }

    // ---- sizes ----
    hb.n_cm = n_cm; hb.n_xf = n_xf; hb.n_sigs = sig_base[n]; hb.n_segs = seg_base[n]; hb.n_ops = slot_base[n_slots];
    hb.blk_end = blk[n]; hb.early_end = early_base[n]; hb.key_end = key_base[n];
    hb.n_decs = blk[n] + (u32)shared_decs.size();
    hb.n_early = early_base[n] + (u32)shared_early.size(); hb.n_key = key_base[n] + (u32)shared_key.size(); hb.n_bulk = bulk_base[n];
    hb.emit_base = hb.n_static + hb.n_decs;
    hb.n_emit = 2 * n_cm;
    hb.n_points = hb.emit_base + hb.n_emit;
    // challenges and range-proof terms follow in transaction order
    std::pmr::vector<u32> chal_off(n, mr), rp_term(n, mr);
    {
        u32 chal = 0, term = hb.sigma_terms;
        for (u64 i = 0; i < n; i++) {
            const u32 T = (u32)recs[i].transfer_off.size(), A = recs[i].n_commitments, lg = recs[i].lg();
            chal_off[i] = chal; chal += 2 * A + 2 * T + 4 + lg;
            rp_term[i] = term; term += 4 + 2 * lg + A + T;
        }
        hb.n_chal = chal; hb.static_term_off = term; hb.n_terms = term + hb.n_static;
    }
    // runs of terms per account key (accounts with a single term need no merging)
    std::pmr::vector<u32> key_run(lv.n_accounts, NONE, mr);
    {
        u32 o = 0;
        hb.key_group_off.clear();
        for (u64 a = 0; a < lv.n_accounts; a++) if (key_cnt[a] >= 2) { key_run[a] = o; hb.key_group_off.push_back(o); o += key_cnt[a]; }
        hb.key_group_off.push_back(o);
        hb.n_key_terms = o;
    }
    // co-signatures by transaction: ms_entries is in transaction order, so a transaction's entries follow its main signature
    std::pmr::vector<u32> ms_first(n + 1, 0, mr);
    { for (auto &e : ms_entries) ms_first[e.tx + 1]++; for (u64 i = 0; i < n; i++) ms_first[i + 1] += ms_first[i]; }
    hb.emit_slot.assign(n_cm, 0);
    hb.ex_txs.resize(n); hb.ex_xfs.resize(n_xf); hb.ex_cms.resize(n_cm);

    lap("tables sized");
    // ---- pass D (parallel): the records the device expands into descriptors ----
    ingest_parallel(n, threads, [&](u64 lo, u64 hi) {
        for (u64 i = lo; i < hi; i++) {
            const tx_record &t = recs[i];
            const u32 T = (u32)t.transfer_off.size(), A = t.n_commitments;
            const u32 sa = src_acc[i];
            expand_tx e;
            memset(&e, 0, sizeof e);
            e.rec = offsets[i]; e.fee = t.fee; e.nonce = t.nonce;
            e.pre_off = t.pre_off; e.pre_len = t.pre_len; e.payload_off = t.payload_off; e.rp_off = t.rp_off; e.cm_off = t.cm_off; e.ms_off = t.ms_off; e.sig_off = t.sig_off;
            e.m = t.m();
            e.hash_idx = hash_idx[i]; e.xfer_first = xf_base[i]; e.cm_first = cm_base[i]; e.chal_off = chal_off[i]; e.rp_term_off = rp_term[i]; e.rp_row = rp_row[i];
            e.tmpl_off = tmpl_of_tx[i]; e.seg_off = seg_base[i]; e.step_err = step_err[i];
            e.blk = blk[i]; e.early_base = early_base[i]; e.key_base = key_base[i]; e.bulk_base = bulk_base[i];
            e.src_pos = src_pos[i]; e.sig_base = sig_base[i]; e.src_keyslot = src_keyslot[i];
            e.ms_first = ms_first[i]; e.ms_count = ms_first[i + 1] - ms_first[i];
            e.cm_term = cm_term[i]; e.xf_term = xf_term[i];
            e.src_key_term = (sa != NONE && key_run[sa] != NONE) ? key_run[sa] + src_keylocal[i] : NONE;
            e.n_transfers = (u16)T; e.n_commitments = t.n_commitments; e.n_multisig = t.n_multisig; e.n_items = t.n_items;
            e.type = t.type; e.lg = (u8)t.lg(); e.version = t.version; e.flags = t.flags;
            e.src_known = sa != NONE; e.nonce_ok = sa != NONE && lv.account_nonces[sa] == t.nonce; e.assets_ok = record_commitment_assets_ok(t);
            hb.ex_txs[i] = e;
            for (u32 k = 0; k < A; k++) {
                const u32 ci = cm_base[i] + k, u = cm_slot[ci];
                expand_cm c;
                memset(&c, 0, sizeof c);
                c.amount = cm_amount[ci]; c.slot = u;
                if (u != NONE) { c.slot_pos = slot_pos[u]; c.op_pos = slot_base[u] + cm_local[ci]; c.fresh = cm_fresh[ci]; hb.emit_slot[ci] = u; }
                c.native = rec_is_zero32(t.commitment(k));
                hb.ex_cms[ci] = c;
            }
            for (u32 k = 0; k < T; k++) {
                const u32 xi = xf_base[i] + k, da = xf_acc[xi], u = xf_slot[xi];
                expand_xf x;
                memset(&x, 0, sizeof x);
                x.off = t.transfer_off[k]; x.len = t.transfer_len[k];
                x.has_account = da != NONE; x.dest_pos = xf_dest_pos[xi];
                x.slot = u;
                if (u != NONE) { x.slot_pos = slot_pos[u]; x.op_pos = slot_base[u] + xf_local[xi]; x.fresh = xf_fresh[xi]; }
                x.key_term = (da != NONE && xf_keylocal[xi] != NONE && key_run[da] != NONE) ? key_run[da] + xf_keylocal[xi] : NONE;
                x.cm = xf_cm[xi];
                hb.ex_xfs[xi] = x;
            }
        }
    }, workers);

    lap("descriptors written");
    // ---- transcript tiles: transactions grouped by template (first appearance order), 64 per tile ----
    {
        std::pmr::vector<u32> order(mr);                       // template offsets in order of first appearance
        std::unordered_map<u32, std::vector<u32>> by_tmpl;
        u32 last_to = TT_NONE; std::vector<u32> *last_vec = nullptr;
        for (u64 i = 0; i < n; i++) {
            const u32 to = tmpl_of_tx[i];
            if (to == TT_NONE) continue;
            if (to != last_to || !last_vec) {
                auto it = by_tmpl.find(to);
                if (it == by_tmpl.end()) { order.push_back(to); it = by_tmpl.emplace(to, std::vector<u32>()).first; }
                last_to = to; last_vec = &it->second;
            }
            last_vec->push_back((u32)i);
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
    lap("transcript tiles");
    return true;
}

}  // namespace xhe
