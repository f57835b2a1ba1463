// CPU build of the product's host/device arithmetic headers (xelis-he_amd/csrc/xhe_*.h), exposed with
// C linkage so the Python tests can compare them with the oracle and with hashlib on the CPU.
// Test support only: the shipped library does not contain this file.
#include <string.h>
#include "../../xelis-he_amd/csrc/xhe_ge.h"
#include "../../xelis-he_amd/csrc/xhe_hash.h"
#include "../../xelis-he_amd/csrc/xhe_record.h"
#include "../../xelis-he_amd/csrc/xhe_transcript_template.h"
using namespace xhe;

extern "C" {
void xt_fe_mul(const u8 a[32], const u8 b[32], u8 out[32]) { fe_tobytes(out, fe_mul(fe_frombytes(a), fe_frombytes(b))); }
void xt_fe_sq(const u8 a[32], u8 out[32]) { fe_tobytes(out, fe_sq(fe_frombytes(a))); }
void xt_fe_invert(const u8 a[32], u8 out[32]) { fe_tobytes(out, fe_invert(fe_frombytes(a))); }
// (a + b) * (a - b) exercises the lazy add/sub magnitudes
void xt_fe_addsub_mul(const u8 a[32], const u8 b[32], u8 out[32]) {
    fe x = fe_frombytes(a), y = fe_frombytes(b);
    fe_tobytes(out, fe_mul(fe_add(x, y), fe_sub(x, y)));
}
void xt_fe_roundtrip(const u8 a[32], u8 out[32]) { fe_tobytes(out, fe_frombytes(a)); }
void xt_sc_mul(const u8 a[32], const u8 b[32], u8 out[32]) { sc_tobytes(out, sc_mul(sc_frombytes(a), sc_frombytes(b))); }
void xt_sc_add(const u8 a[32], const u8 b[32], u8 out[32]) { sc_tobytes(out, sc_add(sc_frombytes(a), sc_frombytes(b))); }
void xt_sc_sub(const u8 a[32], const u8 b[32], u8 out[32]) { sc_tobytes(out, sc_sub(sc_frombytes(a), sc_frombytes(b))); }
void xt_sc_invert(const u8 a[32], u8 out[32]) { sc_tobytes(out, sc_invert(sc_frombytes(a))); }
void xt_sc_from_wide(const u8 a[64], u8 out[32]) { sc_tobytes(out, sc_from_wide_bytes(a)); }
// sum of products with lazy additions: a*b + c*d + a*d
void xt_sc_muladd(const u8 a[32], const u8 b[32], const u8 c[32], const u8 d[32], u8 out[32]) {
    sc A = sc_frombytes(a), B = sc_frombytes(b), C = sc_frombytes(c), D = sc_frombytes(d);
    sc r = sc_add(sc_add_lazy(sc_mul(A, B), sc_mul(C, D)), sc_mul(A, D));
    sc_tobytes(out, r);
}
int xt_sc_canonical(const u8 a[32]) { u32 w[8]; memcpy(w, a, 32); return sc_words_canonical(w); }
int xt_ristretto_decode(const u8 in[32], u8 out_xyzt[128]) {
    ge p; bool ok = ristretto_decode(p, in);
    fe_tobytes(out_xyzt, p.X); fe_tobytes(out_xyzt + 32, p.Y); fe_tobytes(out_xyzt + 64, p.Z); fe_tobytes(out_xyzt + 96, p.T);
    return ok;
}
void xt_ristretto_encode(const u8 xyzt[128], u8 out[32]) {
    ge p; p.X = fe_frombytes(xyzt); p.Y = fe_frombytes(xyzt + 32); p.Z = fe_frombytes(xyzt + 64); p.T = fe_frombytes(xyzt + 96);
    ristretto_encode(out, p);
}
void xt_from_uniform_bytes(const u8 in[64], u8 out[32]) { ristretto_encode(out, ristretto_from_uniform_bytes(in)); }
// out = enc(k*P) by double-and-add using ge_double/ge_add/ge_madd/ge_msub in rotation
int xt_scalarmul(const u8 k[32], const u8 p32[32], u8 out[32]) {
    ge p; if (!ristretto_decode(p, p32)) return -1;
    ge_niels pn = ge_to_niels_affine(p);
    ge r = ge_identity();
    int flip = 0;
    for (int i = 255; i >= 0; i--) {
        r = ge_double(r);
        if ((k[i >> 3] >> (i & 7)) & 1) {
            switch (flip++ % 3) {
                case 0: r = ge_add(r, p); break;
                case 1: r = ge_madd(r, pn); break;
                default: r = ge_madd_signed(ge_msub(ge_madd(r, pn), pn), pn, false); break;
            }
        }
    }
    ristretto_encode(out, r);
    return 0;
}
void xt_sha3_512(const u8 *d, u32 n, u8 out[64]) { sha3_512(out, d, n); }
void xt_sha3_256(const u8 *d, u32 n, u8 out[32]) { sha3_256(out, d, n); }
void xt_shake256(const u8 *d, u32 n, u8 *out, u32 outlen) { sponge s; s.init(136, 0x1f); s.absorb(d, n); s.squeeze(out, outlen); }
void xt_blake3(const u8 *d, u64 n, u8 out[32]) { blake3_hash(out, d, n); }
u64 xt_merlin_script(const char *proto_label, const u8 *ops, u64 ops_len, u8 *out, u64 out_cap) {
    transcript t; t.init(proto_label);
    u64 pos = 0, w = 0;
    while (pos < ops_len) {
        char op = (char)ops[pos++];
        const char *label = (const char *)(ops + pos);
        pos += strlen(label) + 1;
        u32 len; memcpy(&len, ops + pos, 4); pos += 4;
        if (op == 'A') { t.append_message(label, ops + pos, len); pos += len; }
        else { if (w + len > out_cap) return 0; t.challenge_bytes(label, out + w, len); w += len; }
    }
    return w;
}
// Replays the transcript template of a record on the CPU: gathers every block's 21 state words through the
// word descriptors, XORs them into a Keccak state, permutes, and squeezes the challenges where the template says.
// Returns the number of challenges written (32 bytes each), -1 malformed record, -2 unsupported shape.
// n_blocks_out / n_checks_failed_out (nullable) report the block count and the all-zero encodings found.
long long xt_transcript_template_challenges(const u8 *rec, u64 len, u8 *out32, u64 cap, u32 *n_blocks_out, u32 *zero_checks_out) {
    tx_record t;
    if (!record_parse(t, rec, len)) return -1;
    transcript_shape sh{t.type, t.n_commitments, (u32)t.transfer_off.size(), t.n_items, t.m(), t.lg()};
    if (!transcript_shape_supported(sh)) return -2;
    std::vector<u32> tmpl, seg;
    tmpl.assign(3, 0);                                  // some unrelated words in front: offsets must be honoured
    u32 base = build_transcript_template(tmpl, sh);
    transcript_segments(seg, t);
    const u32 *hdr = tmpl.data() + base;
    u32 n_blocks = hdr[TT_H_BLOCKS];
    const u32 *chal_after = hdr + TT_H_WORDS, *checks = chal_after + n_blocks, *desc = tmpl.data() + hdr[TT_H_DESC];
    if (n_blocks_out) *n_blocks_out = n_blocks;
    u32 zeros = 0;
    for (u32 c = 0; c < hdr[TT_H_CHECKS]; c++) {
        const u8 *p = rec + seg[checks[2 * c] >> 16] + (checks[2 * c] & 0xffffu);
        u8 acc = 0; for (int k = 0; k < 32; k++) acc |= p[k];
        if (!acc) zeros++;
    }
    if (zero_checks_out) *zero_checks_out = zeros;
    // the word gather reads up to 7 bytes before and after the fields it wants: give the record the slack the device buffer has
    std::vector<u8> padded(len + 16, 0);
    memcpy(padded.data() + 8, rec, len);
    keccak_state st; st.clear();
    long long n = 0;
    for (u32 b = 0; b < n_blocks; b++) {
        for (u32 w = 0; w < TT_WORDS; w++) st.lanes[w] ^= transcript_word(desc + ((u64)b * TT_WORDS + w) * TT_WDESC, padded.data() + 8, seg.data());
        st.permute();
        if (chal_after[b] != TT_NONE) {
            if ((u64)chal_after[b] < cap) { sc x = sc_from_wide_bytes(st.bytes); sc_tobytes(out32 + 32 * chal_after[b], x); }
            for (int j = 0; j < 64; j++) st.bytes[j] = 0;
            n++;
        }
    }
    return n;
}
}

// Host ingest of a batch (xhe_ingest.h) on the CPU: returns the seconds one ingest_batch call took and a few
// sizes of the tables it laid out, so that the layout can be checked and timed without a GPU.
#include <chrono>
#include "../../xelis-he_amd/csrc/xhe_ingest.h"
extern "C" double xt_ingest(const u8 *records, const u64 *offsets, u64 n, const xhe_ledger_view *view, u64 out_counts[8]) {
    auto t0 = std::chrono::steady_clock::now();
    host_batch hb;
    bool ok = ingest_batch(hb, records, offsets, n, *view);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out_counts[0] = ok; out_counts[1] = hb.n_points; out_counts[2] = hb.n_terms; out_counts[3] = hb.n_sigs;
    out_counts[4] = hb.n_ops; out_counts[5] = hb.tr_tiles.size(); out_counts[6] = hb.n_chal; out_counts[7] = hb.tr_plan.stream_words;
    return sec;
}

// The same ingest on the threads and in the memory a context keeps between uploads (ingest_workers, ingest_arena), twice in a
// row: the tables it lays out must not depend on either.  Returns 1 when both runs give the counts of the plain run.
extern "C" int xt_ingest_pooled_matches(const u8 *records, const u64 *offsets, u64 n, const xhe_ledger_view *view) {
    host_batch plain;
    const bool ok0 = ingest_batch(plain, records, offsets, n, *view);
    ingest_workers workers; workers.start(3);
    ingest_arena arena;
    for (int round = 0; round < 2; round++) {
        host_batch hb;
        const bool ok = ingest_batch(hb, records, offsets, n, *view, &workers, &arena);
        if (ok != ok0 || hb.n_points != plain.n_points || hb.n_terms != plain.n_terms || hb.n_sigs != plain.n_sigs || hb.n_ops != plain.n_ops || hb.n_chal != plain.n_chal ||
            hb.n_decs != plain.n_decs || hb.n_key_terms != plain.n_key_terms || hb.tr_list != plain.tr_list || hb.slot_of_used != plain.slot_of_used || hb.sig_keys != plain.sig_keys)
            return 0;
        if (hb.ex_txs.n != plain.ex_txs.n || (hb.ex_txs.n && memcmp(hb.ex_txs.p, plain.ex_txs.p, hb.ex_txs.n * sizeof(expand_tx)))) return 0;
        if (hb.ex_xfs.n != plain.ex_xfs.n || (hb.ex_xfs.n && memcmp(hb.ex_xfs.p, plain.ex_xfs.p, hb.ex_xfs.n * sizeof(expand_xf)))) return 0;
        if (hb.ex_cms.n != plain.ex_cms.n || (hb.ex_cms.n && memcmp(hb.ex_cms.p, plain.ex_cms.p, hb.ex_cms.n * sizeof(expand_cm)))) return 0;
    }
    return arena.blocks.size() <= 1 ? 1 : 2;       // (2: the arena did not settle on one block -- not an error of the tables)
}

// ingest_workers and ingest_arena on their own: ranges of a parallel pass cover [0, count) exactly once for every thread count up
// to the pool's size and beyond; the arena hands out aligned, disjoint memory and keeps one block after reset().
extern "C" int xt_pool_selftest() {
    ingest_workers workers; workers.start(5);
    for (unsigned threads = 1; threads <= 9; threads++)
        for (u64 count : {0ull, 1ull, 2047ull, 2048ull, 100003ull}) {
            std::vector<std::atomic<u32>> hits(count);
            for (auto &h : hits) h.store(0);
            ingest_parallel(count, threads, [&](u64 lo, u64 hi) { for (u64 i = lo; i < hi; i++) hits[i].fetch_add(1); }, &workers);
            for (auto &h : hits) if (h.load() != 1) return 0;
        }
    ingest_arena arena;
    for (int round = 0; round < 3; round++) {
        arena.reset();
        std::vector<std::pair<char *, size_t>> got;
        for (size_t bytes : {100ul, 5000000ul, 64ul, 9000000ul, 33ul}) {
            char *p = (char *)arena.allocate(bytes, 16);
            if (((size_t)p & 63) != 0) return 0;
            memset(p, round, bytes);
            for (auto &g : got) if (p < g.first + g.second && g.first < p + bytes) return 0;
            got.push_back({p, bytes});
        }
        if (round > 0 && arena.blocks.size() != 1) return 0;
    }
    return 1;
}
