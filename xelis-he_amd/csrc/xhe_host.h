// Host-side group operations for the transaction builder (prover): scalar multiplication, fixed-base
// tables for G and H, a small multiscalar multiplication, the Bulletproofs generator chain and a
// deterministic random generator.  The verification path never calls these: it runs in the kernels.
//
// Reference behaviour restated: /root/reference/src/elgamal.rs:16-24 (H), :97-134 (keys, handles),
// :237-276 (Pedersen commitments), /root/reference/src/proofs.rs:19-22 (BulletproofGens::new(64, 512),
// PedersenGens::default()).
#pragma once
#include <mutex>
#include <vector>
#include "xhe_ge.h"
#include "xhe_hash.h"

namespace xhe {
namespace host {

// SHAKE256(seed || counter) -> wide-reduced scalars; stands in for rand::rng()
struct drbg {
    u8 seed[32]; u64 ctr = 0;
    explicit drbg(const u8 s[32]) { for (int i = 0; i < 32; i++) seed[i] = s[i]; }
    void bytes(u8 *out, u32 n) {
        sponge sp; sp.init(136, 0x1f);
        sp.absorb(seed, 32);
        u8 c[8];
        for (int i = 0; i < 8; i++) c[i] = (u8)(ctr >> (8 * i));
        ctr++;
        sp.absorb(c, 8);
        sp.squeeze(out, n);
    }
    sc scalar() { u8 b[64]; bytes(b, 64); return sc_from_wide_bytes(b); }
    drbg fork() { u8 s[32]; bytes(s, 32); return drbg(s); }
};

inline void sc_bits_words(u32 w[8], const sc &k) { sc_to_words(w, k); }

// variable-base k*P, 4-bit fixed windows
inline ge scalarmul(const sc &k, const ge &p) {
    ge tbl[16];
    tbl[0] = ge_identity(); tbl[1] = p;
    for (int i = 2; i < 16; i++) tbl[i] = (i & 1) ? ge_add(tbl[i - 1], p) : ge_double(tbl[i / 2]);
    u32 w[8]; sc_to_words(w, k);
    ge r = ge_identity();
    for (int i = 63; i >= 0; i--) {
        if (i != 63) { r = ge_double(r); r = ge_double(r); r = ge_double(r); r = ge_double(r); }
        u32 nib = (w[i >> 3] >> ((i & 7) * 4)) & 15;
        if (nib) r = ge_add(r, tbl[nib]);
    }
    return r;
}

// fixed-base table: 64 windows x 15 multiples
struct fixed_base {
    std::vector<ge> tbl;      // [64][16]
    void build(const ge &p) {
        tbl.resize(64 * 16);
        ge base = p;
        for (int wdw = 0; wdw < 64; wdw++) {
            tbl[16 * wdw] = ge_identity();
            tbl[16 * wdw + 1] = base;
            for (int i = 2; i < 16; i++) tbl[16 * wdw + i] = ge_add(tbl[16 * wdw + i - 1], base);
            base = ge_double(ge_double(ge_double(ge_double(base))));
        }
    }
    ge mul(const sc &k) const {
        u32 w[8]; sc_to_words(w, k);
        ge r = ge_identity();
        for (int i = 0; i < 64; i++) {
            u32 nib = (w[i >> 3] >> ((i & 7) * 4)) & 15;
            if (nib) r = ge_add(r, tbl[16 * i + nib]);
        }
        return r;
    }
};

struct generators {
    ge B, H;                      // PC_GENS.B (ristretto basepoint) and B_blinding = H
    u8 B_bytes[32], H_bytes[32];
    fixed_base B_table, H_table;
    std::vector<ge> G_vec, H_vec; // BP_GENS, flat index 64*party + i, grown on demand
    std::mutex mu;
    generators() {
        const u8 bb[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                           0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
        for (int i = 0; i < 32; i++) B_bytes[i] = bb[i];
        ristretto_decode(B, B_bytes);
        u8 h[64]; sha3_512(h, B_bytes, 32);
        H = ristretto_from_uniform_bytes(h);
        ristretto_encode(H_bytes, H);
        B_table.build(B); H_table.build(H);
    }
    // generator i of share `party` = i-th 64-byte block of SHAKE256("GeneratorsChain" || 'G'|'H' || LE32(party))
    void ensure(size_t n_points) {
        std::lock_guard<std::mutex> lk(mu);
        size_t have = G_vec.size();
        if (have >= n_points) return;
        size_t parties = (n_points + 63) / 64;
        G_vec.resize(parties * 64); H_vec.resize(parties * 64);
        for (size_t j = have / 64; j < parties; j++)
            for (int which = 0; which < 2; which++) {
                sponge sp; sp.init(136, 0x1f);
                sp.absorb((const u8 *)"GeneratorsChain", 15);
                u8 label[5] = {(u8)(which ? 'H' : 'G'), (u8)j, (u8)(j >> 8), (u8)(j >> 16), (u8)(j >> 24)};
                sp.absorb(label, 5);
                for (int i = 0; i < 64; i++) {
                    u8 ub[64]; sp.squeeze(ub, 64);
                    (which ? H_vec : G_vec)[64 * j + i] = ristretto_from_uniform_bytes(ub);
                }
            }
    }
};
inline generators &gens() { static generators g; return g; }

inline ge mul_G(const sc &k) { return gens().B_table.mul(k); }
inline ge mul_H(const sc &k) { return gens().H_table.mul(k); }

// sum k_i * P_i on the host: shared doublings, 4-bit windows (the prover's MSMs are a few hundred terms)
inline ge msm(const std::vector<sc> &scalars, const std::vector<ge> &points) {
    size_t n = scalars.size();
    if (n == 0) return ge_identity();
    // Pippenger with c = 6 for larger sizes, interleaved windows otherwise
    if (n >= 32) {
        const int c = n < 300 ? 6 : 8;
        const int W = (253 + c - 1) / c;
        const int nb = 1 << c;
        std::vector<u32> words(8 * n);
        for (size_t i = 0; i < n; i++) sc_to_words(&words[8 * i], scalars[i]);
        ge total = ge_identity();
        std::vector<ge> buckets(nb);
        std::vector<u8> used(nb);
        for (int wdw = W - 1; wdw >= 0; wdw--) {
            for (int k = 0; k < c; k++) total = ge_double(total);
            std::fill(used.begin(), used.end(), 0);
            for (size_t i = 0; i < n; i++) {
                int bit = wdw * c;
                if (bit >= 256) continue;
                u32 word = bit >> 5, off = bit & 31;
                u64 two = (u64)words[8 * i + word] | ((word + 1 < 8) ? ((u64)words[8 * i + word + 1] << 32) : 0);
                u32 d = (u32)(two >> off) & (nb - 1);
                if (!d) continue;
                if (used[d]) buckets[d] = ge_add(buckets[d], points[i]); else { buckets[d] = points[i]; used[d] = 1; }
            }
            ge run = ge_identity(), sum = ge_identity();
            bool any = false;
            for (int b = nb - 1; b >= 1; b--) {
                if (used[b]) { run = any ? ge_add(run, buckets[b]) : buckets[b]; any = true; }
                if (any) sum = ge_add(sum, run);
            }
            total = ge_add(total, sum);
        }
        return total;
    }
    std::vector<ge> tbl(16 * n);
    std::vector<u32> words(8 * n);
    for (size_t i = 0; i < n; i++) {
        sc_to_words(&words[8 * i], scalars[i]);
        tbl[16 * i] = ge_identity(); tbl[16 * i + 1] = points[i];
        for (int j = 2; j < 16; j++) tbl[16 * i + j] = (j & 1) ? ge_add(tbl[16 * i + j - 1], points[i]) : ge_double(tbl[16 * i + j / 2]);
    }
    ge r = ge_identity();
    for (int d = 63; d >= 0; d--) {
        if (d != 63) { r = ge_double(r); r = ge_double(r); r = ge_double(r); r = ge_double(r); }
        for (size_t i = 0; i < n; i++) {
            u32 nib = (words[8 * i + (d >> 3)] >> ((d & 7) * 4)) & 15;
            if (nib) r = ge_add(r, tbl[16 * i + nib]);
        }
    }
    return r;
}

// a*P + b*Q
inline ge mul2(const sc &a, const ge &p, const sc &b, const ge &q) {
    std::vector<sc> s = {a, b}; std::vector<ge> v = {p, q};
    return msm(s, v);
}

inline void encode(u8 out[32], const ge &p) { ristretto_encode(out, p); }

}  // namespace host
}  // namespace xhe
