// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under xelis-he_amd/ may include, link or call this.
//
// Keccak-f[1600] (FIPS 202), SHA3-256/512, SHAKE256, STROBE-128 as Merlin v1.0 uses it, the Merlin
// transcript, and BLAKE3 (hash mode).  The reference reaches these through the crates merlin 4.0.0
// (xelis fork), sha3 0.10.8 and blake3 1.8.2 (/root/reference/Cargo.lock:530-539), none vendored;
// call sites: /root/reference/src/transcript.rs:37-111, /root/reference/src/tx/verify.rs:146-158,
// /root/reference/src/elgamal.rs:53-65 (SHA3-512), /root/reference/src/tx/verify.rs:267 (BLAKE3).
// Pinned by: hashlib SHA3/SHAKE digests, the Merlin "test protocol" vectors, BLAKE3 empty/"abc".
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "xo_sc.h"

namespace xo {

static inline u64 rotl64(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

static inline void keccak_f1600(u64 st[25]) {
    static const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int round = 0; round < 24; round++) {
        u64 C[5], D[5], B[25];
        for (int x = 0; x < 5; x++) C[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
        for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rotl64(C[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) st[i] ^= D[i % 5];
        // rho + pi: B[y][2x+3y] = rot(A[x][y])
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++)
                B[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(st[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++)
                st[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
        st[0] ^= RC[round];
    }
}

// byte-addressed sponge (little-endian lanes)
struct sponge {
    u64 st[25];
    unsigned rate, pos;
    u8 dsuffix;
    bool squeezing;
    void init(unsigned rate_bytes, u8 suffix) { memset(st, 0, sizeof st); rate = rate_bytes; pos = 0; dsuffix = suffix; squeezing = false; }
    void xor_byte(unsigned i, u8 b) { st[i >> 3] ^= (u64)b << (8 * (i & 7)); }
    u8 get_byte(unsigned i) const { return (u8)(st[i >> 3] >> (8 * (i & 7))); }
    void absorb(const u8 *data, size_t len) {
        for (size_t i = 0; i < len; i++) {
            xor_byte(pos++, data[i]);
            if (pos == rate) { keccak_f1600(st); pos = 0; }
        }
    }
    void finish() {
        xor_byte(pos, dsuffix);
        xor_byte(rate - 1, 0x80);
        keccak_f1600(st);
        pos = 0; squeezing = true;
    }
    void squeeze(u8 *out, size_t len) {
        if (!squeezing) finish();
        for (size_t i = 0; i < len; i++) {
            if (pos == rate) { keccak_f1600(st); pos = 0; }
            out[i] = get_byte(pos++);
        }
    }
};

static inline void sha3_512(u8 out[64], const u8 *data, size_t len) { sponge s; s.init(72, 0x06); s.absorb(data, len); s.squeeze(out, 64); }
static inline void sha3_256(u8 out[32], const u8 *data, size_t len) { sponge s; s.init(136, 0x06); s.absorb(data, len); s.squeeze(out, 32); }
static inline sponge shake256_init() { sponge s; s.init(136, 0x1f); return s; }

// ---------------------------------------------------------------------------------------------
// STROBE-128 subset used by Merlin (merlin/src/strobe.rs), and the Merlin transcript.
struct strobe128 {
    u8 st[200];
    u8 pos, pos_begin, cur_flags;
    enum { R = 166, FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };

    void permute() {
        u64 lanes[25];
        memcpy(lanes, st, 200);           // little-endian host
        keccak_f1600(lanes);
        memcpy(st, lanes, 200);
    }
    void run_f() {
        st[pos] ^= pos_begin;
        st[pos + 1] ^= 0x04;
        st[R + 1] ^= 0x80;
        permute();
        pos = 0; pos_begin = 0;
    }
    void absorb(const u8 *d, size_t n) {
        for (size_t i = 0; i < n; i++) { st[pos] ^= d[i]; pos++; if (pos == R) run_f(); }
    }
    void squeeze(u8 *d, size_t n) {
        for (size_t i = 0; i < n; i++) { d[i] = st[pos]; st[pos] = 0; pos++; if (pos == R) run_f(); }
    }
    void begin_op(u8 flags, bool more) {
        if (more) return;                 // continuing the same operation
        u8 old_begin = pos_begin;
        pos_begin = pos + 1;
        cur_flags = flags;
        u8 hdr[2] = {old_begin, flags};
        absorb(hdr, 2);
        bool force_f = (flags & (FLAG_C | FLAG_K)) != 0;
        if (force_f && pos != 0) run_f();
    }
    void init(const u8 *label, size_t n) {
        memset(st, 0, 200);
        const u8 hdr[6] = {1, R + 2, 1, 0, 1, 96};
        memcpy(st, hdr, 6);
        memcpy(st + 6, "STROBEv1.0.2", 12);
        permute();
        pos = 0; pos_begin = 0; cur_flags = 0;
        meta_ad(label, n, false);
    }
    void meta_ad(const u8 *d, size_t n, bool more) { begin_op(FLAG_M | FLAG_A, more); absorb(d, n); }
    void ad(const u8 *d, size_t n, bool more) { begin_op(FLAG_A, more); absorb(d, n); }
    void prf(u8 *d, size_t n, bool more) { begin_op(FLAG_I | FLAG_A | FLAG_C, more); squeeze(d, n); }
};

struct transcript {
    strobe128 s;
    void init(const char *label) {
        s.init((const u8 *)"Merlin v1.0", 11);
        append_message("dom-sep", (const u8 *)label, strlen(label));
    }
    void append_message(const char *label, const u8 *msg, size_t n) {
        u8 len[4] = {(u8)n, (u8)(n >> 8), (u8)(n >> 16), (u8)(n >> 24)};
        s.meta_ad((const u8 *)label, strlen(label), false);
        s.meta_ad(len, 4, true);
        s.ad(msg, n, false);
    }
    void append_u64(const char *label, u64 x) {
        u8 b[8]; for (int i = 0; i < 8; i++) b[i] = (u8)(x >> (8 * i));
        append_message(label, b, 8);
    }
    void challenge_bytes(const char *label, u8 *out, size_t n) {
        u8 len[4] = {(u8)n, (u8)(n >> 8), (u8)(n >> 16), (u8)(n >> 24)};
        s.meta_ad((const u8 *)label, strlen(label), false);
        s.meta_ad(len, 4, true);
        s.prf(out, n, false);
    }
    // transcript.rs:46-51
    sc challenge_scalar(const char *label) {
        u8 buf[64]; challenge_bytes(label, buf, 64);
        return sc_from_bytes_mod_order_wide(buf);
    }
    void append_scalar(const char *label, const sc &x) { u8 b[32]; sc_tobytes(b, x); append_message(label, b, 32); }
    void append_point(const char *label, const u8 p[32]) { append_message(label, p, 32); }
    // transcript.rs:73-84: only the all-zero encoding is rejected
    bool validate_and_append_point(const char *label, const u8 p[32]) {
        u8 acc = 0; for (int i = 0; i < 32; i++) acc |= p[i];
        if (!acc) return false;
        append_message(label, p, 32);
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// BLAKE3, default hash mode, 32-byte output (BLAKE3 spec sections 2.1-2.6).
namespace b3 {
static const u32 IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const int PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };
static inline u32 rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }
static inline void g(u32 s[16], int a, int b, int c, int d, u32 mx, u32 my) {
    s[a] = s[a] + s[b] + mx; s[d] = rotr(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d];      s[b] = rotr(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my; s[d] = rotr(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d];      s[b] = rotr(s[b] ^ s[c], 7);
}
static inline void compress(u32 out[16], const u32 cv[8], const u32 block[16], u64 counter, u32 block_len, u32 flags) {
    u32 s[16] = {cv[0], cv[1], cv[2], cv[3], cv[4], cv[5], cv[6], cv[7], IV[0], IV[1], IV[2], IV[3],
                 (u32)counter, (u32)(counter >> 32), block_len, flags};
    u32 m[16]; memcpy(m, block, 64);
    for (int r = 0; r < 7; r++) {
        g(s, 0, 4, 8, 12, m[0], m[1]); g(s, 1, 5, 9, 13, m[2], m[3]); g(s, 2, 6, 10, 14, m[4], m[5]); g(s, 3, 7, 11, 15, m[6], m[7]);
        g(s, 0, 5, 10, 15, m[8], m[9]); g(s, 1, 6, 11, 12, m[10], m[11]); g(s, 2, 7, 8, 13, m[12], m[13]); g(s, 3, 4, 9, 14, m[14], m[15]);
        u32 p[16]; for (int i = 0; i < 16; i++) p[i] = m[PERM[i]];
        memcpy(m, p, 64);
    }
    for (int i = 0; i < 8; i++) { out[i] = s[i] ^ s[i + 8]; out[i + 8] = s[i + 8] ^ cv[i]; }
}
// chaining value of one chunk (<= 1024 bytes); is_root => final output of a single-chunk input
static inline void chunk_cv(u32 cv_out[8], const u8 *data, size_t len, u64 chunk_index, bool is_root) {
    u32 cv[8]; memcpy(cv, IV, 32);
    size_t nblocks = len == 0 ? 1 : (len + 63) / 64;
    for (size_t b = 0; b < nblocks; b++) {
        u8 blk[64]; memset(blk, 0, 64);
        size_t off = b * 64, bl = len - off < 64 ? len - off : 64;
        memcpy(blk, data + off, bl);
        u32 words[16]; memcpy(words, blk, 64);
        u32 flags = 0;
        if (b == 0) flags |= CHUNK_START;
        if (b == nblocks - 1) { flags |= CHUNK_END; if (is_root) flags |= ROOT; }
        u32 out[16]; compress(out, cv, words, chunk_index, (u32)bl, flags);
        memcpy(cv, out, 32);
    }
    memcpy(cv_out, cv, 32);
}
static inline void parent_cv(u32 out_cv[8], const u32 l[8], const u32 r[8], bool is_root) {
    u32 block[16]; memcpy(block, l, 32); memcpy(block + 8, r, 32);
    u32 out[16]; compress(out, IV, block, 0, 64, PARENT | (is_root ? ROOT : 0));
    memcpy(out_cv, out, 32);
}
// recursive tree: left subtree takes the largest power-of-two number of chunks strictly less than the total
static inline void subtree(u32 cv[8], const u8 *data, size_t len, u64 chunk0, bool is_root) {
    if (len <= 1024) { chunk_cv(cv, data, len, chunk0, is_root); return; }
    size_t chunks = (len + 1023) / 1024;
    size_t left_chunks = 1; while (left_chunks * 2 < chunks) left_chunks *= 2;
    u32 l[8], r[8];
    subtree(l, data, left_chunks * 1024, chunk0, false);
    subtree(r, data + left_chunks * 1024, len - left_chunks * 1024, chunk0 + left_chunks, false);
    parent_cv(cv, l, r, is_root);
}
}  // namespace b3
static inline void blake3_hash(u8 out[32], const u8 *data, size_t len) {
    u32 cv[8]; b3::subtree(cv, data, len, 0, true);
    memcpy(out, cv, 32);
}

}  // namespace xo
