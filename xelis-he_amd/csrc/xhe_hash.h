// Keccak-f[1600], SHA3-512/256, SHAKE256, the STROBE-128 duplex Merlin is built on, the Merlin
// transcript with the reference's ProtocolTranscript extension, and BLAKE3 (host + device).
//
// Reference behaviour restated: /root/reference/src/transcript.rs:37-111 (labels, encodings,
// challenge_scalar = 64-byte PRF output reduced mod l, validate_and_append_point rejecting only the
// all-zero encoding), /root/reference/src/tx/verify.rs:146-158 (transcript header),
// /root/reference/src/elgamal.rs:53-65 (SHA3-512 over pk || msg || R), /root/reference/src/tx/verify.rs:267
// (BLAKE3 of the pre-image).  Merlin framing: meta-AD(label) || meta-AD(LE32 len, more) || AD(data);
// challenges use the same framing followed by PRF.
#pragma once
#include "xhe_sc.h"

namespace xhe {

XHE_HD u64 rotl64(u64 x, int n) { return (x << n) | (x >> (64 - n)); }

// 24 rounds on 25 lanes held in registers (every index below is a compile-time constant)
XHE_HD_NOINLINE inline void keccak_f1600(u64 s[25]) {
    const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    u64 a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = s[i];
    for (int round = 0; round < 24; round++) {
        u64 c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20];
        u64 c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21];
        u64 c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22];
        u64 c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23];
        u64 c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];
        u64 d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1), d3 = c2 ^ rotl64(c4, 1), d4 = c3 ^ rotl64(c0, 1);
#pragma unroll
        for (int y = 0; y < 25; y += 5) { a[y] ^= d0; a[y + 1] ^= d1; a[y + 2] ^= d2; a[y + 3] ^= d3; a[y + 4] ^= d4; }
        // rho + pi
        u64 b[25];
        b[0] = a[0];
        b[10] = rotl64(a[1], 1);   b[20] = rotl64(a[2], 62);  b[5] = rotl64(a[3], 28);   b[15] = rotl64(a[4], 27);
        b[16] = rotl64(a[5], 36);  b[1] = rotl64(a[6], 44);   b[11] = rotl64(a[7], 6);   b[21] = rotl64(a[8], 55);  b[6] = rotl64(a[9], 20);
        b[7] = rotl64(a[10], 3);   b[17] = rotl64(a[11], 10); b[2] = rotl64(a[12], 43);  b[12] = rotl64(a[13], 25); b[22] = rotl64(a[14], 39);
        b[23] = rotl64(a[15], 41); b[8] = rotl64(a[16], 45);  b[18] = rotl64(a[17], 15); b[3] = rotl64(a[18], 21);  b[13] = rotl64(a[19], 8);
        b[14] = rotl64(a[20], 18); b[24] = rotl64(a[21], 2);  b[9] = rotl64(a[22], 61);  b[19] = rotl64(a[23], 56); b[4] = rotl64(a[24], 14);
        // chi
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            a[y] = b[y] ^ (~b[y + 1] & b[y + 2]);
            a[y + 1] = b[y + 1] ^ (~b[y + 2] & b[y + 3]);
            a[y + 2] = b[y + 2] ^ (~b[y + 3] & b[y + 4]);
            a[y + 3] = b[y + 3] ^ (~b[y + 4] & b[y]);
            a[y + 4] = b[y + 4] ^ (~b[y] & b[y + 1]);
        }
        a[0] ^= RC[round];
    }
#pragma unroll
    for (int i = 0; i < 25; i++) s[i] = a[i];
}

// 200-byte duplex state addressed by byte (little-endian lanes)
struct keccak_state {
    union { u64 lanes[25]; u8 bytes[200]; };
    XHE_HD void clear() { for (int i = 0; i < 25; i++) lanes[i] = 0; }
    XHE_HD void permute() { keccak_f1600(lanes); }
};

// FIPS 202 sponge
struct sponge {
    keccak_state st; u32 rate, pos; u8 suffix; bool squeezing;
    XHE_HD void init(u32 rate_bytes, u8 domain_suffix) { st.clear(); rate = rate_bytes; pos = 0; suffix = domain_suffix; squeezing = false; }
    XHE_HD void absorb(const u8 *d, u32 n) {
        for (u32 i = 0; i < n; i++) { st.bytes[pos++] ^= d[i]; if (pos == rate) { st.permute(); pos = 0; } }
    }
    XHE_HD void finish() { st.bytes[pos] ^= suffix; st.bytes[rate - 1] ^= 0x80; st.permute(); pos = 0; squeezing = true; }
    XHE_HD void squeeze(u8 *out, u32 n) {
        if (!squeezing) finish();
        for (u32 i = 0; i < n; i++) { if (pos == rate) { st.permute(); pos = 0; } out[i] = st.bytes[pos++]; }
    }
};
XHE_HD void sha3_512(u8 out[64], const u8 *d, u32 n) { sponge s; s.init(72, 0x06); s.absorb(d, n); s.squeeze(out, 64); }
XHE_HD void sha3_256(u8 out[32], const u8 *d, u32 n) { sponge s; s.init(136, 0x06); s.absorb(d, n); s.squeeze(out, 32); }

// STROBE-128/1600 subset used by Merlin: meta-AD, AD, PRF
struct strobe128 {
    keccak_state st; u8 pos, pos_begin;
    static constexpr u32 R = 166;
    XHE_HD void run_f() {
        st.bytes[pos] ^= pos_begin; st.bytes[pos + 1] ^= 0x04; st.bytes[R + 1] ^= 0x80;
        st.permute(); pos = 0; pos_begin = 0;
    }
    XHE_HD void absorb(const u8 *d, u32 n) { for (u32 i = 0; i < n; i++) { st.bytes[pos++] ^= d[i]; if (pos == R) run_f(); } }
    XHE_HD void squeeze(u8 *d, u32 n) { for (u32 i = 0; i < n; i++) { d[i] = st.bytes[pos]; st.bytes[pos++] = 0; if (pos == R) run_f(); } }
    XHE_HD void begin_op(u8 flags) {
        u8 hdr[2] = {pos_begin, flags};
        pos_begin = pos + 1;
        absorb(hdr, 2);
        if ((flags & (4 | 32)) && pos != 0) run_f();     // C or K flag forces a permutation
    }
    XHE_HD void init(const u8 *label, u32 n) {
        st.clear();
        const u8 hdr[18] = {1, 168, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
        for (int i = 0; i < 18; i++) st.bytes[i] = hdr[i];
        st.permute(); pos = 0; pos_begin = 0;
        begin_op(2 | 16); absorb(label, n);
    }
    XHE_HD void meta_ad(const u8 *d, u32 n, bool more) { if (!more) begin_op(2 | 16); absorb(d, n); }
    XHE_HD void ad(const u8 *d, u32 n) { begin_op(2); absorb(d, n); }
    XHE_HD void prf(u8 *d, u32 n) { begin_op(1 | 2 | 4); squeeze(d, n); }
};

XHE_HD u32 xhe_strlen(const char *s) { u32 n = 0; while (s[n]) n++; return n; }

struct transcript {
    strobe128 s;
    XHE_HD void init(const char *label) {
        const u8 proto[11] = {'M', 'e', 'r', 'l', 'i', 'n', ' ', 'v', '1', '.', '0'};
        s.init(proto, 11);
        append_message("dom-sep", (const u8 *)label, xhe_strlen(label));
    }
    XHE_HD void frame(const char *label, u32 n) {
        u8 len[4] = {(u8)n, (u8)(n >> 8), (u8)(n >> 16), (u8)(n >> 24)};
        s.meta_ad((const u8 *)label, xhe_strlen(label), false);
        s.meta_ad(len, 4, true);
    }
    XHE_HD void append_message(const char *label, const u8 *msg, u32 n) { frame(label, n); s.ad(msg, n); }
    XHE_HD void append_u64(const char *label, u64 x) {
        u8 b[8];
        for (int i = 0; i < 8; i++) b[i] = (u8)(x >> (8 * i));
        append_message(label, b, 8);
    }
    XHE_HD void challenge_bytes(const char *label, u8 *out, u32 n) { frame(label, n); s.prf(out, n); }
    XHE_HD sc challenge_scalar(const char *label) { u8 b[64]; challenge_bytes(label, b, 64); return sc_from_wide_bytes(b); }
    XHE_HD void append_scalar(const char *label, const sc &x) { u8 b[32]; sc_tobytes(b, x); append_message(label, b, 32); }
    XHE_HD void append_point(const char *label, const u8 p[32]) { append_message(label, p, 32); }
    XHE_HD bool validate_and_append_point(const char *label, const u8 p[32]) {
        u8 acc = 0;
        for (int i = 0; i < 32; i++) acc |= p[i];
        if (!acc) return false;
        append_message(label, p, 32);
        return true;
    }
};

// ---- BLAKE3 (hash mode, 32-byte output) ------------------------------------------------------------
XHE_HD u32 rotr32(u32 x, int n) { return (x >> n) | (x << (32 - n)); }
XHE_HD void b3_g(u32 s[16], int a, int b, int c, int d, u32 mx, u32 my) {
    s[a] = s[a] + s[b] + mx; s[d] = rotr32(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d];      s[b] = rotr32(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my; s[d] = rotr32(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d];      s[b] = rotr32(s[b] ^ s[c], 7);
}
XHE_HD_NOINLINE inline void b3_compress(u32 out[8], const u32 cv[8], const u8 block[64], u64 counter, u32 block_len, u32 flags) {
    const u32 IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
    const u8 PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
    u32 s[16], m[16];
    for (int i = 0; i < 8; i++) { s[i] = cv[i]; }
    for (int i = 0; i < 4; i++) s[8 + i] = IV[i];
    s[12] = (u32)counter; s[13] = (u32)(counter >> 32); s[14] = block_len; s[15] = flags;
    for (int i = 0; i < 16; i++) m[i] = (u32)block[4 * i] | ((u32)block[4 * i + 1] << 8) | ((u32)block[4 * i + 2] << 16) | ((u32)block[4 * i + 3] << 24);
    for (int r = 0; r < 7; r++) {
        b3_g(s, 0, 4, 8, 12, m[0], m[1]); b3_g(s, 1, 5, 9, 13, m[2], m[3]); b3_g(s, 2, 6, 10, 14, m[4], m[5]); b3_g(s, 3, 7, 11, 15, m[6], m[7]);
        b3_g(s, 0, 5, 10, 15, m[8], m[9]); b3_g(s, 1, 6, 11, 12, m[10], m[11]); b3_g(s, 2, 7, 8, 13, m[12], m[13]); b3_g(s, 3, 4, 9, 14, m[14], m[15]);
        u32 p[16];
        for (int i = 0; i < 16; i++) p[i] = m[PERM[i]];
        for (int i = 0; i < 16; i++) m[i] = p[i];
    }
    for (int i = 0; i < 8; i++) out[i] = s[i] ^ s[i + 8];
}
// Iterative tree hash with a chaining-value stack (BLAKE3 spec section 5.1.2); any input length.
XHE_HD_NOINLINE inline void blake3_hash(u8 out[32], const u8 *data, u64 len) {
    const u32 IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
    u32 stack[32][8]; int sp = 0;
    u64 n_chunks = len == 0 ? 1 : (len + 1023) / 1024;
    u32 cv[8];
    for (u64 ci = 0; ci < n_chunks; ci++) {
        u64 off = ci * 1024, clen = (len - off < 1024) ? len - off : 1024;
        bool last_chunk = (ci == n_chunks - 1);
        for (int i = 0; i < 8; i++) cv[i] = IV[i];
        u64 n_blocks = clen == 0 ? 1 : (clen + 63) / 64;
        for (u64 b = 0; b < n_blocks; b++) {
            u8 blk[64];
            u64 boff = b * 64, bl = (clen - boff < 64) ? clen - boff : 64;
            for (u64 i = 0; i < 64; i++) blk[i] = (i < bl) ? data[off + boff + i] : 0;
            u32 flags = (b == 0 ? 1u : 0u) | (b == n_blocks - 1 ? 2u : 0u);
            if (b == n_blocks - 1 && last_chunk && n_chunks == 1) flags |= 8u;           // single-chunk root
            u32 o[8]; b3_compress(o, cv, blk, ci, (u32)bl, flags);
            for (int i = 0; i < 8; i++) cv[i] = o[i];
        }
        if (n_chunks == 1) break;
        if (!last_chunk) {
            // merge completed subtrees: one merge per trailing zero bit of the chunk count so far
            u64 total = ci + 1;
            while ((total & 1) == 0) {
                u8 blk[64];
                for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) { blk[4 * i + k] = (u8)(stack[sp - 1][i] >> (8 * k)); blk[32 + 4 * i + k] = (u8)(cv[i] >> (8 * k)); }
                sp--;
                u32 o[8]; b3_compress(o, IV, blk, 0, 64, 4u);
                for (int i = 0; i < 8; i++) cv[i] = o[i];
                total >>= 1;
            }
            for (int i = 0; i < 8; i++) stack[sp][i] = cv[i];
            sp++;
        } else {
            // final chunk: fold the stack from the top, the last merge is the root
            while (sp > 0) {
                u8 blk[64];
                for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) { blk[4 * i + k] = (u8)(stack[sp - 1][i] >> (8 * k)); blk[32 + 4 * i + k] = (u8)(cv[i] >> (8 * k)); }
                sp--;
                u32 o[8]; b3_compress(o, IV, blk, 0, 64, 4u | (sp == 0 ? 8u : 0u));
                for (int i = 0; i < 8; i++) cv[i] = o[i];
            }
        }
    }
    for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) out[4 * i + k] = (u8)(cv[i] >> (8 * k));
}

}  // namespace xhe
