// Transcript templates: the Merlin/STROBE-128 transcript of a transaction, unrolled on the host into the
// sequence of 168-byte blocks that get XORed into the Keccak state before each permutation.
//
// Which byte of which block comes from where depends only on the transaction's shape (type, numbers of
// source commitments / transfers / multisig signers, range-proof size): labels, lengths, STROBE operation
// headers and padding are constants, everything else is a byte of the record at a fixed offset inside one
// of the transaction's segments (pre-image, range proof, commitment k, transfer k, transfer proof k).  A
// template lists, per block, the 21 eight-byte state words as word descriptors plus the index of the
// Fiat-Shamir challenge that is squeezed after the block's permutation (if any), and the encodings that must
// not be all-zero (validate_and_append_point, /root/reference/src/transcript.rs:70-85).  The device kernel
// then only gathers and permutes; it never interprets STROBE.
//
// A word descriptor is eight 32-bit words: the constant bytes of the word (data positions zero) and up to two
// "runs" of record bytes.  STROBE puts at least nine framing bytes between two data fields, so one state word
// never holds bytes of more than two fields.  A run is fetched with one unaligned 8-byte load placed so that
// the wanted source bytes fall on their positions in the word, then masked: {seg << 16 | (start + 8), control}
// with start = source offset of the load relative to the segment (may be down to -7) and control =
// first byte | length << 4 | reversed << 8 | valid << 9 (reversed: the big-endian integers of the record are
// absorbed little-endian, so the eight loaded bytes are swapped end for end first).
//
// Mirrors strobe128 / transcript of xhe_hash.h operation by operation (same begin_op / run_f rules), and
// the append order of k_transcript_serial (/root/reference/src/tx/verify.rs:146-158,315-326,378-399,420-424
// and the range-proof transcript of the bulletproofs dependency).
#pragma once
#include <assert.h>
#include <string.h>
#include <vector>
#include "../../include/xhe.h"
#include "xhe_verify_types.h"

namespace xhe {

static constexpr u32 TT_CONST = 0x80000000u;          // descriptor: constant byte in the low 8 bits
static constexpr u32 TT_NONE = 0xffffffffu;
static constexpr u32 TT_BLOCK_BYTES = 168;            // rate 166 + the two padding bytes
static constexpr u32 TT_SEG_PRE = 0, TT_SEG_RP = 1;   // then commitment k = 2 + k, transfer k = 2 + A + 2k, its proof = 3 + A + 2k
static constexpr u32 TT_MAX_SEGS = 32768, TT_MAX_OFF = 65536;
// header words of a template inside the device array
enum : u32 { TT_H_BLOCKS = 0, TT_H_CHALS = 1, TT_H_CHECKS = 2, TT_H_DESC = 3, TT_H_WORDS = 4 };
// layout: header[4] | chal_after[n_blocks] | checks[n_checks][2] (descriptor of the first byte, error key; key 0 = range-proof flag) |
// padding | wdesc[n_blocks][21][8], the word descriptors starting at word header[TT_H_DESC] of the array, a multiple of 8
static constexpr u32 TT_WORDS = 21, TT_WDESC = 8;      // state words per block, 32-bit words per word descriptor
static constexpr u32 TT_RUN_VALID = 1u << 9, TT_RUN_REVERSED = 1u << 8;

// the eight bytes of a state word from its descriptor (host replay and device gather share this)
XHE_HD u64 transcript_word(const u32 d[8], const u8 *rec, const u32 *seg) {
    u64 word = (u64)d[0] | ((u64)d[1] << 32);
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const u32 src = d[2 + 2 * r], ctl = d[3 + 2 * r];
        if (!(ctl & TT_RUN_VALID)) continue;
        const u8 *p = rec + seg[src >> 16] + (src & 0xffffu) - 8;
        u64 v;
        __builtin_memcpy(&v, p, 8);
        if (ctl & TT_RUN_REVERSED) v = __builtin_bswap64(v);
        const u32 first = ctl & 15, len = (ctl >> 4) & 15;
        const u64 mask = (len >= 8 ? ~0ull : ((1ull << (8 * len)) - 1)) << (8 * first);
        word |= v & mask;
    }
    return word;
}

struct transcript_shape {
    u8 type; u32 n_commitments, n_transfers, n_items, m, lg;
    bool operator==(const transcript_shape &o) const {
        return type == o.type && n_commitments == o.n_commitments && n_transfers == o.n_transfers && n_items == o.n_items && m == o.m && lg == o.lg;
    }
};
struct transcript_shape_hash {
    size_t operator()(const transcript_shape &s) const {
        u64 h = s.type; h = h * 1000003u + s.n_commitments; h = h * 1000003u + s.n_transfers; h = h * 1000003u + s.n_items; h = h * 1000003u + s.m; h = h * 1000003u + s.lg;
        return (size_t)h;
    }
};

// shapes the template path handles; others go through the serial kernel
static inline bool transcript_shape_supported(const transcript_shape &s) {
    u64 segs = 2ull + s.n_commitments + 2ull * s.n_transfers;
    return segs <= 4096 && s.n_items <= 255 && s.lg >= 6 && s.lg <= 15 && (64ull * s.m == (1ull << s.lg));
}

struct strobe_recorder {
    static constexpr u32 R = 166;
    std::vector<u32> desc, chal_after;          // desc: one entry per stream byte (constant, or seg << 16 | offset)
    std::vector<u8> rev;                         // per stream byte: 1 when it belongs to a field absorbed end for end
    u32 cur[TT_BLOCK_BYTES]; u8 cur_rev[TT_BLOCK_BYTES];
    u32 pos = 0, pos_begin = 0, n_chal = 0;
    void clear_cur() { for (u32 j = 0; j < TT_BLOCK_BYTES; j++) { cur[j] = TT_CONST; cur_rev[j] = 0; } }
    void emit() { desc.insert(desc.end(), cur, cur + TT_BLOCK_BYTES); rev.insert(rev.end(), cur_rev, cur_rev + TT_BLOCK_BYTES); chal_after.push_back(TT_NONE); clear_cur(); }
    void xor_const(u32 j, u8 v) { assert(cur[j] & TT_CONST); cur[j] ^= v; }
    void run_f() { xor_const(pos, (u8)pos_begin); xor_const(pos + 1, 0x04); xor_const(R + 1, 0x80); emit(); pos = 0; pos_begin = 0; }
    void absorb_const(const u8 *d, u32 n) { for (u32 i = 0; i < n; i++) { xor_const(pos++, d[i]); if (pos == R) run_f(); } }
    void absorb_data(u32 seg, u32 off, u32 n, bool reversed = false) {
        for (u32 i = 0; i < n; i++) {
            u32 o = reversed ? off + n - 1 - i : off + i;
            assert(seg < TT_MAX_SEGS && o < TT_MAX_OFF && cur[pos] == TT_CONST);
            cur_rev[pos] = reversed ? 1 : 0;
            cur[pos++] = (seg << 16) | o;
            if (pos == R) run_f();
        }
    }
    void begin_op(u8 flags) {
        u8 hdr[2] = {(u8)pos_begin, flags};
        pos_begin = pos + 1;
        absorb_const(hdr, 2);
        if ((flags & (4 | 32)) && pos != 0) run_f();
    }
    void init(const u8 *label, u32 n) {
        clear_cur();
        const u8 hdr[18] = {1, 168, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
        for (int i = 0; i < 18; i++) cur[i] = TT_CONST | hdr[i];
        emit();                                        // the initial permutation takes no padding
        pos = 0; pos_begin = 0;
        begin_op(2 | 16); absorb_const(label, n);
    }
    void frame(const char *label, u32 n) {
        u8 len[4] = {(u8)n, (u8)(n >> 8), (u8)(n >> 16), (u8)(n >> 24)};
        begin_op(2 | 16); absorb_const((const u8 *)label, (u32)strlen(label));
        absorb_const(len, 4);
    }
    void append_const(const char *label, const u8 *msg, u32 n) { frame(label, n); begin_op(2); absorb_const(msg, n); }
    void append_data(const char *label, u32 seg, u32 off, u32 n, bool reversed = false) { frame(label, n); begin_op(2); absorb_data(seg, off, n, reversed); }
    void append_u64(const char *label, u64 x) { u8 b[8]; for (int i = 0; i < 8; i++) b[i] = (u8)(x >> (8 * i)); append_const(label, b, 8); }
    void challenge(const char *label) {
        frame(label, 64);
        begin_op(1 | 2 | 4);
        assert(pos == 0 && !chal_after.empty() && chal_after.back() == TT_NONE);
        chal_after.back() = n_chal++;
        pos = 64;                                      // the 64 squeezed bytes are zeroed by the kernel
    }
};

// words appended to `out`; returns the offset of the template's header inside `out`
static inline u32 build_transcript_template(std::vector<u32> &out, const transcript_shape &s) {
    strobe_recorder r;
    std::vector<u32> checks;
    auto check = [&](u32 seg, u32 off, u32 key) { checks.push_back((seg << 16) | off); checks.push_back(key); };
    const u32 A = s.n_commitments, T = s.n_transfers;
    const u8 proto[11] = {'M', 'e', 'r', 'l', 'i', 'n', ' ', 'v', '1', '.', '0'};
    r.init(proto, 11);
    r.append_const("dom-sep", (const u8 *)"transaction-proof", 17);
    // header: version, source, fee, nonce (integers little-endian in the transcript, big-endian in the record)
    { r.frame("version", 8); r.begin_op(2); r.absorb_data(TT_SEG_PRE, 0, 1); const u8 z[7] = {0}; r.absorb_const(z, 7); }
    r.append_data("source_pubkey", TT_SEG_PRE, 1, 32);
    r.append_data("fee", TT_SEG_PRE, 33, 8, true);
    r.append_data("nonce", TT_SEG_PRE, 41, 8, true);
    for (u32 k = 0; k < A; k++) {
        u32 seg = 2 + k;
        r.append_const("dom-sep", (const u8 *)"new-commitment-proof", 20);
        r.append_data("new_source_commitment_asset", seg, 0, 32);
        r.append_data("new_source_commitment", seg, 32, 32);
        r.append_const("dom-sep", (const u8 *)"equality-proof", 14);
        const char *ys[3] = {"Y_0", "Y_1", "Y_2"};
        for (u32 j = 0; j < 3; j++) { check(seg, 64 + 32 * j, errkey(ST_COMMITMENT, k, 2, XHE_ERR_TRANSCRIPT_IDENTITY)); r.append_data(ys[j], seg, 64 + 32 * j, 32); }
        r.challenge("c");
        r.append_data("z_s", seg, 160, 32); r.append_data("z_x", seg, 192, 32); r.append_data("z_r", seg, 224, 32);
        r.challenge("w");
    }
    if (s.type == XHE_TX_TRANSFERS) {
        for (u32 k = 0; k < T; k++) {
            u32 seg = 2 + A + 2 * k, pseg = seg + 1;
            r.append_const("dom-sep", (const u8 *)"transfer-proof", 14);
            r.append_data("dest_pubkey", seg, 32, 32);
            r.append_data("amount_commitment", seg, 64, 32);
            r.append_data("amount_sender_handle", seg, 96, 32);
            r.append_data("amount_receiver_handle", seg, 128, 32);
            r.append_const("dom-sep", (const u8 *)"validity-proof", 14);
            const char *ys[3] = {"Y_0", "Y_1", "Y_2"};
            for (u32 j = 0; j < 3; j++) { check(pseg, 32 * j, errkey(ST_TRANSFER, k, 3, XHE_ERR_TRANSCRIPT_IDENTITY)); r.append_data(ys[j], pseg, 32 * j, 32); }
            r.challenge("c");
            r.append_data("z_r", pseg, 96, 32); r.append_data("z_x", pseg, 128, 32);
            r.challenge("w");
        }
    } else if (s.type == XHE_TX_BURN) {
        r.append_const("dom-sep", (const u8 *)"burn-proof", 10);
        r.append_data("asset", TT_SEG_PRE, 49, 32);
        r.append_data("amount", TT_SEG_PRE, 49 + 32, 8, true);
    } else if (s.type == XHE_TX_MULTISIG) {
        r.append_const("dom-sep", (const u8 *)"multisig-proof", 14);
        { r.frame("threshold", 8); r.begin_op(2); r.absorb_data(TT_SEG_PRE, 49, 1); const u8 z[7] = {0}; r.absorb_const(z, 7); }
        for (u32 k = 0; k < s.n_items; k++) r.append_data("signer", TT_SEG_PRE, 50 + 32 * k, 32);
    }
    // aggregated range proof
    r.append_const("dom-sep", (const u8 *)"rangeproof v1", 13);
    r.append_u64("n", 64); r.append_u64("m", s.m);
    for (u32 k = 0; k < A; k++) r.append_data("V", 2 + k, 32, 32);
    for (u32 k = 0; k < T; k++) r.append_data("V", 2 + A + 2 * k, 64, 32);
    { const u8 zero[32] = {0}; for (u32 k = A + T; k < s.m; k++) r.append_const("V", zero, 32); }
    check(TT_SEG_RP, 0, 0); r.append_data("A", TT_SEG_RP, 0, 32);
    check(TT_SEG_RP, 32, 0); r.append_data("S", TT_SEG_RP, 32, 32);
    r.challenge("y"); r.challenge("z");
    check(TT_SEG_RP, 64, 0); r.append_data("T_1", TT_SEG_RP, 64, 32);
    check(TT_SEG_RP, 96, 0); r.append_data("T_2", TT_SEG_RP, 96, 32);
    r.challenge("x");
    r.append_data("t_x", TT_SEG_RP, 128, 32); r.append_data("t_x_blinding", TT_SEG_RP, 160, 32); r.append_data("e_blinding", TT_SEG_RP, 192, 32);
    r.challenge("w");
    r.append_const("dom-sep", (const u8 *)"ipp v1", 6);
    r.append_u64("n", 64ull * s.m);
    for (u32 k = 0; k < s.lg; k++) {
        check(TT_SEG_RP, 224 + 64 * k, 0); r.append_data("L", TT_SEG_RP, 224 + 64 * k, 32);
        check(TT_SEG_RP, 224 + 64 * k + 32, 0); r.append_data("R", TT_SEG_RP, 224 + 64 * k + 32, 32);
        r.challenge("u");
    }
    assert(r.n_chal == 2 * A + 2 * T + 4 + s.lg);
    // serialise (the unfinished last block carries nothing a challenge depends on)
    u32 base = (u32)out.size();
    u32 n_blocks = (u32)r.chal_after.size();
    out.push_back(n_blocks); out.push_back(r.n_chal); out.push_back((u32)checks.size() / 2); out.push_back(0);
    out.insert(out.end(), r.chal_after.begin(), r.chal_after.end());
    out.insert(out.end(), checks.begin(), checks.end());
    while (out.size() % 8) out.push_back(0);
    out[base + TT_H_DESC] = (u32)out.size();
    for (u32 b = 0; b < n_blocks; b++)
        for (u32 w = 0; w < TT_WORDS; w++) {
            const u32 *bd = r.desc.data() + (u64)b * TT_BLOCK_BYTES + 8 * w;
            const u8 *br = r.rev.data() + (u64)b * TT_BLOCK_BYTES + 8 * w;
            u32 d[TT_WDESC] = {0, 0, 0, 0, 0, 0, 0, 0};
            u32 n_runs = 0;
            for (u32 j = 0; j < 8;) {
                if (bd[j] & TT_CONST) { u64 c = (u64)(bd[j] & 0xffu) << (8 * j); d[0] |= (u32)c; d[1] |= (u32)(c >> 32); j++; continue; }
                // maximal run of bytes of one segment whose offsets step by +1 (or -1 for a reversed field)
                const u32 seg = bd[j] >> 16, off = bd[j] & 0xffffu; const bool reversed = br[j] != 0;
                u32 len = 1;
                while (j + len < 8 && !(bd[j + len] & TT_CONST) && (bd[j + len] >> 16) == seg && (br[j + len] != 0) == reversed &&
                       (bd[j + len] & 0xffffu) == (reversed ? off - len : off + len)) len++;
                assert(n_runs < 2 && "a state word holds bytes of at most two fields");
                // forward: byte j of the word is source byte off, so the load starts at off - j; reversed: after the
                // end-for-end swap byte j is source byte start + 7 - j, so start = off + j - 7
                const int start = reversed ? (int)off + (int)j - 7 : (int)off - (int)j;
                assert(start + 8 >= 0 && start + 8 < (int)TT_MAX_OFF);
                d[2 + 2 * n_runs] = (seg << 16) | (u32)(start + 8);
                d[3 + 2 * n_runs] = j | (len << 4) | (reversed ? TT_RUN_REVERSED : 0) | TT_RUN_VALID;
                n_runs++;
                j += len;
            }
            out.insert(out.end(), d, d + TT_WDESC);
        }
    while (out.size() % 8) out.push_back(0);
    return base;
}

// segment table of one transaction: offsets relative to the record start
template <typename REC>
static inline void transcript_segments(std::vector<u32> &out, const REC &t) {
    out.push_back(t.pre_off); out.push_back(t.rp_off);
    for (u32 k = 0; k < t.n_commitments; k++) out.push_back(t.cm_off + 256 * k);
    for (size_t k = 0; k < t.transfer_off.size(); k++) { out.push_back(t.transfer_off[k]); out.push_back(t.transfer_off[k] + t.transfer_len[k] - 160); }
}

}  // namespace xhe
