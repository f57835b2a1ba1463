// Device-resident tables of one uploaded batch: what the host ingest (xhe_ingest.h) lays out and the
// verification kernels (xhe_k_*.hip) consume.  All offsets into `bytes` are absolute byte offsets
// into one device byte buffer holding the transaction records followed by the ledger's account keys,
// balance ciphertexts and multisig signer keys.
//
// Error precedence of Transaction::pre_verify (/root/reference/src/tx/verify.rs:203-485) is carried as a
// 32-bit key per transaction, smaller = earlier: [31:28] stage, [27:12] index, [11:8] sub-stage,
// [7:0] error code.  Every kernel that detects a failure does atomicMin on the transaction's key.
#pragma once
#include <stdint.h>
#include "xhe_settings.h"
#include <stdlib.h>
#include <vector>
#include "xhe_fe.h"

namespace xhe {

static constexpr u32 ERRKEY_NONE = 0xffffffffu;
XHE_HD u32 errkey(u32 stage, u32 index, u32 sub, u32 code) { return (stage << 28) | ((index & 0xffffu) << 12) | ((sub & 15u) << 8) | (code & 0xffu); }
enum : u32 {
    ST_NONCE = 0, ST_ASSETS = 1, ST_DECOMPRESS = 2, ST_SIGNATURE = 5, ST_MULTISIG = 6, ST_COMMITMENT = 7, ST_TRANSFER = 8, ST_TYPE_FORMAT = 9
};

// one per transaction
struct dev_tx {
    u64 rec;                 // record start in bytes[]
    u32 pre_off, pre_len;    // signing pre-image (without multisig entries), relative to rec
    u32 payload_off, rp_off, cm_off, ms_off, sig_off;
    u32 xfer_first;          // index of the first transfer of this tx in dev_xfer[]
    u32 cm_first;            // index of the first commitment of this tx in dev_cm[]
    u16 n_transfers, n_commitments, n_multisig, n_items;
    u8 type, lg, version, flags;
    u32 m;                   // value commitments of the range proof (power of two)
    u64 fee, nonce;
    u32 chal_off;            // first challenge of this tx in chal[]: (c,w) per commitment, (c,w) per transfer, y,z,x,w,u_1..u_lg
    u32 rp_term_off;         // first MSM term of the range proof: A,S,T1,T2,L..,R..,V.. (real commitments only)
    u32 rp_pt;               // point index of A (S,T_1,T_2,L_0,R_0,L_1,... follow)
    u32 src_pt;              // point index of the source key
    u32 hash_idx;            // index into ms_hash[] (BLAKE3 of the pre-image) or 0xffffffff
    u32 rp_row;              // row of this tx inside its nm-group's scalar matrix
    u32 tmpl_off;            // word offset of the transaction's transcript template, 0xffffffff: serial transcript kernel
    u32 seg_off;             // first entry of its segment table (record-relative offsets)
    u32 reserved;
};

// one per new source commitment (CommitmentEqProof)
struct dev_cm {
    u32 tx, idx;
    u64 off;                 // asset[32], commitment[32], Y_0,Y_1,Y_2,z_s,z_x,z_r -- absolute offset in bytes[]
    u32 pt;                  // point index of C_new, then Y_0, Y_1, Y_2
    u32 src_ct_pt;           // point index of the sender's new ciphertext C (D follows), produced by the ledger scan
    u32 term_off;            // 7 MSM terms
    u32 slot_pt;             // point index of the slot's initial ciphertext C (D follows)
    u32 first_touch;         // 1 if this commitment is the first reader of the slot's initial value
};

// one per transfer (CiphertextValidityProof)
struct dev_xfer {
    u32 tx, idx;
    u64 off;                 // asset, dest, C, D_s, D_r, [extra], proof -- absolute offset of the transfer
    u64 proof_off;           // absolute offset of Y_0,Y_1,Y_2,z_r,z_x
    u32 pt;                  // point index of C, then D_s, D_r, Y_0, Y_1, Y_2
    u32 dest_pt;             // point index of the destination key (0xffffffff: account unknown to the ledger)
    u32 term_off;            // 8 MSM terms
    u32 slot_pt;             // receiver slot's initial ciphertext
    u32 first_touch;
    u32 state_err;           // 1: receiver account/asset missing in the ledger
};

// one per point to decode
struct dev_dec { u64 off; };      // byte offset of the 32-byte encoding

// ledger scan: elementary operations sorted by (slot, order)
struct dev_op {
    u32 slot;
    u32 kind;                // 0: +point pair, 1: -point pair, 2: -(amount*G) on the commitment only
    u32 c_pt, d_pt;          // point indices (kind 0/1)
    u64 amount;              // kind 2
    u32 emit;                // 0xffffffff, or index into the emitted ciphertext list (sender's new balance)
    u32 first;               // 1: first operation of its slot (the initial balance)
};

// Transcript kernels: transactions sorted by template and cut into tiles of 64 (one wavefront); the staged
// byte stream of a tile starts at stream_off (in 8-byte words) inside the pass's buffer and holds
// [block][state word 0..20][lane] for the n_blocks blocks of the tile's template.
struct transcript_tile { u64 stream_off; u32 tmpl_off; u32 n_blocks; };
// host side: consecutive tiles that share one stream buffer of stream_words words (the passes reuse it)
struct transcript_pass { u32 tile0, n_tiles, max_blocks; };
struct transcript_plan { std::vector<transcript_pass> passes; u64 stream_words = 0; };
// stream buffer per pass: 1 GiB, or XHE_TRANSCRIPT_CAP_WORDS 8-byte words (tests lower it to exercise several passes)
static inline u64 transcript_stream_cap_words() {
    const char *e = setting("XHE_TRANSCRIPT_CAP_WORDS");
    u64 v = e ? strtoull(e, nullptr, 10) : 0;
    return v ? v : (1ull << 27);
}

// one per signature to verify
// Per distinct signing key: SIG_KEY_POS positions (2^(64 j) P) x multiples 0..8, each as the four 10-limb multiplicands of
// the "cached" form (Y - X, Y + X, Z, 2d T): 40 words per entry.  Four positions: 192 doublings per key and 64 per signature;
// with two or eight positions the 10 000-transaction batch (1 024 keys) took 1.47 ms instead of 1.40 ms on the same device.
static constexpr unsigned SIG_KEY_POS = 4, SIG_KEY_ENTRY_WORDS = 40, SIG_KEY_TABLE_WORDS = SIG_KEY_POS * 9 * SIG_KEY_ENTRY_WORDS;
struct dev_sig {
    u32 tx;
    u32 key_pt;              // decoded public key
    u64 key_off;             // its 32 bytes
    u64 msg_off; u32 msg_len;  // message bytes in bytes[] ...
    u32 hash_idx;            // ... or, if != 0xffffffff, the 32-byte BLAKE3 digest ms_hash[hash_idx]
    u64 sig_off;             // s || e
    u32 err_key;             // key raised when the signature does not verify
    u32 key_err_key;         // key raised when the public key does not decode (0 = checked elsewhere)
    u32 key_slot;            // position of the key among the batch's distinct signing keys (its table of multiples)
};


// ---- descriptor expansion --------------------------------------------------------------------------------
// The host's ordered walk over a batch decides few numbers per transaction, transfer and commitment (where its points, terms,
// operations and signatures go); the descriptors above and the lists that follow from them (one entry per point, term or
// operation) are written on the device by k_expand_descriptors from these records.
struct expand_tx {
    u64 rec, fee, nonce;
    u32 pre_off, pre_len, payload_off, rp_off, cm_off, ms_off, sig_off, m;
    u32 hash_idx, xfer_first, cm_first, chal_off, rp_term_off, rp_row, tmpl_off, seg_off;
    u32 step_err;            // precedence key of what the walk itself found (multisig structure, MultiSig payload), or none
    u32 blk, early_base, key_base, bulk_base;      // first position of the transaction's block in decs[] and in the three decode lists
    u32 src_pos;             // position in decs[] of the source key
    u32 sig_base, src_keyslot;
    u32 ms_first, ms_count;  // its co-signatures in expand_ms[]
    u32 cm_term, xf_term;    // first sigma-proof term of its commitments / transfers
    u32 src_key_term;        // where its terms on the source key go in key_terms[] (0xffffffff: that key has a single term or is unknown)
    u16 n_transfers, n_commitments, n_multisig, n_items;
    u8 type, lg, version, flags;
    u8 src_known, nonce_ok, assets_ok, pad;
};
struct expand_xf {
    u32 off, len;            // the transfer inside its record
    u32 dest_pos;            // position in decs[] of the destination key (used when has_account)
    u32 slot, slot_pos, op_pos;      // receiver slot (0xffffffff: the ledger has none), position of its initial ciphertext, of this transfer's operation
    u32 key_term;            // where its term on the destination key goes in key_terms[], or 0xffffffff
    u16 cm;                  // the source commitment with the transfer's asset, 0xffff: none
    u8 fresh, has_account;
};
struct expand_cm {
    u64 amount;              // plain amount leaving the slot besides the fee (burn, contract deposit)
    u32 slot, slot_pos, op_pos;      // sender slot (0xffffffff: none), its initial ciphertext, first of this commitment's operations
    u8 fresh, native, pad[2];
};
struct expand_ms { u64 key_off; u32 tx, k, key_pos, key_slot; };
struct expand_args {
    const expand_tx *txs; const expand_xf *xfs; const expand_cm *cms; const expand_ms *mss; u32 n_txs;
    u32 n_static, emit_base;
    dev_tx *out_txs; dev_cm *out_cms; dev_xfer *out_xfers; dev_dec *decs; u32 *early_decs, *key_decs, *bulk_decs; dev_op *ops; dev_sig *sigs;
    u32 *segs, *merge_pairs, *key_terms, *tx_err, *apply_err;
    // the points shared between transactions and the slots' initial operations
    const dev_dec *shared_decs; const u32 *shared_early, *shared_key, *slot_pos, *slot_base;
    u32 n_shared, n_shared_early, n_shared_key, n_slots, blk_end, early_end, key_end;
    u32 *slot_last_op;
    // the scalars' range check (record_scalars_canonical, xhe_record.h) on the device: the smallest index of a transaction with a
    // scalar that is not below the group order, by atomicMin (0xffffffff before the launch)
    const u8 *bytes; u32 *noncanonical_first;
};

}  // namespace xhe
