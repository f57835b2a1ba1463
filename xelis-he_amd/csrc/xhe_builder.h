// Transaction builder (prover), host C++: the counterpart of the reference's TransactionBuilder
// (/root/reference/src/tx/builder.rs:320-554), CommitmentEqProof::new and CiphertextValidityProof::new
// (/root/reference/src/proofs.rs:73-132,237-279), ElGamalKeypair::{keygen_with_secret, sign}
// (/root/reference/src/elgamal.rs:194-218), Pedersen commitments and decrypt handles (:223-276) and the
// aggregated Bulletproofs prover RangeProof::prove_multiple the reference calls at builder.rs:525-533.
// It emits the flat transaction record of include/xhe.h.  CPU only: it creates the inputs of the
// verification path and is not part of it.
#pragma once
#include <array>
#include <string>
#include <vector>
#include "xhe_host.h"

namespace xhe {
namespace host {

typedef std::array<u8, 32> bytes32;
typedef std::array<u8, 64> bytes64;

struct keypair {
    sc secret; ge pub; bytes32 pub_bytes;
    // ElGamalKeypair::keygen_with_secret: P = s^-1 * H (elgamal.rs:102-107)
    static keypair from_secret(const sc &s) {
        keypair k; k.secret = s; k.pub = mul_H(sc_invert(s)); encode(k.pub_bytes.data(), k.pub);
        return k;
    }
};

struct ciphertext { ge C, D; };
inline bytes64 ct_bytes(const ciphertext &c) { bytes64 r; encode(r.data(), c.C); encode(r.data() + 32, c.D); return r; }
inline bool ct_from_bytes(ciphertext &c, const u8 b[64]) { return ristretto_decode(c.C, b) && ristretto_decode(c.D, b + 32); }

// PedersenCommitment::new_with_opening: x*G + r*H (elgamal.rs:266-271)
inline ge commit(u64 amount, const sc &opening) { return ge_add(mul_G(sc_from_u64(amount)), mul_H(opening)); }
// ElGamalPubkey::encrypt_with_opening (elgamal.rs:116-125)
inline ciphertext encrypt(const ge &pub, u64 amount, const sc &opening) { ciphertext c; c.C = commit(amount, opening); c.D = scalarmul(opening, pub); return c; }

// Schnorr signature (elgamal.rs:194-200): k random, r = k*H, e = H(pk || msg || r), s = sk^-1 * e + k
bytes64 sign(const keypair &kp, const u8 *msg, size_t n, drbg &rng);
// Signature::verify on the host (elgamal.rs:38-42), used by the builder's self-checks and tests
bool signature_verify_host(const u8 sig[64], const u8 *msg, size_t n, const u8 pub[32]);

struct transfer_in { bytes32 asset; u64 amount; bytes32 dest; std::vector<u8> extra_data; /* plaintext, empty = None */ };
struct asset_balance { bytes32 asset; u64 balance; bytes64 ct; };   // GetBlockchainAccountBalance for one asset

struct builder_in {
    u8 version = 1;
    u8 type = 0;                          // XHE_TX_*
    std::vector<transfer_in> transfers;   // Transfers
    bytes32 burn_asset{}; u64 burn_amount = 0;          // Burn
    std::vector<bytes32> signers; u8 threshold = 0;     // MultiSig
    bytes32 contract{}; std::vector<std::pair<bytes32, u64>> contract_assets; std::vector<u8> opaque;  // CallContract (params bytes) / DeployContract (code)
    u64 fee = 0, nonce = 0;
    std::vector<asset_balance> balances;  // state for every used asset
};

// TransactionUnsigned (builder.rs:110-119): the record without signatures
struct unsigned_tx {
    std::vector<u8> header;     // 16 bytes + extra_len table
    std::vector<u8> preimage;   // to_bytes() without multisig
    size_t rp_off = 0, rp_len = 0;   // where the range proof lies inside the pre-image
    std::vector<std::pair<u8, bytes64>> multisig; bool has_multisig = false;
    bytes32 hash() const;       // BLAKE3 of the pre-image (builder.rs:192-195)
    std::vector<u8> sign(const keypair &kp, drbg &rng) const;   // builder.rs:203-218 -> full record
};

enum { BUILD_OK = 0, BUILD_ERR_DECOMPRESSION = 1, BUILD_ERR_INSUFFICIENT_FUNDS = 2, BUILD_ERR_FORMAT = 3, BUILD_ERR_STATE = 4 };

// A range proof left to the batched GPU prover: everything RangeProof::prove_multiple starts from.
struct pending_range_proof {
    transcript t;                   // after "rangeproof v1", n, m and the value commitments
    std::vector<u64> values; std::vector<sc> openings;
    u8 seed[32]; u64 ctr;           // the random generator at the proof's first draw
};

// TransactionBuilder::build_unsigned.  new_balances (optional) receives the source's ciphertext per
// used asset after this transaction, in the order of the emitted commitments.  With `defer` the range
// proof is not computed: its place in the pre-image (rp_off, rp_len) holds zeros and `defer` what the
// prover needs; the random generator ends in the state it would have after proving.
int build_unsigned(const builder_in &in, const keypair &source, drbg &rng, unsigned_tx &out,
                   std::vector<asset_balance> *new_balances, pending_range_proof *defer = nullptr);

// the transcript part of the proof that precedes every random draw: domain separator, n, m, value commitments
void range_proof_begin(transcript &t, const std::vector<u64> &values, const std::vector<sc> &openings);

// aggregated 64-bit range proof over values/openings (m = power of two), appended to `out`
void range_proof_prove(std::vector<u8> &out, transcript &t, const std::vector<u64> &values, const std::vector<sc> &openings, drbg &rng);

}  // namespace host
}  // namespace xhe
