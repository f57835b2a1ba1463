"""Numeric codes shared with include/xhe.h."""
OK = 0
ERR_STATE = 1
ERR_INVALID_NONCE = 2
ERR_SIGNATURE = 3
ERR_DECOMPRESSION = 4
ERR_COMMITMENT_EQ = 5
ERR_CT_VALIDITY = 6
ERR_GENERIC_PROOF = 7
ERR_RANGE_PROOF = 8
ERR_TRANSCRIPT_IDENTITY = 9
ERR_FORMAT = 10
ERR_MALFORMED = 11

E_ARG = -1
E_HIP = -2
E_NO_DEVICE = -3
E_DECODE = -4

TX_TRANSFERS = 0
TX_BURN = 1
TX_MULTISIG = 2
TX_CALL_CONTRACT = 3
TX_DEPLOY_CONTRACT = 4

NAMES = {
    OK: "Ok", ERR_STATE: "State", ERR_INVALID_NONCE: "InvalidNonce", ERR_SIGNATURE: "Proof(Signature)",
    ERR_DECOMPRESSION: "Proof(Decompression)", ERR_COMMITMENT_EQ: "Proof(CommitmentEqProof)",
    ERR_CT_VALIDITY: "Proof(CiphertextValidityProof)", ERR_GENERIC_PROOF: "Proof(GenericProof)",
    ERR_RANGE_PROOF: "Proof(RangeProof)", ERR_TRANSCRIPT_IDENTITY: "Proof(Transcript(IdentityPoint))",
    ERR_FORMAT: "Proof(Format)", ERR_MALFORMED: "Malformed",
}
