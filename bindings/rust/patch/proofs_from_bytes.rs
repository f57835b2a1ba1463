
// ---- appended by bindings/rust/apply.sh (MI355X binding): the inverses of the proofs' to_bytes ----
#[cfg(feature = "mi355x")]
impl CommitmentEqProof {
    /// Y_0 || Y_1 || Y_2 || z_s || z_x || z_r, 192 bytes; None when a scalar is not canonical.
    pub(crate) fn from_bytes(bytes: &[u8]) -> Option<Self> {
        if bytes.len() != 192 {
            return None;
        }
        let point = |i: usize| CompressedRistretto(bytes[32 * i..32 * i + 32].try_into().unwrap());
        let scalar = |i: usize| Option::<Scalar>::from(Scalar::from_canonical_bytes(bytes[32 * i..32 * i + 32].try_into().unwrap()));
        Some(Self { Y_0: point(0), Y_1: point(1), Y_2: point(2), z_s: scalar(3)?, z_x: scalar(4)?, z_r: scalar(5)? })
    }
}

#[cfg(feature = "mi355x")]
impl CiphertextValidityProof {
    /// Y_0 || Y_1 || Y_2 || z_r || z_x, 160 bytes; None when a scalar is not canonical.
    pub(crate) fn from_bytes(bytes: &[u8]) -> Option<Self> {
        if bytes.len() != 160 {
            return None;
        }
        let point = |i: usize| CompressedRistretto(bytes[32 * i..32 * i + 32].try_into().unwrap());
        let scalar = |i: usize| Option::<Scalar>::from(Scalar::from_canonical_bytes(bytes[32 * i..32 * i + 32].try_into().unwrap()));
        Some(Self { Y_0: point(0), Y_1: point(1), Y_2: point(2), z_r: scalar(3)?, z_x: scalar(4)? })
    }
}
