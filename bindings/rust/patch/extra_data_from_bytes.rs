
// ---- appended by bindings/rust/apply.sh (MI355X binding): the inverse of ExtraData::to_bytes ----
#[cfg(feature = "mi355x")]
impl ExtraData {
    /// cipher || sender_handle || receiver_handle (the handles are the last 64 bytes)
    pub(crate) fn from_bytes(bytes: &[u8]) -> Option<Self> {
        if bytes.len() < 64 {
            return None;
        }
        let n = bytes.len() - 64;
        Some(Self {
            cipher: AeCipher(bytes[..n].to_vec()),
            sender_handle: CompressedHandle(bytes[n..n + 32].try_into().unwrap()),
            receiver_handle: CompressedHandle(bytes[n + 32..].try_into().unwrap()),
        })
    }
}
