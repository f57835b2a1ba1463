//! `Transaction::verify_batch` on an MI355X through the C ABI of libxhe.so (include/xhe.h).
//!
//! This file is a module of the xelis-he crate (`src/tx/gpu.rs`, behind the feature `mi355x`): as a child of `tx` it reads
//! `Transaction`'s crate-private fields and builds `Transfer` values.  Nothing else of the crate changes; the CPU
//! `verify_batch`, the builder and the wallet code keep working as they are.
//!
//! * `GpuContext` owns one `xhe_ctx` (one GPU).
//! * `Transaction::verify_batch_gpu(ctx, txs, state)` has the contract of `Transaction::verify_batch` (verify.rs:487-517):
//!   `Ok(())`, or the first error in transaction order, `GenericProof` before `RangeProof` for the aggregate checks.  The
//!   state is read once up front through the trait (verify.rs:25-77) and written once on accept; on `Err` it is untouched
//!   (the CPU path leaves it half-updated, its callers clone).
//! * `flatten_records` / `transaction_from_record` convert between `Transaction` and the flat record of include/xhe.h.
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int};

use bulletproofs::RangeProof;
use curve25519_dalek::Scalar;
use rand::RngCore;

use super::{MultiSig, NewSourceCommitment, SmartContractCall, Transaction, TransactionType, Transfer};
use super::verify::{BlockchainVerificationState, VerificationError};
use crate::{
    compressed::{CompressedCommitment, CompressedHandle, DecompressionError},
    extra_data::ExtraData,
    proofs::{CiphertextValidityProof, CommitmentEqProof},
    CompressedCiphertext, CompressedPubkey, ElGamalCiphertext, Hash, ProofVerificationError, Role, Signature, TranscriptError,
};

#[repr(C)]
pub struct XheCtx {
    _private: [u8; 0],
}

#[repr(C)]
pub struct XheLedgerView {
    n_accounts: u64,
    account_keys: *const u8,
    account_nonces: *const u64,
    n_balances: u64,
    balance_account: *const u32,
    balance_assets: *const u8,
    balance_cts: *const u8,
    n_multisig: u64,
    multisig_account: *const u32,
    multisig_threshold: *const u8,
    multisig_offsets: *const u32,
    multisig_signers: *const u8,
}

#[repr(C)]
pub struct XheLedgerOut {
    balance_cts: *mut u8,
    account_nonces: *mut u64,
    multisig_set_by: *mut i64,
}

#[repr(C)]
#[derive(Default, Debug, Clone, Copy)]
pub struct XheVerdict {
    pub accept: i32,
    pub err_code: i32,
    pub first_bad_tx: i64,
}

extern "C" {
    fn xhe_create(device_id: c_int) -> *mut XheCtx;
    fn xhe_destroy(ctx: *mut XheCtx);
    fn xhe_last_error(ctx: *const XheCtx) -> *const c_char;
    fn xhe_verify_batch(
        ctx: *mut XheCtx,
        records: *const u8,
        offsets: *const u64,
        n_txs: u64,
        ledger_in: *const XheLedgerView,
        ledger_out: *const XheLedgerOut,
        verdict: *mut XheVerdict,
        rng_seed: *const u8,
    ) -> i32;
}

/// One GPU.  Not `Sync`: a context serves one call at a time (include/xhe.h).
pub struct GpuContext(*mut XheCtx);

impl GpuContext {
    pub fn new(device_id: i32) -> Result<Self, String> {
        let ctx = unsafe { xhe_create(device_id as c_int) };
        if ctx.is_null() {
            let msg = unsafe { CStr::from_ptr(xhe_last_error(std::ptr::null())) };
            return Err(msg.to_string_lossy().into_owned());
        }
        Ok(GpuContext(ctx))
    }

    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(xhe_last_error(self.0)) }.to_string_lossy().into_owned()
    }
}

impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe { xhe_destroy(self.0) }
    }
}

// Transaction types of the record header (include/xhe.h XHE_TX_*)
const TX_TRANSFERS: u8 = 0;
const TX_BURN: u8 = 1;
const TX_MULTISIG: u8 = 2;
const TX_CALL_CONTRACT: u8 = 3;
const TX_DEPLOY_CONTRACT: u8 = 4;

/// One record per transaction (layout: include/xhe.h, "Transaction record"): a 16-byte header, for Transfers the table of
/// extra-data lengths, then `Transaction::to_bytes` (verify.rs:623-688), which already ends with the multisig entries,
/// then the signature.
pub fn flatten_records(txs: &[Transaction]) -> (Vec<u8>, Vec<u64>) {
    let mut records = Vec::new();
    let mut offsets = Vec::with_capacity(txs.len() + 1);
    for tx in txs {
        offsets.push(records.len() as u64);
        let (bytes, _multisig_index) = tx.to_bytes();
        let (ty, n_items, opaque_len) = match &tx.data {
            TransactionType::Transfers(t) => (TX_TRANSFERS, t.len(), 0usize),
            TransactionType::Burn { .. } => (TX_BURN, 0, 0),
            TransactionType::MultiSig { signers, .. } => (TX_MULTISIG, signers.len(), 0),
            TransactionType::CallContract(c) => (TX_CALL_CONTRACT, c.assets.len(), c.params.iter().map(|(k, v)| k.len() + v.len()).sum()),
            TransactionType::DeployContract(code) => (TX_DEPLOY_CONTRACT, 0, code.len()),
        };
        let n_multisig = tx.multisig.as_ref().map_or(0, |m| m.len());
        records.push(ty);
        records.push(tx.multisig.is_some() as u8); // flags bit 0: the multisig field is Some
        records.extend_from_slice(&(n_items as u16).to_le_bytes());
        records.extend_from_slice(&(tx.new_source_commitments.len() as u16).to_le_bytes());
        records.extend_from_slice(&(n_multisig as u16).to_le_bytes());
        records.extend_from_slice(&(opaque_len as u32).to_le_bytes());
        records.extend_from_slice(&(tx.range_proof.to_bytes().len() as u16).to_le_bytes());
        records.extend_from_slice(&0u16.to_le_bytes());
        if let TransactionType::Transfers(transfers) = &tx.data {
            for t in transfers {
                let len = t.extra_data.as_ref().map_or(0, |e| e.to_bytes().len()); // 0 = None
                records.extend_from_slice(&(len as u32).to_le_bytes());
            }
        }
        records.extend_from_slice(&bytes); // pre-image + multisig entries
        records.extend_from_slice(&tx.signature.to_bytes()); // s || e
    }
    offsets.push(records.len() as u64);
    (records, offsets)
}

fn take<'a>(bytes: &mut &'a [u8], n: usize) -> Option<&'a [u8]> {
    if bytes.len() < n {
        return None;
    }
    let (head, tail) = bytes.split_at(n);
    *bytes = tail;
    Some(head)
}
fn take32(bytes: &mut &[u8]) -> Option<[u8; 32]> {
    take(bytes, 32).map(|b| b.try_into().unwrap())
}
fn take_u64_be(bytes: &mut &[u8]) -> Option<u64> {
    take(bytes, 8).map(|b| u64::from_be_bytes(b.try_into().unwrap()))
}
fn take_scalar(bytes: &mut &[u8]) -> Option<Scalar> {
    Option::from(Scalar::from_canonical_bytes(take32(bytes)?))
}
fn take_signature(bytes: &mut &[u8]) -> Option<Signature> {
    let s = take_scalar(bytes)?;
    let e = take_scalar(bytes)?;
    Some(Signature::new(s, e))
}

/// The inverse of `flatten_records` for one record; None for a record the reference's deserialisation would refuse
/// (wrong lengths, scalars that are not canonical, a range proof that does not parse).  CallContract and DeployContract
/// records do not come back: their parameters are a HashMap whose byte order is not recoverable (verify.rs:651-658).
pub fn transaction_from_record(record: &[u8]) -> Option<Transaction> {
    if record.len() < 16 {
        return None;
    }
    let ty = record[0];
    let flags = record[1];
    let n_items = u16::from_le_bytes([record[2], record[3]]) as usize;
    let n_commitments = u16::from_le_bytes([record[4], record[5]]) as usize;
    let n_multisig = u16::from_le_bytes([record[6], record[7]]) as usize;
    let rp_len = u16::from_le_bytes([record[12], record[13]]) as usize;
    let mut rest = &record[16..];
    let mut extra_lens = Vec::new();
    if ty == TX_TRANSFERS {
        for _ in 0..n_items {
            extra_lens.push(u32::from_le_bytes(take(&mut rest, 4)?.try_into().unwrap()) as usize);
        }
    }
    let version = take(&mut rest, 1)?[0];
    let source = CompressedPubkey(take32(&mut rest)?);
    let fee = take_u64_be(&mut rest)?;
    let nonce = take_u64_be(&mut rest)?;
    let data = match ty {
        TX_TRANSFERS => {
            let mut transfers = Vec::with_capacity(n_items);
            for extra_len in extra_lens {
                let asset = Hash(take32(&mut rest)?);
                let dest_pubkey = CompressedPubkey(take32(&mut rest)?);
                let amount_commitment = CompressedCommitment(take32(&mut rest)?);
                let amount_sender_handle = CompressedHandle(take32(&mut rest)?);
                let amount_receiver_handle = CompressedHandle(take32(&mut rest)?);
                let extra_data = if extra_len == 0 { None } else { Some(ExtraData::from_bytes(take(&mut rest, extra_len)?)?) };
                let ct_validity_proof = CiphertextValidityProof::from_bytes(take(&mut rest, 160)?)?;
                transfers.push(Transfer { asset, dest_pubkey, extra_data, amount_commitment, amount_sender_handle, amount_receiver_handle, ct_validity_proof });
            }
            TransactionType::Transfers(transfers)
        }
        TX_BURN => {
            let asset = Hash(take32(&mut rest)?);
            let amount = take_u64_be(&mut rest)?;
            TransactionType::Burn { asset, amount }
        }
        TX_MULTISIG => {
            let threshold = take(&mut rest, 1)?[0];
            let mut signers = Vec::with_capacity(n_items);
            for _ in 0..n_items {
                signers.push(CompressedPubkey(take32(&mut rest)?));
            }
            TransactionType::MultiSig { signers, threshold }
        }
        _ => return None,
    };
    let range_proof = RangeProof::from_bytes(take(&mut rest, rp_len)?).ok()?;
    let mut new_source_commitments = Vec::with_capacity(n_commitments);
    for _ in 0..n_commitments {
        let asset = Hash(take32(&mut rest)?);
        let new_source_commitment = CompressedCommitment(take32(&mut rest)?);
        let new_commitment_eq_proof = CommitmentEqProof::from_bytes(take(&mut rest, 192)?)?;
        new_source_commitments.push(NewSourceCommitment { new_source_commitment, new_commitment_eq_proof, asset });
    }
    let multisig: Option<MultiSig> = if flags & 1 != 0 {
        let mut entries = Vec::with_capacity(n_multisig);
        for _ in 0..n_multisig {
            let id = take(&mut rest, 1)?[0];
            entries.push((id, take_signature(&mut rest)?));
        }
        Some(entries)
    } else if n_multisig != 0 {
        return None;
    } else {
        None
    };
    let signature = take_signature(&mut rest)?;
    if !rest.is_empty() {
        return None;
    }
    Some(Transaction { version, source, data, fee, nonce, new_source_commitments, range_proof, multisig, signature })
}

/// Every (account, asset) balance, nonce and multisig configuration the batch can touch, read once through the trait
/// (verify.rs:25-77) in the order the CPU path would first touch them.
#[derive(Default)]
pub struct LedgerSnapshot {
    keys: Vec<u8>,
    nonces: Vec<u64>,
    account_of: HashMap<CompressedPubkey, u32>,
    is_sender: Vec<bool>,
    balance_account: Vec<u32>,
    balance_assets: Vec<u8>,
    balance_cts: Vec<u8>,
    balance_role_sender: Vec<bool>,
    slot_of: HashMap<(u32, Hash), u32>,
    ms_account: Vec<u32>,
    ms_threshold: Vec<u8>,
    ms_offsets: Vec<u32>,
    ms_signers: Vec<u8>,
}

impl LedgerSnapshot {
    pub fn collect<B: BlockchainVerificationState>(txs: &[Transaction], state: &B) -> Result<Self, B::Error> {
        let mut s = LedgerSnapshot { ms_offsets: vec![0], ..Default::default() };
        for tx in txs {
            let src = s.account(&tx.source, state, true)?; // nonce + multisig configuration of a sender
            for c in &tx.new_source_commitments {
                s.slot(src, &tx.source, &c.asset, Role::Sender, state)?;
            }
            if let TransactionType::Transfers(transfers) = &tx.data {
                for t in transfers {
                    let dst = s.account(&t.dest_pubkey, state, false)?;
                    s.slot(dst, &t.dest_pubkey, &t.asset, Role::Receiver, state)?;
                }
            }
        }
        Ok(s)
    }

    fn account<B: BlockchainVerificationState>(&mut self, key: &CompressedPubkey, state: &B, sender: bool) -> Result<u32, B::Error> {
        if let Some(&i) = self.account_of.get(key) {
            if sender && !self.is_sender[i as usize] {
                // first seen as a receiver: a sender also needs its nonce and its multisig configuration
                self.is_sender[i as usize] = true;
                self.nonces[i as usize] = state.get_account_nonce(key)?;
                self.multisig(i, key, state)?;
            }
            return Ok(i);
        }
        let i = self.nonces.len() as u32;
        self.keys.extend_from_slice(&key.0);
        self.nonces.push(if sender { state.get_account_nonce(key)? } else { 0 });
        self.is_sender.push(sender);
        self.account_of.insert(*key, i);
        if sender {
            self.multisig(i, key, state)?;
        }
        Ok(i)
    }

    fn multisig<B: BlockchainVerificationState>(&mut self, account: u32, key: &CompressedPubkey, state: &B) -> Result<(), B::Error> {
        if let Some((signers, threshold)) = state.get_multisig_for_account(key)? {
            self.ms_account.push(account);
            self.ms_threshold.push(threshold);
            for k in &signers {
                self.ms_signers.extend_from_slice(&k.0);
            }
            self.ms_offsets.push((self.ms_signers.len() / 32) as u32);
        }
        Ok(())
    }

    fn slot<B: BlockchainVerificationState>(&mut self, acc: u32, key: &CompressedPubkey, asset: &Hash, role: Role, state: &B) -> Result<(), B::Error> {
        if self.slot_of.contains_key(&(acc, asset.clone())) {
            return Ok(());
        }
        let is_sender = matches!(role, Role::Sender);
        let ct = state.get_account_balance(key, asset, role)?; // commitment || handle
        self.slot_of.insert((acc, asset.clone()), self.balance_account.len() as u32);
        self.balance_account.push(acc);
        self.balance_assets.extend_from_slice(&asset.0);
        self.balance_cts.extend_from_slice(&ct.0[0]);
        self.balance_cts.extend_from_slice(&ct.0[1]);
        self.balance_role_sender.push(is_sender);
        Ok(())
    }

    /// The pointers and lengths of the vectors above (valid while `self` is not moved or changed).
    pub fn view(&self) -> XheLedgerView {
        XheLedgerView {
            n_accounts: self.nonces.len() as u64,
            account_keys: self.keys.as_ptr(),
            account_nonces: self.nonces.as_ptr(),
            n_balances: self.balance_account.len() as u64,
            balance_account: self.balance_account.as_ptr(),
            balance_assets: self.balance_assets.as_ptr(),
            balance_cts: self.balance_cts.as_ptr(),
            n_multisig: self.ms_account.len() as u64,
            multisig_account: self.ms_account.as_ptr(),
            multisig_threshold: self.ms_threshold.as_ptr(),
            multisig_offsets: self.ms_offsets.as_ptr(),
            multisig_signers: self.ms_signers.as_ptr(),
        }
    }

    /// On accept: the final ciphertext of every balance and the multisig configurations the batch set, through the
    /// trait's update calls (what the CPU path does incrementally at verify.rs:329-341, 366-374, 426).  The nonce is
    /// stored back unchanged, as the CPU path does (verify.rs:214-221).
    pub fn write_back<B: BlockchainVerificationState>(&self, state: &mut B, out_cts: &[u8], ms_set_by: &[i64], txs: &[Transaction]) -> Result<(), B::Error> {
        for (slot, &acc) in self.balance_account.iter().enumerate() {
            let key = CompressedPubkey(self.keys[32 * acc as usize..32 * acc as usize + 32].try_into().unwrap());
            let asset = Hash(self.balance_assets[32 * slot..32 * slot + 32].try_into().unwrap());
            let ct = CompressedCiphertext([out_cts[64 * slot..64 * slot + 32].try_into().unwrap(), out_cts[64 * slot + 32..64 * slot + 64].try_into().unwrap()]);
            let role = if self.balance_role_sender[slot] { Role::Sender } else { Role::Receiver };
            state.update_account_balance(&key, &asset, ct, role)?;
        }
        for (acc, &sender) in self.is_sender.iter().enumerate() {
            if sender {
                let key = CompressedPubkey(self.keys[32 * acc..32 * acc + 32].try_into().unwrap());
                state.update_account_nonce(&key, self.nonces[acc])?;
            }
        }
        for (acc, &by) in ms_set_by.iter().enumerate() {
            if by >= 0 {
                if let TransactionType::MultiSig { signers, threshold } = &txs[by as usize].data {
                    let key = CompressedPubkey(self.keys[32 * acc..32 * acc + 32].try_into().unwrap());
                    state.set_multisig_for_account(&key, signers, *threshold)?;
                }
            }
        }
        Ok(())
    }

    pub fn balance_bytes(&self) -> &[u8] {
        &self.balance_cts
    }
}

/// What `set_output_ciphertext` receives per transaction and spent asset (verify.rs:107-144, 339): the fee (native asset),
/// burnt or deposited amounts as plain scalars and the sender's side of every transfer of that asset.  The library
/// returns final balances only, so a state that records outputs gets them recomputed here after an accepted batch: a
/// decompression and an addition per transfer, no proof work.  Every point was decoded by the verification already, so
/// a failure here is a bug.
pub fn sender_output_ct(tx: &Transaction, asset: &Hash) -> ElGamalCiphertext {
    let mut out = ElGamalCiphertext::zero();
    if asset.is_zeros() {
        out += Scalar::from(tx.fee);
    }
    match &tx.data {
        TransactionType::Transfers(transfers) => {
            for t in transfers.iter().filter(|t| &t.asset == asset) {
                out += t.get_ciphertext(Role::Sender).decompress().expect("verified ciphertext");
            }
        }
        TransactionType::Burn { amount, asset: burnt } if burnt == asset => out += Scalar::from(*amount),
        TransactionType::CallContract(SmartContractCall { assets, .. }) => {
            if let Some(amount) = assets.get(asset) {
                out += Scalar::from(*amount);
            }
        }
        _ => (),
    }
    out
}

/// include/xhe.h error codes -> the reference's variants (verify.rs:16-21, lib.rs:71-89)
pub fn map_error<E>(code: i32) -> VerificationError<E> {
    use ProofVerificationError as P;
    let proof = match code {
        1 => panic!("xhe: unknown account or asset; the snapshot was collected from the same transactions, so this is a bug"),
        2 => return VerificationError::InvalidNonce,
        3 => P::Signature,
        4 => P::Decompression(DecompressionError),
        5 => P::CommitmentEqProof,
        6 => P::CiphertextValidityProof,
        7 => P::GenericProof,
        8 => P::RangeProof(bulletproofs::ProofError::VerificationError),
        9 => P::Transcript(TranscriptError::IdentityPoint),
        _ => P::Format, // 10 = Format; 11 = malformed record, which flatten_records never produces
    };
    VerificationError::Proof(proof)
}

impl Transaction {
    /// Same contract as the CPU `verify_batch` (verify.rs:487-517), including the `set_output_ciphertext` calls
    /// (verify.rs:339), which are made after the balances were written, in transaction order.
    pub fn verify_batch_gpu<B: BlockchainVerificationState>(ctx: &GpuContext, txs: &[Transaction], state: &mut B) -> Result<(), VerificationError<B::Error>> {
        Self::verify_batch_gpu_with(ctx, txs, state, true)
    }

    /// `outputs` = false leaves `set_output_ciphertext` out: for states that ignore it (the mock ledger, lib.rs:172-179)
    /// this saves the only per-transaction point arithmetic left on the host.
    pub fn verify_batch_gpu_with<B: BlockchainVerificationState>(ctx: &GpuContext, txs: &[Transaction], state: &mut B, outputs: bool) -> Result<(), VerificationError<B::Error>> {
        let (records, offsets) = flatten_records(txs);
        let snap = LedgerSnapshot::collect(txs, state).map_err(VerificationError::State)?;
        let mut out_cts = snap.balance_cts.clone();
        let mut out_nonces = snap.nonces.clone();
        let mut ms_set_by = vec![-1i64; snap.nonces.len()];
        let out = XheLedgerOut { balance_cts: out_cts.as_mut_ptr(), account_nonces: out_nonces.as_mut_ptr(), multisig_set_by: ms_set_by.as_mut_ptr() };
        let mut seed = [0u8; 32];
        rand::rng().fill_bytes(&mut seed); // the batch factors' seed (the CPU path draws them from the OS, proofs.rs:181,326)
        let mut v = XheVerdict::default();
        let view = snap.view();
        let rc = unsafe { xhe_verify_batch(ctx.0, records.as_ptr(), offsets.as_ptr(), txs.len() as u64, &view, &out, &mut v, seed.as_ptr()) };
        if rc != 0 {
            panic!("xhe_verify_batch: {} ({})", rc, ctx.last_error());
        }
        if v.accept != 1 {
            return Err(map_error(v.err_code));
        }
        snap.write_back(state, &out_cts, &ms_set_by, txs).map_err(VerificationError::State)?;
        if outputs {
            for tx in txs {
                for c in &tx.new_source_commitments {
                    state.set_output_ciphertext(&tx.source, &c.asset, sender_output_ct(tx, &c.asset)).map_err(VerificationError::State)?;
                }
            }
        }
        Ok(())
    }
}
