//! A/B: the reference's CPU `Transaction::verify_batch` against `Transaction::verify_batch_gpu` on clones of one ledger.
//!
//! An in-crate test module (`src/tx/gpu_ab.rs`; apply.sh adds `#[cfg(all(test, feature = "mi355x"))] mod gpu_ab;` to
//! `src/tx/mod.rs`): `tx`, `VerificationError` and the compressed point types are private to the crate, so a file under
//! the crate's `tests/` directory could not name them.  Transactions come from the reference's own builder
//! (builder.rs:547-554) in the pattern of its bench (benches/tx.rs:129-189): one sender, a chain of transfers.
//!
//!     cargo test --features mi355x gpu_ab -- --test-threads 1
use std::collections::HashMap;

use super::builder::{TransactionBuilder, TransactionTypeBuilder, TransferBuilder};
use super::gpu::{flatten_records, transaction_from_record, GpuContext};
use super::verify::VerificationError;
use super::{Transaction, TransactionType};
use crate::mock::{Account, GenerationBalance, Ledger};
use crate::{CompressedPubkey, Hash, ProofVerificationError};

const NATIVE: Hash = Hash([0; 32]);

struct Scene {
    ledger: Ledger,
    txs: Vec<Transaction>,
}

fn ledger_of(accounts: &[&Account]) -> Ledger {
    Ledger {
        accounts: accounts.iter().map(|a| (a.keypair.pubkey().compress(), (*a).clone())).collect(),
        multisig_accounts: Default::default(),
    }
}

/// `n_txs` transactions of `n_transfers` transfers each from one sender to `n_transfers` receivers; the prover's view of
/// the sender advances with `apply_without_verify` as in the reference's bench.
fn chain(n_txs: usize, n_transfers: usize) -> Scene {
    let sender = Account::new([(NATIVE, 1_000_000)]);
    let receivers: Vec<Account> = (0..n_transfers).map(|_| Account::new([(NATIVE, 0)])).collect();
    let mut all: Vec<&Account> = vec![&sender];
    all.extend(receivers.iter());
    let ledger = ledger_of(&all);
    let mut prover = ledger.clone();
    let source = sender.keypair.pubkey().compress();
    let mut balance = 1_000_000u64;
    let mut txs = Vec::new();
    for _ in 0..n_txs {
        let builder = TransactionBuilder {
            version: 1,
            source,
            data: TransactionTypeBuilder::Transfers(
                receivers.iter().map(|r| TransferBuilder { dest_pubkey: r.keypair.pubkey().compress(), amount: 3, asset: NATIVE, extra_data: Default::default() }).collect(),
            ),
            fee: 2,
            nonce: 0,
        };
        let tx = builder
            .build(&mut GenerationBalance { balances: [(NATIVE, balance)].into(), account: prover.get_account(&source).clone() }, &sender.keypair)
            .unwrap();
        tx.apply_without_verify(&mut prover).unwrap();
        balance -= 2 + 3 * n_transfers as u64;
        txs.push(tx);
    }
    Scene { ledger, txs }
}

fn same_error(a: &VerificationError<()>, b: &VerificationError<()>) -> bool {
    use ProofVerificationError as P;
    match (a, b) {
        (VerificationError::InvalidNonce, VerificationError::InvalidNonce) => true,
        (VerificationError::State(_), VerificationError::State(_)) => true,
        (VerificationError::Proof(x), VerificationError::Proof(y)) => match (x, y) {
            (P::RangeProof(_), P::RangeProof(_)) => true, // the inner bulletproofs error is not part of the contract
            _ => std::mem::discriminant(x) == std::mem::discriminant(y),
        },
        _ => false,
    }
}

fn balances(ledger: &Ledger) -> HashMap<(CompressedPubkey, Hash), [[u8; 32]; 2]> {
    let mut out = HashMap::new();
    for (key, account) in &ledger.accounts {
        for (asset, ct) in &account.balances {
            out.insert((*key, asset.clone()), ct.0);
        }
    }
    out
}

fn ab(ctx: &GpuContext, scene: &Scene, txs: &[Transaction]) {
    let mut cpu = scene.ledger.clone();
    let mut gpu = scene.ledger.clone();
    let want = Transaction::verify_batch(txs, &mut cpu);
    let got = Transaction::verify_batch_gpu(ctx, txs, &mut gpu);
    match (&want, &got) {
        (Ok(()), Ok(())) => {
            assert_eq!(balances(&cpu), balances(&gpu), "ledger after an accepted batch");
            assert_eq!(cpu.multisig_accounts, gpu.multisig_accounts);
        }
        (Err(a), Err(b)) => {
            assert!(same_error(a, b), "cpu {:?} gpu {:?}", a, b);
            assert_eq!(balances(&scene.ledger), balances(&gpu), "a rejected batch leaves the state as it was");
        }
        _ => panic!("cpu {:?} gpu {:?}", want.is_ok(), got.is_ok()),
    }
}

#[test]
fn record_round_trip() {
    let scene = chain(3, 2);
    let (records, offsets) = flatten_records(&scene.txs);
    for (i, tx) in scene.txs.iter().enumerate() {
        let back = transaction_from_record(&records[offsets[i] as usize..offsets[i + 1] as usize]).expect("record parses");
        assert_eq!(back.to_bytes(), tx.to_bytes());
        assert_eq!(back.signature.to_bytes(), tx.signature.to_bytes());
    }
}

#[test]
fn accepts_what_the_cpu_accepts() {
    let ctx = GpuContext::new(0).unwrap();
    for (n_txs, n_transfers) in [(1, 1), (4, 1), (3, 8), (32, 1)] {
        let scene = chain(n_txs, n_transfers);
        ab(&ctx, &scene, &scene.txs);
    }
    let scene = chain(2, 1);
    ab(&ctx, &scene, &[]);
}

#[test]
fn rejects_what_the_cpu_rejects() {
    let ctx = GpuContext::new(0).unwrap();
    let scene = chain(6, 2);

    // a gap in the chain: later proofs are against a balance never reached (aggregate GenericProof)
    let mut gap: Vec<Transaction> = scene.txs.clone();
    gap.remove(1);
    ab(&ctx, &scene, &gap);

    // fee changed after signing: the signature covers it (verify.rs:283-290)
    let mut fee = scene.txs.clone();
    fee[2].fee += 1;
    ab(&ctx, &scene, &fee);

    // nonce that does not match the account's (verify.rs:214-221)
    let mut nonce = scene.txs.clone();
    nonce[0].nonce = 5;
    ab(&ctx, &scene, &nonce);

    // two transfers swapped inside a transaction: signature and transcript change
    let mut swapped = scene.txs.clone();
    if let TransactionType::Transfers(t) = &mut swapped[3].data {
        t.swap(0, 1);
    }
    ab(&ctx, &scene, &swapped);

    // a commitment that is not a curve point
    let mut point = scene.txs.clone();
    point[4].new_source_commitments[0].new_source_commitment.0 = [0xff; 32];
    ab(&ctx, &scene, &point);

    // the range proof of another transaction
    let mut rp = scene.txs.clone();
    rp[1].range_proof = scene.txs[0].range_proof.clone();
    ab(&ctx, &scene, &rp);
}

#[test]
fn multisig_configuration_and_use() {
    let ctx = GpuContext::new(0).unwrap();
    let alice = Account::new([(NATIVE, 100)]);
    let bob = Account::new([(NATIVE, 0)]);
    let charlie = Account::new([(NATIVE, 0)]);
    let source = alice.keypair.pubkey().compress();
    let mut ledger = ledger_of(&[&alice, &bob, &charlie]);

    // set-up transaction (verify.rs:366-374), then on a ledger that has it a co-signed transfer (lib.rs:310-366)
    let setup = TransactionBuilder {
        version: 1,
        source,
        data: TransactionTypeBuilder::MultiSig { signers: vec![charlie.keypair.pubkey().compress()], threshold: 1 },
        fee: 1,
        nonce: 0,
    }
    .build(&mut GenerationBalance { balances: [(NATIVE, 100)].into(), account: alice.clone() }, &alice.keypair)
    .unwrap();
    let scene = Scene { ledger: ledger.clone(), txs: vec![setup] };
    ab(&ctx, &scene, &scene.txs);

    use super::verify::BlockchainVerificationState;
    ledger.set_multisig_for_account(&source, &vec![charlie.keypair.pubkey().compress()], 1).unwrap();
    let transfer = |cosign: bool| {
        let mut unsigned = TransactionBuilder {
            version: 1,
            source,
            data: TransactionTypeBuilder::Transfers(vec![TransferBuilder { dest_pubkey: bob.keypair.pubkey().compress(), amount: 10, asset: NATIVE, extra_data: Default::default() }]),
            fee: 1,
            nonce: 0,
        }
        .build_unsigned(&mut GenerationBalance { balances: [(NATIVE, 100)].into(), account: alice.clone() }, &alice.keypair)
        .unwrap();
        if cosign {
            let hash = unsigned.hash();
            unsigned.set_multisig(vec![(0, charlie.keypair.sign(&hash.0))]);
        }
        unsigned.sign(&alice.keypair)
    };
    let scene = Scene { ledger: ledger.clone(), txs: vec![transfer(true)] };
    ab(&ctx, &scene, &scene.txs);
    let scene = Scene { ledger, txs: vec![transfer(false)] }; // required and missing: Format (verify.rs:246-251)
    ab(&ctx, &scene, &scene.txs);
}
