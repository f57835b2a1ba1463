"""Python mirror of the reference's transaction surface over the C ABI.

Names follow the reference (/root/reference/src/lib.rs:97-242 mock::{Ledger, Account},
/root/reference/src/tx/builder.rs:62-84 TransactionBuilder / TransferBuilder,
/root/reference/src/tx/mod.rs:103-119 Transaction, /root/reference/src/tx/verify.rs:487-542
Transaction::verify_batch / verify).  Everything here is plumbing: keys, proofs and ledger updates
come from the CPU builder entry points of libxhe.so (include/xhe_builder.h), verification from the
GPU entry points (include/xhe.h)."""
import ctypes
import hashlib
import os

from . import codes
from ._capi import XheError, load_library

NATIVE = bytes(32)
u8p = ctypes.POINTER(ctypes.c_uint8)


class TxRequest(ctypes.Structure):
    _fields_ = [
        ("version", ctypes.c_uint8), ("type", ctypes.c_uint8), ("n_transfers", ctypes.c_uint32),
        ("transfer_assets", ctypes.c_char_p), ("transfer_amounts", ctypes.POINTER(ctypes.c_uint64)),
        ("transfer_dests", ctypes.c_char_p), ("transfer_extra", ctypes.POINTER(ctypes.c_char_p)),
        ("transfer_extra_len", ctypes.POINTER(ctypes.c_uint32)),
        ("burn_asset", ctypes.c_uint8 * 32), ("burn_amount", ctypes.c_uint64),
        ("n_signers", ctypes.c_uint32), ("signers", ctypes.c_char_p), ("threshold", ctypes.c_uint8),
        ("contract", ctypes.c_uint8 * 32), ("n_contract_assets", ctypes.c_uint32),
        ("contract_assets", ctypes.c_char_p), ("contract_amounts", ctypes.POINTER(ctypes.c_uint64)),
        ("opaque", ctypes.c_char_p), ("opaque_len", ctypes.c_uint32),
        ("fee", ctypes.c_uint64), ("nonce", ctypes.c_uint64),
        ("n_balances", ctypes.c_uint32), ("balance_assets", ctypes.c_char_p),
        ("balance_amounts", ctypes.POINTER(ctypes.c_uint64)), ("balance_cts", ctypes.c_char_p),
    ]


class LedgerView(ctypes.Structure):
    _fields_ = [
        ("n_accounts", ctypes.c_uint64), ("account_keys", ctypes.c_char_p), ("account_nonces", ctypes.POINTER(ctypes.c_uint64)),
        ("n_balances", ctypes.c_uint64), ("balance_account", ctypes.POINTER(ctypes.c_uint32)), ("balance_assets", ctypes.c_char_p),
        ("balance_cts", ctypes.c_char_p),
        ("n_multisig", ctypes.c_uint64), ("multisig_account", ctypes.POINTER(ctypes.c_uint32)), ("multisig_threshold", ctypes.c_char_p),
        ("multisig_offsets", ctypes.POINTER(ctypes.c_uint32)), ("multisig_signers", ctypes.c_char_p),
    ]


class LedgerOut(ctypes.Structure):
    _fields_ = [("balance_cts", u8p), ("account_nonces", ctypes.POINTER(ctypes.c_uint64)), ("multisig_set_by", ctypes.POINTER(ctypes.c_int64))]


class Verdict(ctypes.Structure):
    _fields_ = [("accept", ctypes.c_int32), ("err_code", ctypes.c_int32), ("first_bad_tx", ctypes.c_int64)]


class VerificationError(Exception):
    """VerificationError of the reference (src/tx/verify.rs:16-21): .code is a codes.ERR_* value."""

    def __init__(self, code, first_bad_tx=-1):
        super().__init__("%s (tx %d)" % (codes.NAMES.get(code, str(code)), first_bad_tx))
        self.code, self.first_bad_tx = code, first_bad_tx


_seed_counter = [0]


def _fresh_seed(seed=None):
    if seed is not None:
        return hashlib.sha3_256(seed if isinstance(seed, bytes) else str(seed).encode()).digest()
    _seed_counter[0] += 1
    return hashlib.sha3_256(os.urandom(32) + _seed_counter[0].to_bytes(8, "little")).digest()


def _lib():
    lib = load_library()
    if not getattr(lib, "_builder_ready", False):
        lib.xhe_tx_build_unsigned.restype = ctypes.c_int32
        lib.xhe_tx_finalize.restype = ctypes.c_int32
        lib.xhe_tx_hash.restype = ctypes.c_int32
        lib.xhe_tx_apply_host.restype = ctypes.c_int32
        if hasattr(lib, "xhe_verify_batch"):
            lib.xhe_verify_batch.restype = ctypes.c_int32
            lib.xhe_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64,
                                             ctypes.POINTER(LedgerView), ctypes.POINTER(LedgerOut), ctypes.POINTER(Verdict), ctypes.c_char_p]
        # 64-bit amounts: without declared argument types ctypes passes a Python int as a C int
        u8p = ctypes.POINTER(ctypes.c_uint8)
        lib.xhe_encrypt.restype = ctypes.c_int32
        lib.xhe_encrypt.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, u8p]
        lib.xhe_ct_decrypts_to.restype = ctypes.c_int32
        lib.xhe_ct_decrypts_to.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64]
        lib.xhe_decrypt_to_point.restype = ctypes.c_int32
        lib.xhe_decrypt_to_point.argtypes = [ctypes.c_char_p, ctypes.c_char_p, u8p]
        lib.xhe_synth_transfers.restype = ctypes.c_void_p
        lib.xhe_synth_transfers.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int32]
        lib.xhe_synth_transfers_gpu.restype = ctypes.c_void_p
        lib.xhe_synth_transfers_gpu.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int32]
        lib.xhe_synth_mixed.restype = ctypes.c_void_p
        lib.xhe_synth_mixed.argtypes = [ctypes.c_uint64, ctypes.c_char_p, ctypes.c_int32]
        lib.xhe_synth_mixed_gpu.restype = ctypes.c_void_p
        lib.xhe_synth_mixed_gpu.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_int32]
        lib.xhe_synth_free.argtypes = [ctypes.c_void_p]
        lib.xhe_synth_records.restype = ctypes.c_void_p
        lib.xhe_synth_records.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        lib.xhe_synth_offsets.restype = ctypes.POINTER(ctypes.c_uint64)
        lib.xhe_synth_offsets.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        lib.xhe_synth_ledger.restype = ctypes.POINTER(LedgerView)
        lib.xhe_synth_ledger.argtypes = [ctypes.c_void_p]
        lib.xhe_synth_secrets.restype = ctypes.c_void_p
        lib.xhe_synth_secrets.argtypes = [ctypes.c_void_p]
        lib._builder_ready = True
    return lib


class Keypair:
    """ElGamalKeypair (src/elgamal.rs:152-218)."""

    def __init__(self, secret: bytes):
        pub = (ctypes.c_uint8 * 32)()
        if _lib().xhe_keygen(secret, pub) != 0:
            raise XheError("invalid secret scalar")
        self.secret, self.pubkey = bytes(secret), bytes(pub)

    @staticmethod
    def keygen(seed=None):
        sec, pub = (ctypes.c_uint8 * 32)(), (ctypes.c_uint8 * 32)()
        _lib().xhe_keygen_from_seed(_fresh_seed(seed), sec, pub)
        return Keypair(bytes(sec))

    def sign(self, message: bytes, seed=None) -> bytes:
        sig = (ctypes.c_uint8 * 64)()
        _lib().xhe_sign(self.secret, message, len(message), _fresh_seed(seed), sig)
        return bytes(sig)

    def encrypt(self, amount: int, seed=None) -> bytes:
        ct = (ctypes.c_uint8 * 64)()
        _lib().xhe_encrypt(self.pubkey, amount, _fresh_seed(seed), ct)
        return bytes(ct)

    def decrypts_to(self, ct: bytes, amount: int) -> bool:
        return _lib().xhe_ct_decrypts_to(self.secret, ct, amount) == 1

    def decrypt(self, ct: bytes) -> "ECDLPInstance":
        """ElGamalSecretKey::decrypt (src/elgamal.rs:140-145): the point amount*G, still to be decoded."""
        out = (ctypes.c_uint8 * 32)()
        if _lib().xhe_decrypt_to_point(self.secret, ct, out) != 0:
            raise XheError("ciphertext does not decode")
        return ECDLPInstance(bytes(out))


class ECDLPInstance:
    """src/elgamal.rs:68-92: a decrypted point; decode() finds the integer with the GPU's baby-step table."""

    def __init__(self, point32: bytes):
        self.point = bytes(point32)

    def as_point(self) -> bytes:
        return self.point

    def decode(self, tables, max_bits=48):
        """tables: xelis_he_amd.EcdlpTable.  The amount, or None (the reference's Option<i64>)."""
        return tables.decode(self.point, max_bits)[0]


class Account:
    """mock::Account (src/lib.rs:221-241): a keypair, encrypted balances per asset, a nonce."""

    def __init__(self, balances, seed=None):
        self.keypair = Keypair.keygen(seed)
        self.balances = {bytes(a): self.keypair.encrypt(v, None if seed is None else (str(seed) + a.hex())) for a, v in balances}
        self.nonce = 0

    @property
    def pubkey(self):
        return self.keypair.pubkey


class Ledger:
    """mock::Ledger (src/lib.rs:106-201), flattened on demand into the C ABI's ledger snapshot."""

    def __init__(self, accounts):
        self.accounts = {a.pubkey: a for a in accounts}
        self.multisig_accounts = {}

    def clone(self):
        import copy
        out = Ledger([])
        out.accounts = {k: copy.copy(v) for k, v in self.accounts.items()}
        for v in out.accounts.values():
            v.balances = dict(v.balances)
        out.multisig_accounts = {k: (list(s), t) for k, (s, t) in self.multisig_accounts.items()}
        return out

    def set_multisig_for_account(self, account, signers, threshold):
        if not signers:
            self.multisig_accounts.pop(account, None)
        else:
            self.multisig_accounts[account] = (list(signers), threshold)

    def get_multisig_for_account(self, account):
        return self.multisig_accounts.get(account)

    def get_account(self, key):
        return self.accounts[key]

    def balance_decrypts_to(self, account, asset, amount):
        a = self.accounts[account]
        return a.keypair.decrypts_to(a.balances[asset], amount)

    # ---- flattening ----
    def flatten(self):
        keys = list(self.accounts.keys())
        index = {k: i for i, k in enumerate(keys)}
        slots = [(index[k], asset) for k in keys for asset in self.accounts[k].balances]
        f = {"keys": keys, "slots": slots}
        f["account_keys"] = b"".join(keys)
        f["nonces"] = (ctypes.c_uint64 * max(len(keys), 1))(*[self.accounts[k].nonce for k in keys])
        f["balance_account"] = (ctypes.c_uint32 * max(len(slots), 1))(*[s[0] for s in slots])
        f["balance_assets"] = b"".join(s[1] for s in slots)
        f["balance_cts"] = b"".join(self.accounts[keys[s[0]]].balances[s[1]] for s in slots)
        ms = [(index[k], v) for k, v in self.multisig_accounts.items() if k in index]
        f["ms_account"] = (ctypes.c_uint32 * max(len(ms), 1))(*[m[0] for m in ms])
        f["ms_threshold"] = bytes(m[1][1] for m in ms)
        offs, signers = [0], b""
        for m in ms:
            signers += b"".join(m[1][0])
            offs.append(offs[-1] + len(m[1][0]))
        f["ms_offsets"] = (ctypes.c_uint32 * len(offs))(*offs)
        f["ms_signers"] = signers
        v = LedgerView()
        v.n_accounts, v.account_keys, v.account_nonces = len(keys), f["account_keys"], f["nonces"]
        v.n_balances, v.balance_account, v.balance_assets, v.balance_cts = len(slots), f["balance_account"], f["balance_assets"], f["balance_cts"]
        v.n_multisig, v.multisig_account, v.multisig_threshold = len(ms), f["ms_account"], f["ms_threshold"]
        v.multisig_offsets, v.multisig_signers = f["ms_offsets"], f["ms_signers"]
        f["view"] = v
        return f

    def absorb(self, flat, out_cts, out_nonces, ms_set_by, txs):
        """Writes a verified batch's resulting state back (the C ABI returns it as flat arrays)."""
        for i, (acc_i, asset) in enumerate(flat["slots"]):
            self.accounts[flat["keys"][acc_i]].balances[asset] = bytes(out_cts[64 * i:64 * i + 64])
        for i, k in enumerate(flat["keys"]):
            self.accounts[k].nonce = int(out_nonces[i])
            if ms_set_by[i] >= 0:
                tx = txs[ms_set_by[i]]
                self.set_multisig_for_account(k, tx.multisig_signers(), tx.multisig_threshold())


class TransferBuilder:
    def __init__(self, dest_pubkey, amount, asset=NATIVE, extra_data=None):
        self.dest_pubkey, self.amount, self.asset, self.extra_data = dest_pubkey, amount, asset, extra_data


class Transaction:
    """A built transaction: the flat record of include/xhe.h."""

    def __init__(self, record: bytes):
        self.record = bytes(record)

    # header accessors
    @property
    def type(self):
        return self.record[0]

    def _pre_off(self):
        n_items = int.from_bytes(self.record[2:4], "little")
        return 16 + (4 * n_items if self.type == codes.TX_TRANSFERS else 0)

    def multisig_threshold(self):
        return self.record[self._pre_off() + 49]

    def multisig_signers(self):
        n = int.from_bytes(self.record[2:4], "little")
        o = self._pre_off() + 50
        return [self.record[o + 32 * i:o + 32 * i + 32] for i in range(n)]

    def hash(self) -> bytes:
        h = (ctypes.c_uint8 * 32)()
        if _lib().xhe_tx_hash(self.record, len(self.record), h) != 0:
            raise XheError("malformed record")
        return bytes(h)

    def apply_without_verify(self, ledger: Ledger):
        """Transaction::apply_without_verify (verify.rs:545-619), CPU, for the prover's copy of the ledger."""
        f = ledger.flatten()
        cts = ctypes.create_string_buffer(f["balance_cts"], len(f["balance_cts"]))
        rc = _lib().xhe_tx_apply_host(self.record, len(self.record), len(f["keys"]), f["account_keys"], len(f["slots"]),
                                      f["balance_account"], f["balance_assets"], cts)
        if rc != 0:
            raise XheError("apply_without_verify: unknown account/asset or ill-formed ciphertext")
        for i, (acc_i, asset) in enumerate(f["slots"]):
            ledger.accounts[f["keys"][acc_i]].balances[asset] = cts.raw[64 * i:64 * i + 64]
        if self.type == codes.TX_MULTISIG:
            src = self.record[self._pre_off() + 1:self._pre_off() + 33]
            ledger.set_multisig_for_account(src, self.multisig_signers(), self.multisig_threshold())

    @staticmethod
    def verify_batch(engine, txs, ledger: Ledger, seed=None):
        """Transaction::verify_batch (verify.rs:487-517) on the GPU.  Mutates `ledger` on success, raises
        VerificationError otherwise."""
        f = ledger.flatten()
        blob = b"".join(t.record for t in txs)
        offs = [0]
        for t in txs:
            offs.append(offs[-1] + len(t.record))
        offsets = (ctypes.c_uint64 * len(offs))(*offs)
        out_cts = (ctypes.c_uint8 * max(64 * len(f["slots"]), 1))()
        out_nonces = (ctypes.c_uint64 * max(len(f["keys"]), 1))()
        ms_set = (ctypes.c_int64 * max(len(f["keys"]), 1))()
        out = LedgerOut(out_cts, out_nonces, ms_set)
        v = Verdict()
        rc = _lib().xhe_verify_batch(engine.ctx, blob, offsets, len(txs), ctypes.byref(f["view"]), ctypes.byref(out), ctypes.byref(v), _fresh_seed(seed))
        engine._check(rc)
        if not v.accept:
            raise VerificationError(v.err_code, v.first_bad_tx)
        ledger.absorb(f, bytes(out_cts), list(out_nonces), list(ms_set), txs)

    @staticmethod
    def apply_batch(engine, txs, ledger: Ledger):
        """Transaction::apply_without_verify (verify.rs:545-619) for every transaction in order, on the GPU (the ledger
        scan alone).  Mutates `ledger`; raises VerificationError for an unknown account/asset or an ill-formed ciphertext."""
        lib = _lib()
        f = ledger.flatten()
        blob = b"".join(t.record for t in txs)
        offs = [0]
        for t in txs:
            offs.append(offs[-1] + len(t.record))
        offsets = (ctypes.c_uint64 * len(offs))(*offs)
        handle = ctypes.c_void_p()
        v = Verdict()
        lib.xhe_batch_upload.restype = ctypes.c_int32
        lib.xhe_batch_apply.restype = ctypes.c_int32
        lib.xhe_batch_read_ledger.restype = ctypes.c_int32
        ctx = ctypes.c_void_p(engine.ctx)
        engine._check(lib.xhe_batch_upload(ctx, blob, offsets, ctypes.c_uint64(len(txs)), ctypes.byref(f["view"]), ctypes.byref(handle), ctypes.byref(v)))
        try:
            if v.accept:
                engine._check(lib.xhe_batch_apply(ctx, handle, ctypes.byref(v)))
            if not v.accept:
                raise VerificationError(v.err_code, v.first_bad_tx)
            out_cts = (ctypes.c_uint8 * max(64 * len(f["slots"]), 1))()
            out_nonces = (ctypes.c_uint64 * max(len(f["keys"]), 1))()
            ms_set = (ctypes.c_int64 * max(len(f["keys"]), 1))()
            out = LedgerOut(out_cts, out_nonces, ms_set)
            engine._check(lib.xhe_batch_read_ledger(ctx, handle, ctypes.byref(out)))
        finally:
            lib.xhe_batch_free(ctx, handle)
        ledger.absorb(f, bytes(out_cts), list(out_nonces), list(ms_set), txs)

    def verify(self, engine, ledger: Ledger, seed=None):
        """Transaction::verify (verify.rs:520-542): a batch of one."""
        Transaction.verify_batch(engine, [self], ledger, seed)


class TransactionUnsigned:
    """TransactionUnsigned (builder.rs:110-219)."""

    def __init__(self, record: bytes):
        self.record = bytes(record)
        self.multisig = None

    def hash(self) -> bytes:
        return Transaction(self.record).hash()

    def set_multisig(self, entries):
        self.multisig = list(entries)

    def sign(self, keypair: Keypair, seed=None) -> Transaction:
        ms = self.multisig
        ids = bytes(e[0] for e in ms) if ms else b""
        sigs = b"".join(e[1] for e in ms) if ms else b""
        out = (ctypes.c_uint8 * (len(self.record) + 65 * (len(ms) if ms else 0) + 64))()
        n = ctypes.c_uint64()
        rc = _lib().xhe_tx_finalize(self.record, ctypes.c_uint64(len(self.record)), 1 if ms is not None else 0, len(ms) if ms else 0, ids, sigs,
                                    keypair.secret, _fresh_seed(seed), out, ctypes.c_uint64(len(out)), ctypes.byref(n))
        if rc != 0:
            raise XheError("xhe_tx_finalize failed: %d" % rc)
        return Transaction(bytes(out)[:n.value])


class TransactionBuilder:
    """TransactionBuilder (builder.rs:77-84, 221-554).  `data` is one of
    ("transfers", [TransferBuilder...]) / ("burn", asset, amount) / ("multisig", signers, threshold) /
    ("call_contract", contract, [(asset, amount)...], params_bytes) / ("deploy_contract", code_bytes)."""

    def __init__(self, version, source, data, fee, nonce):
        self.version, self.source, self.data, self.fee, self.nonce = version, source, data, fee, nonce

    def get_transaction_cost(self, asset):
        cost = self.fee if asset == NATIVE else 0
        kind = self.data[0]
        if kind == "transfers":
            cost += sum(t.amount for t in self.data[1] if t.asset == asset)
        elif kind == "burn" and self.data[1] == asset:
            cost += self.data[2]
        elif kind == "call_contract":
            cost += sum(a for (x, a) in self.data[2] if x == asset)
        return cost

    def used_assets(self):
        s = [NATIVE]
        kind = self.data[0]
        cand = []
        if kind == "transfers":
            cand = [t.asset for t in self.data[1]]
        elif kind == "burn":
            cand = [self.data[1]]
        elif kind == "call_contract":
            cand = [x for (x, _) in self.data[2]]
        for a in cand:
            if a not in s:
                s.append(a)
        return s

    def build_unsigned(self, balances, keypair: Keypair, seed=None, account=None):
        """balances: {asset: plaintext balance}; account: the source's Account (its ciphertexts), as the
        reference's GenerationBalance pairs them (src/lib.rs:203-219)."""
        rq = TxRequest()
        rq.version, rq.fee, rq.nonce = self.version, self.fee, self.nonce
        kind = self.data[0]
        keep = []
        if kind == "transfers":
            ts = self.data[1]
            rq.type, rq.n_transfers = codes.TX_TRANSFERS, len(ts)
            rq.transfer_assets = b"".join(t.asset for t in ts)
            amounts = (ctypes.c_uint64 * max(len(ts), 1))(*[t.amount for t in ts]); keep.append(amounts)
            rq.transfer_amounts = amounts
            rq.transfer_dests = b"".join(t.dest_pubkey for t in ts)
            if any(t.extra_data for t in ts):
                ex = (ctypes.c_char_p * len(ts))(*[(t.extra_data or b"") for t in ts]); keep.append(ex)
                exl = (ctypes.c_uint32 * len(ts))(*[len(t.extra_data or b"") for t in ts]); keep.append(exl)
                rq.transfer_extra, rq.transfer_extra_len = ex, exl
        elif kind == "burn":
            rq.type = codes.TX_BURN
            rq.burn_asset = (ctypes.c_uint8 * 32)(*self.data[1]); rq.burn_amount = self.data[2]
        elif kind == "multisig":
            rq.type, rq.n_signers, rq.threshold = codes.TX_MULTISIG, len(self.data[1]), self.data[2]
            rq.signers = b"".join(self.data[1])
        elif kind == "call_contract":
            rq.type = codes.TX_CALL_CONTRACT
            rq.contract = (ctypes.c_uint8 * 32)(*self.data[1])
            rq.n_contract_assets = len(self.data[2])
            rq.contract_assets = b"".join(x for (x, _) in self.data[2])
            ca = (ctypes.c_uint64 * max(len(self.data[2]), 1))(*[a for (_, a) in self.data[2]]); keep.append(ca)
            rq.contract_amounts = ca
            rq.opaque, rq.opaque_len = self.data[3], len(self.data[3])
        elif kind == "deploy_contract":
            rq.type = codes.TX_DEPLOY_CONTRACT
            rq.opaque, rq.opaque_len = self.data[1], len(self.data[1])
        else:
            raise ValueError(kind)
        assets = list(balances.keys())
        rq.n_balances = len(assets)
        rq.balance_assets = b"".join(assets)
        ba = (ctypes.c_uint64 * max(len(assets), 1))(*[balances[a] for a in assets]); keep.append(ba)
        rq.balance_amounts = ba
        rq.balance_cts = b"".join(account.balances[a] for a in assets)
        cap = 1 << 20
        out = (ctypes.c_uint8 * cap)()
        n = ctypes.c_uint64()
        rc = _lib().xhe_tx_build_unsigned(ctypes.byref(rq), keypair.secret, _fresh_seed(seed), out, ctypes.c_uint64(cap), ctypes.byref(n), None, None)
        if rc != 0:
            raise XheError({101: "Decompression", 102: "InsufficientFunds", 103: "Format", 104: "State"}.get(rc, "build error %d" % rc))
        return TransactionUnsigned(bytes(out)[:n.value])

    def build(self, balances, keypair: Keypair, seed=None, account=None) -> Transaction:
        u = self.build_unsigned(balances, keypair, seed, account)
        return u.sign(keypair, None if seed is None else str(seed) + "/sig")


class SyntheticBatch:
    """benches/tx.rs:129-189 generalised: n_txs transfer transactions over n_senders sender/receiver pairs."""

    def __init__(self, n_txs, n_transfers=1, n_senders=1, seed=b"synthetic", threads=None, mixed=False, engine=None):
        """mixed=True: n_txs / 16 groups of mixed transactions (multisig set-ups, one- and two-asset transfers, burns,
        co-signed transfers) over disjoint account groups; n_transfers and n_senders are ignored.
        engine: an Engine whose GPU generates the aggregated range proofs (the records are the same bytes as without it)."""
        lib = _lib()
        threads = threads or min(os.cpu_count() or 1, 16)
        if mixed:
            if n_txs % 16:
                raise ValueError("a mixed batch is made of groups of 16 transactions")
            self.handle = (lib.xhe_synth_mixed_gpu(engine.ctx, n_txs // 16, _fresh_seed(seed), threads) if engine is not None
                           else lib.xhe_synth_mixed(n_txs // 16, _fresh_seed(seed), threads))
        elif engine is not None:
            self.handle = lib.xhe_synth_transfers_gpu(engine.ctx, n_txs, n_transfers, n_senders, _fresh_seed(seed), threads)
        else:
            self.handle = lib.xhe_synth_transfers(n_txs, n_transfers, n_senders, _fresh_seed(seed), threads)
        if not self.handle:
            raise XheError("synthetic batch generation failed")
        n = ctypes.c_uint64()
        p = lib.xhe_synth_records(self.handle, ctypes.byref(n))
        self.records = ctypes.string_at(p, n.value)
        k = ctypes.c_uint64()
        offs = lib.xhe_synth_offsets(self.handle, ctypes.byref(k))
        self.n_txs = k.value
        self.offsets = (ctypes.c_uint64 * (self.n_txs + 1))(*[offs[i] for i in range(self.n_txs + 1)])
        self.view = lib.xhe_synth_ledger(self.handle)     # pointer into the handle's storage
        self.n_transfers, self.n_senders = n_transfers, n_senders

    def close(self):
        if self.handle:
            _lib().xhe_synth_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
