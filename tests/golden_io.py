"""Loads the committed fixtures of tests/golden/ (see tests/golden/make_golden.py for what they are)."""
import ctypes
import json
import os

from xelis_he_amd.api import LedgerOut, LedgerView, Verdict

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def primitives():
    with open(os.path.join(HERE, "primitives.json")) as f:
        return json.load(f)


def tx_cases():
    with open(os.path.join(HERE, "tx_cases.json")) as f:
        return json.load(f)["cases"]


class FlatCase:
    """One fixture turned into the C ABI's arguments (keeps every buffer alive)."""

    def __init__(self, case):
        lg = case["ledger"]
        self.name, self.expect = case["name"], case["expect"]
        self.records = [bytes.fromhex(r) for r in case["records"]]
        self.blob = b"".join(self.records)
        offs = [0]
        for r in self.records:
            offs.append(offs[-1] + len(r))
        self.offsets = (ctypes.c_uint64 * len(offs))(*offs)
        self.seed = bytes.fromhex(case["seed"])
        self.keys = bytes.fromhex(lg["account_keys"])
        n_acc, n_bal = len(self.keys) // 32, len(lg["balance_account"])
        self.nonces = (ctypes.c_uint64 * max(n_acc, 1))(*lg["nonces"])
        self.balance_account = (ctypes.c_uint32 * max(n_bal, 1))(*lg["balance_account"])
        self.balance_assets = bytes.fromhex(lg["balance_assets"])
        self.balance_cts = bytes.fromhex(lg["balance_cts"])
        self.ms_account = (ctypes.c_uint32 * max(len(lg["ms_account"]), 1))(*lg["ms_account"])
        self.ms_threshold = bytes.fromhex(lg["ms_threshold"])
        self.ms_offsets = (ctypes.c_uint32 * len(lg["ms_offsets"]))(*lg["ms_offsets"])
        self.ms_signers = bytes.fromhex(lg["ms_signers"])
        v = LedgerView()
        v.n_accounts, v.account_keys, v.account_nonces = n_acc, self.keys, self.nonces
        v.n_balances, v.balance_account, v.balance_assets, v.balance_cts = n_bal, self.balance_account, self.balance_assets, self.balance_cts
        v.n_multisig, v.multisig_account, v.multisig_threshold = len(lg["ms_account"]), self.ms_account, self.ms_threshold
        v.multisig_offsets, v.multisig_signers = self.ms_offsets, self.ms_signers
        self.view, self.n_acc, self.n_bal = v, n_acc, n_bal
        self.out_cts = (ctypes.c_uint8 * max(64 * n_bal, 1))()
        self.out_nonces = (ctypes.c_uint64 * max(n_acc, 1))()
        self.ms_set = (ctypes.c_int64 * max(n_acc, 1))()
        self.out = LedgerOut(self.out_cts, self.out_nonces, self.ms_set)
        self.verdict = Verdict()

    def check(self):
        e, v = self.expect, self.verdict
        assert (int(v.accept), int(v.err_code), int(v.first_bad_tx)) == (e["accept"], e["err_code"], e["first_bad_tx"]), self.name
        if e["accept"]:
            assert bytes(self.out_cts)[:64 * self.n_bal].hex() == e["out_cts"], self.name
            assert [int(x) for x in self.ms_set][:self.n_acc] == e["multisig_set_by"], self.name
