#!/usr/bin/env python3
"""Benchmark of the hot path: Transaction::verify_batch on MI355X.

A step is one xhe_verify_batch call on one synthetic batch held in host buffers: the drop-in call of
/root/reference/src/tx/verify.rs:487-490 (parsing, ledger-key resolution, the host-to-device copy, the verification, the
ledger read-back).  The workload is BASELINE.json's configs[2], the largest single-GPU configuration: a 100 000-transaction
batch, 8 transfers each (aggregated 64-bit range proofs over 16 value commitments), single (native) asset, per GPU; its
100 000 range proofs are generated on the GPU before the timed region (xhe_synth_transfers_gpu).  `resident` is the same
verification on a batch that was parsed and uploaded once (xhe_batch_upload, then xhe_batch_verify per step): the GPU work
alone.  After the timed region rank 0 also measures the other single-GPU shapes of BASELINE.json: configs[1] (10 000 x 1
transfer), one GPU's share of configs[3] (125 000 x 1 transfer) and of configs[4] (100 000 mixed transactions), and a run of
batches through xhe_verify_batches.  With --gpus N every rank verifies its own batch (transactions shard by account group, no
data-path collective) and the ranks AND-reduce their accept bits with one RCCL all-reduce (MIN) per step; value =
transactions of all ranks / max-over-ranks time (weak scaling).

Prints one JSON line (rank 0).  roofline: the dominant kernel is the bucket accumulation of the Pippenger multiscalar
check; `achieved` = algorithmic bytes per launch / that kernel's average launch time inside the timed region, measured with
HIP events on the stream each launch ran on (xhe_msm_stats).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak


class MsmStats(ctypes.Structure):
    _fields_ = [("launches", ctypes.c_uint64), ("entries", ctypes.c_uint64), ("terms_nonzero", ctypes.c_uint64), ("accumulate_ms", ctypes.c_double)]


def _profile_csv(kind, tag):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s_%s.csv" % (kind, tag))))
    return files[-1] if files else None


def measured_traffic(tag):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this very workload
    (FETCH_SIZE and WRITE_SIZE need separate passes and cannot be collected from inside this process): (FETCH_SIZE +
    WRITE_SIZE) KiB as reported.  Returns (bytes, fetch bytes, write bytes) or None when no matching summary is committed."""
    import csv
    parts = []
    for counter in ("fetch_size", "write_size"):
        path = _profile_csv(counter, tag)
        if not path:
            return None
        got = None
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["kernel"].endswith("k_msm_accumulate"):
                    got = float(row["avg"]) * 1024.0
        if got is None:
            return None
        parts.append(got)
    return int(parts[0] + parts[1]), int(parts[0]), int(parts[1])


def measured_valu_instructions(tag):
    """Wave-level VALU instructions per launch of the dominant kernel (SQ_INSTS_VALU) from the committed rocprofv3 --pmc
    pass of this very workload; None when no matching summary is committed."""
    import csv
    path = _profile_csv("valu", tag)
    if not path:
        return None
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["kernel"].endswith("k_msm_accumulate") and row["counter"] == "SQ_INSTS_VALU":
                return float(row["avg"])
    return None


def gather_calibration():
    """The committed run of tools/calibrate_gather.sh: `fetch_size_factor` = bytes of the 128-byte lines really fetched per byte that
    FETCH_SIZE reports, for the accumulation kernel's access shape (six 16-byte LDS-DMA loads per lane from a random 128-byte
    record; 1.05 here, against 2.0 for a coalesced streaming read), and `gather_line_GBps`, the rate the memory system sustains
    for that shape on a table beyond the on-die cache; None when not committed."""
    path = os.path.join(ROOT, "profiles", "r03_gather_calibration.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--txs", type=int, default=100000, help="transactions per GPU batch")
    ap.add_argument("--transfers", type=int, default=8)
    ap.add_argument("--value-basis", choices=["end_to_end", "resident"], default="end_to_end",
                    help="what a timed step is: xhe_verify_batch from host buffers (default), or xhe_batch_verify on a batch uploaded before the timed region")
    ap.add_argument("--lean", action="store_true", help="the timed loop only (profiling runs): no resident steps, stage breakdown, CPU baseline, pipelined run or other configurations")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other configurations after the timed region")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the xhe_verify_batches run after the timed region")
    ap.add_argument("--resident-steps", type=int, default=30, help="steps of the resident measurement beside the timed region (0 = skip)")
    ap.add_argument("--workload", choices=["transfers", "mixed"], default="transfers",
                    help="mixed: groups of 16 transactions (multisig set-ups, one- and two-asset transfers, burns, co-signed transfers)")
    ap.add_argument("--senders", type=int, default=1024, help="sender accounts per batch (1 = the reference bench's single chain)")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="transactions of the batch the CPU baseline verifies (0 = skip)")
    ap.add_argument("--gen-threads", type=int, default=16)
    ap.add_argument("--cpu-prover", action="store_true", help="generate the range proofs of the synthetic batch on the CPU (same bytes, slower)")
    ap.add_argument("--backend", default="nccl", help="process-group backend for --gpus > 1 (nccl = RCCL; gloo reduces through host memory, for rehearsals)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal on a one-GPU box: every rank uses device 0 (only with --backend gloo)")
    ap.add_argument("--split-one-batch", action="store_true",
                    help="with --gpus N: rank 0 generates ONE batch of N x --txs transactions, cuts it along its account groups (xhe_partition_batch) and hands "
                         "every rank its shard; a step verifies the shards and MIN-reduces the 8-byte verdict key (the one-batch-over-eight-GPUs shape of the benchmark configurations); "
                         "the default is one independent batch per rank")
    args = ap.parse_args()
    if args.lean:
        args.no_secondary = args.no_pipelined = True
        args.resident_steps = args.cpu_sample = 0
    return args


def term_counts(txs, transfers, mixed):
    """MSM terms of one verification: dynamic terms (sigma proofs, range proofs, value commitments) and static generators."""
    T, A = transfers, 1
    m = 1
    while m < A + T:
        m *= 2
    lg = 6 + (m.bit_length() - 1)
    dyn_terms = txs * (7 * A + 8 * T + 4 + 2 * lg + A + T)
    static_terms = 2 + 2 * 64 * m
    if mixed:
        # per group of 16: 2 set-ups (24 terms), 4 one-asset transfers (35), 4 two-asset double transfers (54), 2 native burns (24),
        # 2 second-asset burns (34), 2 co-signed transfers (35); four value commitments at most -> 256 generators per side
        dyn_terms = (txs // 16) * (2 * 24 + 4 * 35 + 4 * 54 + 2 * 24 + 2 * 34 + 2 * 35)
        static_terms = 2 + 2 * 256
    return dyn_terms, static_terms


def workload_tag(txs, transfers, mixed):
    return "%dtx_mixed" % txs if mixed else "%dtx%s" % (txs, "" if transfers == 1 else "_t%d" % transfers)


class FlatBatch:
    """A batch held as plain buffers (what a rank receives in --split-one-batch mode): the fields Resident and the library read."""

    def __init__(self, records, offsets, ledger):
        from xelis_he_amd.api import LedgerView
        self.records, self.n_txs = records, len(offsets) - 1
        self.offsets = (ctypes.c_uint64 * len(offsets))(*offsets)
        self._keep = ledger
        n_acc, n_bal = len(ledger["keys"]) // 32, len(ledger["assets"]) // 32
        self._nonces = (ctypes.c_uint64 * max(n_acc, 1))()
        self._bal_acc = (ctypes.c_uint32 * max(n_bal, 1))(*ledger["balance_account"])
        v = LedgerView()
        v.n_accounts, v.account_keys, v.account_nonces = n_acc, ledger["keys"], self._nonces
        v.n_balances, v.balance_account, v.balance_assets, v.balance_cts = n_bal, self._bal_acc, ledger["assets"], ledger["cts"]
        self._view = v
        self.view = ctypes.pointer(v)

    def close(self):
        pass


def split_one_batch(args, dist, rank, world, engine):
    """Rank 0 generates the whole batch, cuts it (xhe_partition_batch, in the library) and scatters the shards as byte tensors;
    returns this rank's FlatBatch and its transactions' global indices."""
    import numpy as np
    import torch
    from xelis_he_amd import sharding
    from xelis_he_amd.api import SyntheticBatch
    payload = None
    if rank == 0:
        big = SyntheticBatch(args.txs * world, args.transfers, min(args.senders * world, args.txs * world), seed=b"bench-split", threads=args.gen_threads,
                             engine=None if args.cpu_prover else engine)
        payload = sharding.cut_for_ranks(big, world)      # per rank: (records, offsets, global indices, ledger arrays), all numpy uint8/uint64/uint32
        big.close()
    # sizes first (one small object list), then the raw bytes of every array with torch.distributed.scatter (no pickling of the data)
    names = ("records", "offsets", "index", "keys", "assets", "cts", "balance_account")
    sizes = [None]
    dist.scatter_object_list(sizes, [[int(p[n].nbytes) for n in names] for p in payload] if rank == 0 else None, src=0)
    got = {}
    for j, n in enumerate(names):
        width = max(sizes[0][j], 1)
        longest = torch.tensor([width], dtype=torch.int64)
        dist.all_reduce(longest, op=dist.ReduceOp.MAX)
        L = int(longest.item())
        mine = torch.zeros(L, dtype=torch.uint8)
        chunks = None
        if rank == 0:
            chunks = []
            for p in payload:
                t = torch.zeros(L, dtype=torch.uint8)
                raw = np.frombuffer(p[n].tobytes(), dtype=np.uint8)
                t[:raw.size] = torch.from_numpy(raw.copy())
                chunks.append(t)
        if dist.get_backend() == "nccl":
            dev = torch.device("cuda", torch.cuda.current_device())
            mine = mine.to(dev)
            chunks = [c.to(dev) for c in chunks] if chunks else None
        dist.scatter(mine, chunks, src=0)
        got[n] = mine.cpu().numpy()[:sizes[0][j]].tobytes()
    offsets = list(np.frombuffer(got["offsets"], dtype=np.uint64))
    index = [int(x) for x in np.frombuffer(got["index"], dtype=np.uint64)]
    ledger = {"keys": got["keys"], "assets": got["assets"], "cts": got["cts"], "balance_account": [int(x) for x in np.frombuffer(got["balance_account"], dtype=np.uint32)]}
    return FlatBatch(got["records"], [int(x) for x in offsets], ledger), index


class HostCall:
    """xhe_verify_batch from host buffers: the drop-in call."""

    def __init__(self, engine, lib, batch):
        from xelis_he_amd.api import LedgerOut, Verdict
        self.engine, self.lib, self.batch = engine, lib, batch
        view = batch.view.contents
        self.out_cts = (ctypes.c_uint8 * max(64 * view.n_balances, 1))()
        self.out_nonces = (ctypes.c_uint64 * max(view.n_accounts, 1))()
        self.ms_set = (ctypes.c_int64 * max(view.n_accounts, 1))()
        self.out = LedgerOut(self.out_cts, self.out_nonces, self.ms_set)
        self.verdict = Verdict()

    def step(self, i):
        b = self.batch
        self.engine._check(self.lib.xhe_verify_batch(ctypes.c_void_p(self.engine.ctx), b.records, b.offsets, ctypes.c_uint64(b.n_txs), b.view, ctypes.byref(self.out),
                                                     ctypes.byref(self.verdict), (i + 1).to_bytes(8, "little") * 4))
        return self.verdict.accept

    def free(self):
        pass


class Resident:
    """One synthetic batch parsed and uploaded once (xhe_batch_upload); step() = xhe_batch_verify on it."""

    def __init__(self, engine, lib, batch):
        from xelis_he_amd.api import Verdict
        self.engine, self.lib, self.batch = engine, lib, batch
        self.handle = ctypes.c_void_p()
        sv = Verdict()
        t0 = time.time()
        engine._check(lib.xhe_batch_upload(ctypes.c_void_p(engine.ctx), batch.records, batch.offsets, ctypes.c_uint64(batch.n_txs), batch.view,
                                           ctypes.byref(self.handle), ctypes.byref(sv)))
        self.upload_s = time.time() - t0
        assert sv.accept == 1, "synthetic batch is malformed"
        self.verdict = Verdict()

    def step(self, i):
        seed = (i + 1).to_bytes(8, "little") * 4
        self.engine._check(self.lib.xhe_batch_verify(ctypes.c_void_p(self.engine.ctx), self.handle, ctypes.byref(self.verdict), seed))
        return self.verdict.accept

    def free(self):
        self.lib.xhe_batch_free(ctypes.c_void_p(self.engine.ctx), self.handle)


def read_stats(engine, lib, reset=True):
    s = MsmStats()
    engine._check(lib.xhe_msm_stats_read(ctypes.c_void_p(engine.ctx), ctypes.byref(s), 1 if reset else 0))
    return s


def timed_steps(runner, steps, warmup, first, sync):
    """warmup untimed steps, then `steps` timed ones between two synchronisations; returns seconds."""
    for i in range(warmup):
        assert runner.step(first + i) == 1, "verification rejected a valid batch"
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        assert runner.step(first + warmup + i) == 1, "verification rejected a valid batch"
    sync()
    return time.perf_counter() - t0


def roofline_of(stats, steps, txs, transfers, mixed):
    """The bucket accumulation kernel against the HBM roofline, from the counters of `steps` verifications."""
    if not stats.launches or stats.accumulate_ms <= 0:
        return None
    dyn_terms, static_terms = term_counts(txs, transfers, mixed)
    # algorithmic bytes of the multiscalar check: per term a 32-byte scalar + a 128-byte decoded point record, 32 bytes of result
    algo_step = (dyn_terms + static_terms) * 160 + 32
    launches_per_step = stats.launches / steps
    kernel_ms = stats.accumulate_ms / stats.launches
    algo_launch = algo_step / launches_per_step
    achieved = algo_launch / (kernel_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": None,
         "kernel": "k_msm_accumulate", "kernel_ms": round(kernel_ms, 4), "launches_per_step": round(launches_per_step, 2), "algorithmic_bytes": int(algo_launch),
         "terms": int((dyn_terms + static_terms) / launches_per_step), "terms_accumulated": int(stats.terms_nonzero / stats.launches),
         "entries_accumulated": int(stats.entries / stats.launches),
         "terms_note": "terms: the contractual count (every sigma-proof, range-proof, value-commitment and static-generator term) per launch; terms_accumulated: terms with a "
                       "non-zero scalar that entered the bucket sort after terms on one point were merged and the static generators were taken from fixed-base tables "
                       "(read back from the device); entries_accumulated: their non-zero digits over all windows = point records gathered"}
    return r


def attach_traffic(roofline, tag, launches_per_step_in_profile=None):
    """HBM-side bytes per launch from the committed --pmc passes of this workload, with the calibrated FETCH_SIZE factor."""
    if roofline is None:
        return
    t = measured_traffic(tag)
    cal = gather_calibration()
    if t is None:
        roofline["traffic_source"] = "unmeasured for this workload (no committed --pmc pass named *_%s.csv)" % tag
        return
    total, fetch, write = t
    corr = float(cal["fetch_size_factor"]) if cal else 1.0
    roofline["traffic"] = int(fetch * corr + write)
    roofline["traffic_correction"] = {"fetch_size_factor": corr, "fetch_size_reported": fetch, "write_size_reported": write,
                                      "source": ("profiles/r03_gather_calibration.json: bytes of the lines fetched per byte FETCH_SIZE reports, for this kernel's access shape "
                                                 "(six 16-byte LDS-DMA loads per lane from a random 128-byte record)") if cal else "no calibration committed: counters as reported"}
    roofline["traffic_source"] = "FETCH_SIZE x factor + WRITE_SIZE of the kernel per launch, newest committed rocprofv3 --pmc passes of this workload under profiles/"
    if cal and cal.get("gather_line_GBps") and roofline.get("kernel_ms"):
        # the bytes the kernel really moves (one point record per entry and window, not one per term) against the rate measured for
        # exactly this access shape: how close the kernel is to the memory system, beside the contractual HBM line above
        rate = roofline["traffic"] / (roofline["kernel_ms"] * 1e-3) / 1e9
        roofline["gather"] = {"achieved": round(rate, 1), "peak": float(cal["gather_line_GBps"]), "unit": "GB/s", "frac": round(rate / float(cal["gather_line_GBps"]), 4),
                              "what": "measured traffic per launch / launch time against the random 128-byte-record gather rate of tools/ubench_gather (table of 2 GiB, every record once)"}


def valu_issue_of(roofline, tag, torch, local_rank):
    """The kernel is bound by VALU issue, not by HBM: wave-instructions per launch (PMC) / launch time against the chip's vector
    issue rate, one wave-instruction per 2 cycles per SIMD (a 64-wide wavefront on a SIMD-32 takes two passes).  The kernel's
    mix is mostly v_mad_u64_u32, which issues in 5.5 cycles (tools/ubench_valu.hip, 8 waves per SIMD): `mix_bound_frac` prices
    the measured instruction count at the mix-weighted cost of profiles/r02_accumulate_mix.json when that file is committed."""
    n_valu = measured_valu_instructions(tag)
    if not n_valu or roofline is None:
        return None
    prop = torch.cuda.get_device_properties(local_rank)
    clock_hz = getattr(prop, "clock_rate", 2400000) * 1e3
    simds = prop.multi_processor_count * 4
    peak = simds * clock_hz / 2.0
    rate = n_valu / (roofline["kernel_ms"] * 1e-3)
    valu = {"kernel": "k_msm_accumulate", "wave_instructions": int(n_valu), "achieved": round(rate / 1e9, 2), "peak": round(peak / 1e9, 2),
            "unit": "G wave-instructions/s", "frac": round(rate / peak, 4), "cycles_per_wave_instruction_at_peak": 2, "simds": simds, "clock_mhz": round(clock_hz / 1e6),
            "source": "SQ_INSTS_VALU per launch from the newest committed --pmc pass of this workload / the launch time measured in this run"}
    mix_path = os.path.join(ROOT, "profiles", "r02_accumulate_mix.json")
    if os.path.exists(mix_path):
        with open(mix_path) as f:
            mix = json.load(f)
        cyc = float(mix["cycles_per_wave_instruction"])
        valu["mix_cycles_per_wave_instruction"] = cyc
        valu["mix_bound_frac"] = round(rate / (simds * clock_hz / cyc), 4)
    return valu


def measure_config(engine, lib, torch, batch, txs, transfers, mixed, label, steps=20, warmup=5):
    """A configuration beside the headline one: the call from host buffers and the resident verification, with the kernel's roofline."""
    sync = torch.cuda.synchronize
    host = HostCall(engine, lib, batch)
    read_stats(engine, lib)
    e = timed_steps(host, steps, warmup, 5000, sync)
    read_stats(engine, lib)
    res = Resident(engine, lib, batch)
    read_stats(engine, lib)
    r = timed_steps(res, steps, warmup, 6000, sync)
    st = read_stats(engine, lib)
    # (the resident run's counters: warm-up steps included in the counters and in the launch count alike)
    roof = roofline_of(st, steps + warmup, txs, transfers, mixed)
    attach_traffic(roof, workload_tag(txs, transfers, mixed) + "_resident")
    res.free()
    return {"workload": label, "steps": steps, "warmup": warmup, "ms_per_step": round(e / steps * 1e3, 4), "value": round(txs * steps / e, 1), "unit": "tx/s",
            "resident": {"ms_per_step": round(r / steps * 1e3, 4), "value": round(txs * steps / r, 1)}, "roofline": roof}


def measure_ecdlp(engine, table_bits=24, max_bits=40, n=4096, reps=5):
    """Balance decoding beside the verification (ECDLPInstance::decode, src/elgamal.rs:68-92): n ciphertexts of random amounts below
    2^max_bits under one key, decrypted and decoded on the GPU with a baby-step table of 2^table_bits entries.  A giant step is one
    affine point addition and one 8-byte read at a random place of the table (a 64-byte sector of HBM)."""
    import random
    from xelis_he_amd import EcdlpTable
    from xelis_he_amd.api import Keypair
    rnd = random.Random(5)
    kp = Keypair.keygen(b"bench-ecdlp".ljust(32, b"\0"))
    amounts = [rnd.randrange(1 << max_bits) for _ in range(n)]
    cts = b"".join(kp.encrypt(a, seed=i.to_bytes(4, "little") + bytes(28)) for i, a in enumerate(amounts))
    t0 = time.time()
    table = EcdlpTable(engine, table_bits)
    build_s = time.time() - t0
    try:
        calls, walks, steps = [], [], 0
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            got = table.decrypt_decode(kp.secret, cts, max_bits)
            calls.append(time.perf_counter() - t0)
            st = table.last_stats()
            walks.append(st["walk_ms"]); steps = st["steps"]
            if got != amounts:
                raise SystemExit("balance decoding returned a wrong amount")
        call_s, walk_ms = min(calls[1:]), min(walks[1:])
    finally:
        table.close()
    cal = gather_calibration()
    out = {"workload": "%d ciphertexts of amounts below 2^%d under one key: decrypt + decode, baby-step table of 2^%d entries (%d MiB)" % (n, max_bits, table_bits, (16 << table_bits) >> 20),
           "table_build_s": round(build_s, 3), "ms_per_call": round(call_s * 1e3, 3), "value": round(n / call_s, 1), "unit": "balances/s",
           "giant_steps": steps, "walk_ms": round(walk_ms, 3), "steps_per_s": round(steps / (walk_ms * 1e-3), 1),
           "table_read_GBps_at_64B_sectors": round(steps * 64 / (walk_ms * 1e-3) / 1e9, 1),
           "note": "every amount checked against the encrypted one; the giant steps are found-early-exit rounds over 2^(max_bits - table_bits - 1) steps per instance"}
    if cal and cal.get("kernels", {}).get("gather<4,64>"):
        out["random_64B_gather_peak_GBps"] = cal["kernels"]["gather<4,64>"]["line_GBps"]
    return out


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    dist = None
    rccl_error = None
    if args.one_device:
        if world > 1 and args.backend != "gloo":
            raise SystemExit("--one-device is a rehearsal mode for --backend gloo")
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback for the verification path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import datetime
        import torch.distributed as dist_mod
        dist = dist_mod
        # The library keeps its four kernel streams.  The collective of a step runs between two verifications, never beside one, so
        # RCCL's stream sharing a hardware queue (the process has four) with one of the library's costs nothing.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if args.backend == "nccl":
                dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=120))
                probe = torch.ones(1, dtype=torch.int64, device="cuda")
                dist.all_reduce(probe, op=dist.ReduceOp.MIN)          # the first collective builds the communicator: fail here, with a message, not inside the timed loop
                torch.cuda.synchronize()
            else:
                dist.init_process_group(backend=args.backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
        except Exception as e:      # an RCCL start-up failure: say so in the line instead of hanging to the driver's limit
            rccl_error = "%s: %s" % (type(e).__name__, str(e)[:400])
            if rank == 0:
                print(json.dumps({"metric": "verified tx/sec (batched)", "value": None, "unit": "tx/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                                  "higher_is_better": True, "scaling": "weak", "rccl_error": rccl_error}), flush=True)
            sys.exit(3)

    if world > 1 and "XHE_INGEST_THREADS" not in os.environ:
        # the ranks share the node's host cores: the uploads' parsing threads are divided among them (the library reads its
        # switches when its first context is created, below)
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 16
        os.environ["XHE_INGEST_THREADS"] = str(max(4, min(16, cores // world)))
        args.gen_threads = max(4, min(args.gen_threads, cores // world))
    import xelis_he_amd
    from xelis_he_amd.api import SyntheticBatch, Verdict, _lib
    lib = _lib()
    lib.xhe_batch_upload.restype = ctypes.c_int32
    lib.xhe_batch_verify.restype = ctypes.c_int32
    lib.xhe_msm_stats_enable.restype = lib.xhe_msm_stats_read.restype = ctypes.c_int32
    engine = xelis_he_amd.Engine(local_rank)
    engine._check(lib.xhe_msm_stats_enable(ctypes.c_void_p(engine.ctx), 1))

    t0 = time.time()
    senders = min(args.senders, args.txs)
    mixed = args.workload == "mixed"
    if mixed:
        args.txs -= args.txs % 16
    gen_engine = None if args.cpu_prover else engine
    split = args.split_one_batch and dist is not None and not mixed
    global_index = None
    if split:
        batch, global_index = split_one_batch(args, dist, rank, world, engine)
    else:
        batch = SyntheticBatch(args.txs, args.transfers, senders, seed=b"bench-rank-%d" % rank, threads=args.gen_threads, mixed=mixed, engine=gen_engine)
    gen_s = time.time() - t0
    # committed --pmc summaries are per launch: those of the call from host buffers (a launch per part) and those of the resident
    # verification (one launch over the whole batch) are kept apart by name
    tag = workload_tag(args.txs, args.transfers, mixed) + ("" if args.value_basis == "end_to_end" else "_resident")

    on_host = args.backend != "nccl"              # gloo gets host tensors only
    flag = torch.ones(1, dtype=torch.int64, device="cpu" if (on_host or dist is None) else "cuda")

    def reduce_accept(ok, verdict):
        nonlocal rccl_error
        if dist is None:
            return ok
        try:
            if split:
                # one batch over all ranks: the MIN of the shards' 8-byte verdict keys is the batch's verdict (accept = the largest key)
                from xelis_he_amd import sharding
                flag.fill_(sharding.verdict_key(bool(ok), verdict.err_code, verdict.first_bad_tx, global_index))
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return 1 if int(flag.item()) == sharding.ACCEPT_KEY else 0
            flag.fill_(ok)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)       # AND of the per-shard accept bits
            return int(flag.item())
        except Exception as e:
            rccl_error = "%s: %s" % (type(e).__name__, str(e)[:400])
            if rank == 0:
                print(json.dumps({"metric": "verified tx/sec (batched)", "value": None, "unit": "tx/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                                  "higher_is_better": True, "scaling": "weak", "rccl_error": rccl_error}), flush=True)
            os._exit(3)

    class Reduced:
        def __init__(self, inner):
            self.inner = inner

        def step(self, i):
            return reduce_accept(self.inner.step(i), self.inner.verdict)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    engine.set_stage_timing(1)
    host_call = HostCall(engine, lib, batch)
    resident = Resident(engine, lib, batch)
    timed = host_call if args.value_basis == "end_to_end" else resident
    other = resident if args.value_basis == "end_to_end" else host_call

    # ---- the timed region ----
    for i in range(args.warmup):
        assert Reduced(timed).step(i) == 1, "verification rejected a valid batch"
    read_stats(engine, lib)
    stage_sum = {}
    barrier()
    t0 = time.perf_counter()
    runner = Reduced(timed)
    for i in range(args.steps):
        assert runner.step(args.warmup + i) == 1, "verification rejected a valid batch"
        if timed is resident:
            for name, ms in engine.stage_times():
                stage_sum[name] = stage_sum.get(name, 0.0) + ms
    barrier()
    elapsed = time.perf_counter() - t0
    stats = read_stats(engine, lib)
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_host else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    roofline = roofline_of(stats, args.steps, args.txs, args.transfers, mixed)
    attach_traffic(roofline, tag)
    valu = valu_issue_of(roofline, tag, torch, local_rank)

    # ---- beside it: the other basis, same batch, same process (every rank runs it; rank 0 reports its own) ----
    beside = None
    stages, detail = {}, {}
    if args.resident_steps > 0:
        read_stats(engine, lib)
        sec = timed_steps(other, args.resident_steps, 5, 100000, torch.cuda.synchronize)
        ostats = read_stats(engine, lib)
        oroof = roofline_of(ostats, args.resident_steps + 5, args.txs, args.transfers, mixed)
        beside = {"what": ("xhe_batch_verify on the batch parsed and uploaded once (xhe_batch_upload before): the GPU work of a verification alone" if other is resident
                           else "xhe_verify_batch from host buffers (parse + ledger keys + H2D + verify + ledger read-back)"),
                  "steps": args.resident_steps, "warmup": 5, "ms_per_step": round(sec / args.resident_steps * 1e3, 4), "value": round(args.txs * args.resident_steps / sec, 1), "unit": "tx/s",
                  "kernel_ms": oroof["kernel_ms"] if oroof else None, "roofline_frac": oroof["frac"] if oroof else None,
                  "launches_per_step": oroof["launches_per_step"] if oroof else None}
    if rank == 0 and not args.lean:
        # per-stage breakdown of the resident verification: HIP events at the stage boundaries, outside every timed loop
        engine.set_stage_timing(1)
        for i in range(5):
            assert resident.step(200000 + i) == 1
            for name, ms in engine.stage_times():
                stages[name] = stages.get(name, 0.0) + ms / 5
        engine.set_stage_timing(2)
        for i in range(3):
            assert resident.step(200100 + i) == 1
            for name, ms in engine.stage_times():
                detail[name] = detail.get(name, 0.0) + ms / 3
        engine.set_stage_timing(0)

    cpu_baseline = None
    if rank == 0 and args.gpus == 1 and args.cpu_sample > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import helpers
        oracle = helpers.load_oracle()
        n_s = min(args.cpu_sample, batch.n_txs)
        cores = min(os.cpu_count() or 1, 16)
        ok = ctypes.c_int32()
        sec = oracle.xo_bench_verify_batch(batch.records, batch.offsets, ctypes.c_uint64(n_s), batch.view, cores, ctypes.byref(ok))
        if ok.value == 1 and sec > 0:
            one = oracle.xo_bench_verify_batch(batch.records, batch.offsets, ctypes.c_uint64(min(n_s, 256)), batch.view, 1, ctypes.byref(ok))
            cpu_baseline = {"value": round(cores * n_s / sec, 1), "unit": "tx/s", "cores": cores, "kind": "port",
                            "sample": "first %d transactions of the batch, one ledger clone per thread, %d threads (benches/tx.rs:307-325 style)" % (n_s, cores),
                            "single_thread_ms_per_tx": round(one / min(n_s, 256) * 1e3, 4), "reference_published_ms_per_tx_1_transfer": 0.40}

    # after the timed region, rank 0 at N = 1: a run of batches through the library's pipeline, and the other single-GPU configurations
    pipelined, configs = None, None
    if rank == 0 and args.gpus == 1:
        engine.set_stage_timing(0)
        if not args.no_pipelined:
            class BatchArgs(ctypes.Structure):
                _fields_ = [("records", ctypes.c_char_p), ("offsets", ctypes.POINTER(ctypes.c_uint64)), ("n_txs", ctypes.c_uint64), ("view", ctypes.c_void_p),
                            ("out", ctypes.c_void_p), ("seed", ctypes.c_char_p)]
            lib.xhe_verify_batches.restype = ctypes.c_int32

            def run_of(n_b):
                arr = (BatchArgs * n_b)()
                seeds = [(2000 + i).to_bytes(8, "little") * 4 for i in range(n_b)]
                for i in range(n_b):
                    arr[i].records, arr[i].offsets, arr[i].n_txs = batch.records, batch.offsets, batch.n_txs
                    arr[i].view, arr[i].out, arr[i].seed = ctypes.cast(batch.view, ctypes.c_void_p), ctypes.cast(ctypes.pointer(host_call.out), ctypes.c_void_p), seeds[i]
                vs = (Verdict * n_b)()
                t0 = time.perf_counter()
                engine._check(lib.xhe_verify_batches(ctypes.c_void_p(engine.ctx), ctypes.c_uint64(n_b), arr, vs))
                dt = time.perf_counter() - t0
                assert all(v_.accept == 1 for v_ in vs)
                return dt

            # a run of 3 and a run of 9: the difference is six batches in the steady state (the first upload of a run overlaps nothing)
            t3, t9 = run_of(3), run_of(9)
            steady = max(t9 - t3, 1e-9) / 6
            pipelined = {"call": "xhe_verify_batches over 9 such batches from host buffers", "ms_per_batch": round(t9 / 9 * 1e3, 3),
                         "value": round(9 * batch.n_txs / t9, 1), "unit": "tx/s", "steady_ms_per_batch": round(steady * 1e3, 3),
                         "steady_value": round(batch.n_txs / steady, 1), "steady_note": "(run of 9 - run of 3) / 6"}
        if not args.no_secondary and not mixed:
            configs = {}
            todo = [("config1_10000x1", 10000, 1, False, "10000-tx batch, 1 transfer each, single asset, 1024 sender accounts (BASELINE configs[1])", 50, 10),
                    ("config3_share_125000x1", 125000, 1, False, "125000-tx batch, 1 transfer each: one GPU's share of the 1 000 000-tx batch over 8 GPUs (BASELINE configs[3])", 20, 5),
                    ("config4_share_mixed_100000", 100000, 1, True, "100000-tx mixed batch (per 16: 2 multisig set-ups, 4 one-asset transfers, 4 two-asset double transfers, 2 + 2 burns, "
                                                                   "2 co-signed transfers; 6 250 account groups): one GPU's share of BASELINE configs[4]", 20, 5)]
            for key, n, t, mx, label, st_, wu_ in todo:
                if (n, t, mx) == (args.txs, args.transfers, mixed):
                    continue
                g0 = time.time()
                b2 = SyntheticBatch(n, t, 1024, seed=("bench-%s" % key).encode(), threads=args.gen_threads, mixed=mx, engine=gen_engine)
                g2 = time.time() - g0
                configs[key] = measure_config(engine, lib, torch, b2, n, t, mx, label, st_, wu_)
                configs[key]["generate_s"] = round(g2, 2)
                b2.close()

    ecdlp = None
    if rank == 0 and dist is None and not args.lean and not args.no_secondary and not mixed:
        ecdlp = measure_ecdlp(engine)
        # the same call with a table sixteen times as large (4 GiB of the 288): a sixteenth of the giant steps
        ecdlp["table_2^28"] = {k: v for k, v in measure_ecdlp(engine, table_bits=28).items() if k not in ("note", "random_64B_gather_peak_GBps")}

    if rank == 0:
        total_txs = args.txs * args.gpus * args.steps
        basis_e2e = args.value_basis == "end_to_end"
        collective = "no collective (one rank)" if args.gpus == 1 else ("RCCL" if args.backend == "nccl" else args.backend + " (through host memory)")
        out = {
            "metric": "verified tx/sec (batched)", "value": round(total_txs / elapsed, 1), "unit": "tx/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": ("%d-tx mixed batch per GPU (multisig set-ups, one- and two-asset transfers, burns, co-signed transfers), %d account groups" % (args.txs, args.txs // 16)) if mixed
                       else "%d-tx batch per GPU, %d transfer(s) each, single asset, %d sender accounts" % (args.txs, args.transfers, senders),
                       "txs_per_gpu": args.txs, "transfers_per_tx": args.transfers, "parallelism": ("one %d-tx batch cut along account groups into %d shards, %s MIN all-reduce of the 8-byte verdict key" % (args.txs * args.gpus, args.gpus, collective)) if split
                       else "tx-sharded x%d (one batch per rank), %s MIN all-reduce of accept bits" % (args.gpus, collective),
                       "step": ("one xhe_verify_batch call from host buffers: parse + ledger keys + host-to-device copy + verification + ledger read-back (the reference's "
                                "Transaction::verify_batch seam); `resident` is the verification of an already uploaded batch" if basis_e2e
                                else "one xhe_batch_verify on a batch uploaded before the timed region; `end_to_end` is the call from host buffers"),
                       "record_bytes": len(batch.records)},
            "value_basis": args.value_basis,
            ("resident" if basis_e2e else "end_to_end"): beside,
            "roofline": roofline, "valu_issue": valu, "cpu_baseline": cpu_baseline, "pipelined": pipelined, "configs": configs, "balance_decoding": ecdlp,
            "stages_ms": {k: round(v, 4) for k, v in stages.items()},
            "stage_detail_ms": {k: round(v, 4) for k, v in detail.items()},
            "setup": {"generate_s": round(gen_s, 2), "range_proofs": "cpu" if gen_engine is None else "gpu", "parse_upload_s": round(resident.upload_s, 4)},
        }
        print(json.dumps(out), flush=True)
    resident.free()
    engine.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
