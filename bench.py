#!/usr/bin/env python3
"""Benchmark of the hot path: Transaction::verify_batch on MI355X.

A step is one verification of one synthetic batch that is already resident in HBM
(xhe_batch_upload once, then xhe_batch_verify per step).  The workload is BASELINE.json's configs[2], the
largest single-GPU configuration: a 100 000-transaction batch, 8 transfers each (aggregated 64-bit range
proofs over 16 value commitments), single (native) asset, per GPU; its 100 000 range proofs are generated
on the GPU before the timed region (xhe_synth_transfers_gpu).  configs[1] (10 000 transactions, 1 transfer
each) is measured after the timed region and reported as `secondary`; `end_to_end` is xhe_verify_batch from
host buffers (parsing, ledger-key resolution and the host-to-device copy included).  With --gpus N every rank
verifies its own batch (transactions shard by account group, no data-path collective) and the ranks
AND-reduce their accept bits with one RCCL all-reduce (MIN) per step; value = transactions of all ranks
/ max-over-ranks time (weak scaling).

Prints one JSON line (rank 0).  roofline: the dominant kernel is the bucket accumulation of the
Pippenger multiscalar check; `achieved` = algorithmic bytes of the whole multiscalar check / that
kernel's average launch time, measured with HIP events on the library's stream.
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


def measured_traffic(txs, transfers):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this very
    workload (FETCH_SIZE and WRITE_SIZE need separate passes and cannot be collected from inside this process):
    (FETCH_SIZE + WRITE_SIZE) KiB as reported, no correction factor.  None when no matching summary is committed."""
    import csv
    import glob
    total, found = 0.0, 0
    for counter in ("fetch_size", "write_size"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s_%dtx%s.csv" % (counter, txs, "" if transfers == 1 else "_t%d" % transfers))))
        if not files:
            return None
        with open(files[-1]) as f:
            for row in csv.DictReader(f):
                if row["kernel"].endswith("k_msm_accumulate"):
                    total += float(row["avg"]) * 1024.0
                    found += 1
    return int(total) if found == 2 else None


def measured_valu_instructions(txs, transfers):
    """Wave-level VALU instructions per launch of the dominant kernel (SQ_INSTS_VALU) from the committed rocprofv3 --pmc
    pass of this very workload; None when no matching summary is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_valu_%dtx%s.csv" % (txs, "" if transfers == 1 else "_t%d" % transfers))))
    if not files:
        return None
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if row["kernel"].endswith("k_msm_accumulate") and row["counter"] == "SQ_INSTS_VALU":
                return float(row["avg"])
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--txs", type=int, default=100000, help="transactions per GPU batch")
    ap.add_argument("--transfers", type=int, default=8)
    ap.add_argument("--no-secondary", action="store_true", help="skip the 10 000 x 1-transfer measurement after the timed region")
    ap.add_argument("--e2e-reps", type=int, default=3, help="xhe_verify_batch calls from host buffers after the timed region (0 = skip)")
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
    return ap.parse_args()


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
    """Rank 0 generates the whole batch, cuts it and scatters the shards; returns this rank's FlatBatch and its transactions' global indices."""
    from xelis_he_amd import sharding
    from xelis_he_amd.api import SyntheticBatch
    parts = None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        big = SyntheticBatch(args.txs * world, args.transfers, min(args.senders * world, args.txs * world), seed=b"bench-split", threads=args.gen_threads,
                             engine=None if args.cpu_prover else engine)
        offs = [big.offsets[i] for i in range(big.n_txs + 1)]
        recs = [big.records[offs[i]:offs[i + 1]] for i in range(big.n_txs)]
        shards = sharding.partition(recs, world)

        class RawView(ctypes.Structure):
            _fields_ = [("n_accounts", ctypes.c_uint64), ("account_keys", ctypes.c_void_p), ("account_nonces", ctypes.c_void_p), ("n_balances", ctypes.c_uint64),
                        ("balance_account", ctypes.c_void_p), ("balance_assets", ctypes.c_void_p), ("balance_cts", ctypes.c_void_p), ("n_multisig", ctypes.c_uint64),
                        ("multisig_account", ctypes.c_void_p), ("multisig_threshold", ctypes.c_void_p), ("multisig_offsets", ctypes.c_void_p), ("multisig_signers", ctypes.c_void_p)]
        rv = ctypes.cast(big.view, ctypes.POINTER(RawView)).contents
        bal_acc = ctypes.cast(rv.balance_account, ctypes.POINTER(ctypes.c_uint32))
        ledger = {"keys": ctypes.string_at(rv.account_keys, 32 * rv.n_accounts), "assets": ctypes.string_at(rv.balance_assets, 32 * rv.n_balances),
                  "cts": ctypes.string_at(rv.balance_cts, 64 * rv.n_balances), "balance_account": [bal_acc[i] for i in range(rv.n_balances)]}
        parts = []
        for s in shards:
            o = [0]
            for i in s:
                o.append(o[-1] + len(recs[i]))
            parts.append((b"".join(recs[i] for i in s), o, s, ledger))
        big.close()
    got = [None]
    dist.scatter_object_list(got, parts, src=0)
    blob, o, idx, ledger = got[0]
    return FlatBatch(blob, o, ledger), idx


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


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    dist = None
    if args.one_device:
        if world > 1 and args.backend != "gloo":
            raise SystemExit("--one-device is a rehearsal mode for --backend gloo")
        local_rank = 0
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        # The library keeps its four kernel streams (the signatures on their own: 20.7 ms per headline step against 23.2 ms with
        # them behind the bulk decode).  The collective of a step runs between two verifications, never beside one, so RCCL's
        # stream sharing a hardware queue (the process has four) with one of the library's costs nothing.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback for the verification path")
    torch.cuda.set_device(local_rank)

    import xelis_he_amd
    from xelis_he_amd.api import LedgerOut, SyntheticBatch, Verdict, _lib
    lib = _lib()
    lib.xhe_batch_upload.restype = ctypes.c_int32
    lib.xhe_batch_verify.restype = ctypes.c_int32
    engine = xelis_he_amd.Engine(local_rank)

    t0 = time.time()
    senders = min(args.senders, args.txs)
    mixed = args.workload == "mixed"
    if mixed:
        args.txs -= args.txs % 16
    gen_engine = None if (mixed or args.cpu_prover) else engine
    split = args.split_one_batch and dist is not None and not mixed
    global_index = None
    if split:
        batch, global_index = split_one_batch(args, dist, rank, world, engine)
    else:
        batch = SyntheticBatch(args.txs, args.transfers, senders, seed=b"bench-rank-%d" % rank, threads=args.gen_threads, mixed=mixed, engine=gen_engine)
    gen_s = time.time() - t0
    res = Resident(engine, lib, batch)

    on_host = args.backend != "nccl"              # gloo gets host tensors only
    flag = torch.ones(1, dtype=torch.int64, device="cpu" if (on_host or dist is None) else "cuda")

    def step(i):
        ok = res.step(i)
        if dist is not None and split:
            # one batch over all ranks: the MIN of the shards' 8-byte verdict keys is the batch's verdict (accept = the largest key)
            from xelis_he_amd import sharding
            flag.fill_(sharding.verdict_key(bool(ok), res.verdict.err_code, res.verdict.first_bad_tx, global_index))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = 1 if int(flag.item()) == sharding.ACCEPT_KEY else 0
        elif dist is not None:
            flag.fill_(ok)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)       # AND of the per-shard accept bits
            ok = int(flag.item())
        return ok

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # timed region: HIP events only around the dominant kernel (stage timing level 1)
    engine.set_stage_timing(1)
    for i in range(args.warmup):
        assert step(i) == 1, "verification rejected a valid batch"
    stage_sum, stage_n = {}, 0
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        assert step(args.warmup + i) == 1, "verification rejected a valid batch"
        for name, ms in engine.stage_times():
            stage_sum[name] = stage_sum.get(name, 0.0) + ms
        stage_n += 1
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_host else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    stages = {k: v / max(stage_n, 1) for k, v in stage_sum.items()}
    # outside the timed region: the per-stage breakdown (an event per stage boundary), three more verifications
    engine.set_stage_timing(2)
    detail, reps = {}, 3
    for i in range(reps):
        assert step(args.warmup + args.steps + i) == 1
        for name, ms in engine.stage_times():
            detail[name] = detail.get(name, 0.0) + ms / reps
    # algorithmic bytes of the multiscalar check: per dynamic term a 32-byte scalar + a 128-byte decoded
    # point record, per static generator the same, 32 bytes of result
    dyn_terms, static_terms = term_counts(args.txs, args.transfers, mixed)
    algo_bytes = (dyn_terms + static_terms) * 160 + 32
    acc_ms = stages.get("msm_accumulate", 0.0)
    roofline = None
    if acc_ms > 0:
        achieved = algo_bytes / (acc_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6),
                    "traffic": None if mixed else measured_traffic(args.txs, args.transfers), "kernel": "k_msm_accumulate", "kernel_ms": round(acc_ms, 4),
                    "algorithmic_bytes": algo_bytes, "terms": dyn_terms + static_terms}
        roofline["traffic_source"] = ("FETCH_SIZE + WRITE_SIZE of the kernel per launch, newest committed rocprofv3 --pmc passes of this workload under profiles/"
                                      if roofline["traffic"] is not None else "unmeasured for this workload (no committed --pmc pass)")

    # The kernel is bound by VALU issue, not by HBM: wave-instructions per launch (PMC) / launch time against the chip's vector
    # issue rate, one wave-instruction per 2 cycles per SIMD (a 64-wide wavefront on a SIMD-32 takes two passes).  The kernel's
    # mix is mostly v_mad_u64_u32, which issues in 5.5 cycles (tools/ubench_valu.hip, 8 waves per SIMD): `mix_bound_frac` prices
    # the measured instruction count at the mix-weighted cost of profiles/r02_accumulate_mix.json when that file is committed.
    valu = None
    n_valu = measured_valu_instructions(args.txs, args.transfers) if not mixed else None
    if n_valu and acc_ms > 0:
        prop = torch.cuda.get_device_properties(local_rank)
        clock_hz = getattr(prop, "clock_rate", 2400000) * 1e3
        simds = prop.multi_processor_count * 4
        peak = simds * clock_hz / 2.0
        rate = n_valu / (acc_ms * 1e-3)
        valu = {"kernel": "k_msm_accumulate", "wave_instructions": int(n_valu), "achieved": round(rate / 1e9, 2), "peak": round(peak / 1e9, 2),
                "unit": "G wave-instructions/s", "frac": round(rate / peak, 4), "cycles_per_wave_instruction_at_peak": 2, "simds": simds, "clock_mhz": round(clock_hz / 1e6)}
        mix_path = os.path.join(ROOT, "profiles", "r02_accumulate_mix.json")
        if os.path.exists(mix_path):
            with open(mix_path) as f:
                mix = json.load(f)
            cyc = float(mix["cycles_per_wave_instruction"])
            valu["mix_cycles_per_wave_instruction"] = cyc
            valu["mix_bound_frac"] = round(rate / (simds * clock_hz / cyc), 4)

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

    # after the timed region, rank 0 at N = 1: the drop-in call from host buffers, and the 10 000 x 1-transfer configuration
    end_to_end, secondary = None, None
    if rank == 0 and args.gpus == 1:
        engine.set_stage_timing(0)
        if args.e2e_reps > 0:
            lib.xhe_verify_batch.restype = ctypes.c_int32
            view = batch.view.contents
            out_cts = (ctypes.c_uint8 * max(64 * view.n_balances, 1))()
            out_nonces = (ctypes.c_uint64 * max(view.n_accounts, 1))()
            ms_set = (ctypes.c_int64 * max(view.n_accounts, 1))()
            out = LedgerOut(out_cts, out_nonces, ms_set)
            v = Verdict()
            times = []
            for i in range(args.e2e_reps + 1):
                t0 = time.perf_counter()
                engine._check(lib.xhe_verify_batch(ctypes.c_void_p(engine.ctx), batch.records, batch.offsets, ctypes.c_uint64(batch.n_txs), batch.view, ctypes.byref(out),
                                                   ctypes.byref(v), (1000 + i).to_bytes(8, "little") * 4))
                times.append(time.perf_counter() - t0)
                assert v.accept == 1
            best = min(times[1:])
            end_to_end = {"call": "xhe_verify_batch from host buffers (parse + ledger keys + H2D + verify + ledger read-back; batches from 128 MiB of records "
                                  "on are cut into parts inside the call, the host work of a part beside the verification of the part before)",
                          "ms_per_batch": round(best * 1e3, 3),
                          "value": round(batch.n_txs / best, 1), "unit": "tx/s", "record_bytes": len(batch.records), "reps": args.e2e_reps}
            # the same call over a run of batches, pipelined inside the library (upload of batch k + 1 beside the verification of batch k)
            class BatchArgs(ctypes.Structure):
                _fields_ = [("records", ctypes.c_char_p), ("offsets", ctypes.POINTER(ctypes.c_uint64)), ("n_txs", ctypes.c_uint64), ("view", ctypes.c_void_p),
                            ("out", ctypes.c_void_p), ("seed", ctypes.c_char_p)]
            lib.xhe_verify_batches.restype = ctypes.c_int32

            def run_of(n_b):
                arr = (BatchArgs * n_b)()
                seeds = [(2000 + i).to_bytes(8, "little") * 4 for i in range(n_b)]
                for i in range(n_b):
                    arr[i].records, arr[i].offsets, arr[i].n_txs = batch.records, batch.offsets, batch.n_txs
                    arr[i].view, arr[i].out, arr[i].seed = ctypes.cast(batch.view, ctypes.c_void_p), ctypes.cast(ctypes.pointer(out), ctypes.c_void_p), seeds[i]
                vs = (Verdict * n_b)()
                t0 = time.perf_counter()
                engine._check(lib.xhe_verify_batches(ctypes.c_void_p(engine.ctx), ctypes.c_uint64(n_b), arr, vs))
                dt = time.perf_counter() - t0
                assert all(v_.accept == 1 for v_ in vs)
                return dt

            # a run of 3 and a run of 9: the difference is six batches in the steady state (the first upload of a run overlaps nothing)
            t3, t9 = run_of(3), run_of(9)
            steady = max(t9 - t3, 1e-9) / 6
            end_to_end["pipelined"] = {"call": "xhe_verify_batches over 9 such batches from host buffers", "ms_per_batch": round(t9 / 9 * 1e3, 3),
                                       "value": round(9 * batch.n_txs / t9, 1), "unit": "tx/s", "steady_ms_per_batch": round(steady * 1e3, 3),
                                       "steady_value": round(batch.n_txs / steady, 1), "steady_note": "(run of 9 - run of 3) / 6"}
        if not args.no_secondary and not mixed and (args.txs, args.transfers) != (10000, 1):
            t0 = time.time()
            b2 = SyntheticBatch(10000, 1, 1024, seed=b"bench-secondary", threads=args.gen_threads, engine=gen_engine)
            g2 = time.time() - t0
            engine.set_stage_timing(1)
            r2 = Resident(engine, lib, b2)
            for i in range(10):
                assert r2.step(i) == 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            acc2 = 0.0
            for i in range(50):
                assert r2.step(10 + i) == 1
                acc2 += dict(engine.stage_times()).get("msm_accumulate", 0.0)
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t0
            d2, s2 = term_counts(10000, 1, False)
            secondary = {"workload": "10000-tx batch, 1 transfer each, single asset, 1024 sender accounts", "ms_per_step": round(e2 / 50 * 1e3, 4), "value": round(10000 * 50 / e2, 1),
                         "unit": "tx/s", "steps": 50, "warmup": 10, "msm_accumulate_ms": round(acc2 / 50, 4),
                         "roofline_frac": round(((d2 + s2) * 160 + 32) / (acc2 / 50 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6) if acc2 > 0 else None, "generate_s": round(g2, 2)}
            r2.free()
            b2.close()

    if rank == 0:
        total_txs = args.txs * args.gpus * args.steps
        out = {
            "metric": "verified tx/sec (batched)", "value": round(total_txs / elapsed, 1), "unit": "tx/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": ("%d-tx mixed batch per GPU (multisig set-ups, one- and two-asset transfers, burns, co-signed transfers), %d account groups" % (args.txs, args.txs // 16)) if mixed
                       else "%d-tx batch per GPU, %d transfer(s) each, single asset, %d sender accounts" % (args.txs, args.transfers, senders),
                       "txs_per_gpu": args.txs, "transfers_per_tx": args.transfers, "parallelism": ("one %d-tx batch cut along account groups into %d shards, RCCL MIN all-reduce of the 8-byte verdict key" % (args.txs * args.gpus, args.gpus)) if split
                       else "tx-sharded x%d (one batch per rank), RCCL MIN all-reduce of accept bits" % args.gpus,
                       "inputs": "resident in HBM (xhe_batch_upload before the timed region); see end_to_end for the call from host buffers"},
            "roofline": roofline, "valu_issue": valu, "cpu_baseline": cpu_baseline, "end_to_end": end_to_end, "secondary": secondary,
            "stages_ms": {k: round(v, 4) for k, v in stages.items()},
            "stage_detail_ms": {k: round(v, 4) for k, v in detail.items()},
            "setup": {"generate_s": round(gen_s, 2), "range_proofs": "cpu" if gen_engine is None else "gpu", "parse_upload_s": round(res.upload_s, 4)},
        }
        print(json.dumps(out), flush=True)
    res.free()
    engine.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
