"""The batched range-proof prover on the GPU (xhe_range_proofs_prove behind xhe_synth_transfers_gpu) against the CPU prover:
same seeds -> the same record bytes; the records verify on the GPU and in the oracle."""
import ctypes
import time

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import xelis_he_amd
    e = xelis_he_amd.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("n_txs,transfers,senders", [(5, 1, 2), (70, 1, 7), (9, 2, 3), (6, 3, 2), (5, 5, 1), (67, 8, 5), (3, 15, 1)])
def test_gpu_prover_matches_cpu_prover(engine, oracle, n_txs, transfers, senders):
    from xelis_he_amd.api import SyntheticBatch, Verdict, _lib
    seed = b"prover-%d-%d" % (n_txs, transfers)
    t0 = time.time()
    cpu = SyntheticBatch(n_txs, transfers, senders, seed=seed, threads=8)
    t1 = time.time()
    gpu = SyntheticBatch(n_txs, transfers, senders, seed=seed, threads=8, engine=engine)
    t2 = time.time()
    print("cpu %.2fs gpu %.2fs" % (t1 - t0, t2 - t1), flush=True)
    assert list(gpu.offsets) == list(cpu.offsets)
    if gpu.records != cpu.records:
        first = next(i for i in range(len(cpu.records)) if cpu.records[i] != gpu.records[i])
        tx = max(k for k in range(n_txs) if cpu.offsets[k] <= first)
        raise AssertionError("records differ first at byte %d (tx %d, offset %d inside the record)" % (first, tx, first - cpu.offsets[tx]))
    lib = _lib()
    v = Verdict()
    rc = lib.xhe_verify_batch(engine.ctx, gpu.records, gpu.offsets, gpu.n_txs, gpu.view, None, ctypes.byref(v), b"\x05" * 32)
    assert rc == 0 and v.accept == 1, (rc, v.accept, v.err_code, v.first_bad_tx)
