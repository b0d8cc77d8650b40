"""Short runs of the randomised parity soaks (tests/fuzz_*.py) inside the GPU suite: each script draws random
shapes, metrics, filters and batch sizes for a few seconds, compares every case with the CPU oracle and stops at
the first difference.  The long soaks are run by hand with more seconds and other seeds."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def run_soak(script, seconds, seed, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(HERE, script), "--seconds", str(seconds), "--seed", str(seed)],
                       capture_output=True, text=True, timeout=600, env=e)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, f"{script} seed {seed} failed:\n{tail}"
    assert "no failure" in r.stdout, tail
    return r.stdout


def test_flat_soak(native):
    run_soak("fuzz_flat.py", 12, 101)


def test_flat_soak_row_resident_kernel(native):
    # VDB_RR_MIN_ITEMS=1: batched Manhattan / Chebyshev over rows of 32, 64 or 128 floats take the row-resident
    # kernel at every size instead of only from a few thousand row chunks on
    # VDB_FLAT_BOUND_MIN_ROWS: the bounded form of those batches (sample bound + per-query queues) from 4096 rows on
    run_soak("fuzz_flat.py", 12, 102, env={"VDB_RR_MIN_ITEMS": "1", "VDB_FLAT_BOUND_MIN_ROWS": "4096"})


def test_ivf_pq_soak(native):
    run_soak("fuzz_ivf_pq.py", 15, 103)


def test_ivf_pq_soak_pq_bound_pass_on_small_sets(native):
    # VDB_PQ_BOUND_MIN_TILES=8: PQ scans of one to three queries take their bound pass and the packed candidate
    # queues from 2048 codes on instead of from 4M
    run_soak("fuzz_ivf_pq.py", 12, 105, env={"VDB_PQ_BOUND_MIN_TILES": "8"})


def test_bf16_soak(native):
    run_soak("fuzz_bf16.py", 10, 104)


def test_prefilter_soak(native):
    # VDB_SAMPLE_MIN_TILES=32: batches take the candidate-queue kernels (and with them the bf16 filter pass) from
    # 4096 rows on, so a short run covers many shapes; every case is compared bit for bit with the filter off
    run_soak("fuzz_prefilter.py", 12, 106, env={"VDB_SAMPLE_MIN_TILES": "32"})

