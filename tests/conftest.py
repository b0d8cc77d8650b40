import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def native():
    """The C-ABI library, initialised on device 0.  GPU tests must run through
    it; a missing library or GPU is a failure, not a skip."""
    from vectordb_amd import _native
    lib = _native.load()
    _native.init(0)
    return lib


def same_ids_or_near_ties(got_rows, ref_rows, ref_dist_of_row, kth_ref_dist, rtol=1e-5, atol=1e-6):
    """Set equality of result ids, except that an id may differ when the
    reference's own distance for it is within rounding of the reference's k-th
    distance (BLAS-vs-MFMA summation order moves such rows across the cut)."""
    got, ref = set(int(r) for r in got_rows), set(int(r) for r in ref_rows)
    for r in got ^ ref:
        d = float(ref_dist_of_row(r))
        if abs(d - kth_ref_dist) > rtol * abs(kth_ref_dist) + atol:
            return False
    return True
