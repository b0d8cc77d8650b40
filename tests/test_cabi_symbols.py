"""The C-ABI library loads without a GPU and exports every function that
include/vdb_core.h declares; the ctypes binding covers the same set."""

import ctypes
import os
import re

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "vdb_core.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vdb_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_functions():
    names = declared_functions()
    assert len(names) >= 40 and "vdb_flat_search" in names and "vdb_pq_search" in names


def test_library_exports_every_declared_symbol():
    from vectordb_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"not exported: {missing}"


def test_binding_covers_header():
    from vectordb_amd import _native
    assert sorted(_native.SIGNATURES) == declared_functions()
    lib = _native.load()          # binds argtypes for every symbol; no GPU call is made
    assert lib.vdb_last_error() is not None


def test_pack_mask_layout():
    import numpy as np
    from vectordb_amd._native import pack_mask
    m = np.zeros(70, dtype=bool)
    m[[0, 31, 32, 69]] = True
    w = pack_mask(m)
    assert w.dtype == np.uint32 and len(w) == 3
    assert w[0] == (1 | (1 << 31)) and w[1] == 1 and w[2] == (1 << 5)
