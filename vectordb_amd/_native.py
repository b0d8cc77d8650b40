"""ctypes binding of libvdbcore.so (the C ABI declared in include/vdb_core.h).

There is no CPU fallback: if the library cannot be loaded, or a call fails,
an exception is raised.  The status codes are mapped to the exception types
the reference raises at the same places (ValueError for bad shapes/metrics,
RuntimeError for state errors; index/base.py:443-449, index/ivf.py:459,595).
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VDB_LIB_PATH selects another build of the same library (kernel experiments)
LIB_PATH = os.environ.get("VDB_LIB_PATH") or os.path.join(_HERE, "_lib", "libvdbcore.so")

VDB_OK, VDB_ERR_INVALID, VDB_ERR_HIP, VDB_ERR_STATE, VDB_ERR_NOMEM = 0, -1, -2, -3, -4

# vdb_metric
EUCLIDEAN, EUCLIDEAN_SQ, COSINE, DOT, MANHATTAN, CHEBYSHEV = 0, 1, 2, 3, 4, 5
EUCLIDEAN_DIRECT, COSINE_PAIR, EUCLIDEAN_SQ_DIRECT = 6, 7, 8
HAMMING, ANGULAR = 9, 10

METRIC_CODES = {
    "euclidean": EUCLIDEAN,
    "euclidean_squared": EUCLIDEAN_SQ,
    "cosine": COSINE,
    "dot": DOT,
    "manhattan": MANHATTAN,
    "chebyshev": CHEBYSHEV,
}

c_float_p = C.POINTER(C.c_float)
c_i64_p = C.POINTER(C.c_int64)
c_i32_p = C.POINTER(C.c_int32)
c_u32_p = C.POINTER(C.c_uint32)
c_u8_p = C.POINTER(C.c_uint8)
c_double_p = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol declared in include/vdb_core.h
SIGNATURES = {
    "vdb_init": (C.c_int, [C.c_int]),
    "vdb_last_error": (C.c_char_p, []),
    "vdb_device_info": (C.c_int, [C.c_char_p, C.c_int, c_i64_p, c_i64_p]),
    "vdb_synchronize": (C.c_int, []),
    "vdb_measure_ceilings": (C.c_int, [c_float_p, c_float_p, c_float_p, c_float_p]),
    "vdb_profile_enable": (C.c_int, [C.c_int]),
    "vdb_profile_read": (C.c_int, [c_float_p, C.POINTER(C.c_int)]),
    "vdb_mask_create": (C.c_int, [C.POINTER(C.c_void_p), c_u32_p, C.c_int64]),
    "vdb_mask_destroy": (C.c_int, [C.c_void_p]),
    "vdb_mask_info": (C.c_int, [C.c_void_p, c_i64_p, C.POINTER(C.c_void_p)]),
    "vdb_flat_search_masked": (C.c_int, [C.c_void_p, C.c_int, c_float_p, C.c_int, C.c_int, C.c_void_p, c_float_p, c_i64_p]),
    "vdb_ivf_search_masked": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_int, C.c_void_p, c_float_p, c_i64_p]),
    "vdb_pq_search_masked": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_void_p, c_float_p, c_i64_p]),
    "vdb_ivfpq_search_masked": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_int, C.c_void_p, c_float_p, c_i64_p]),
    "vdb_rows_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "vdb_rows_create_typed": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "vdb_rows_destroy": (C.c_int, [C.c_void_p]),
    "vdb_rows_append": (C.c_int, [C.c_void_p, c_float_p, C.c_int64]),
    "vdb_rows_append_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "vdb_rows_set": (C.c_int, [C.c_void_p, C.c_int64, c_float_p]),
    "vdb_rows_truncate": (C.c_int, [C.c_void_p, C.c_int64]),
    "vdb_rows_size": (C.c_int, [C.c_void_p, c_i64_p]),
    "vdb_rows_norms": (C.c_int, [C.c_void_p, c_float_p, c_float_p]),
    "vdb_rows_set_cosine_eps": (C.c_int, [C.c_void_p, C.c_float]),
    "vdb_rows_set_prefilter": (C.c_int, [C.c_void_p, C.c_int]),
    "vdb_flat_search_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vdb_flat_search": (C.c_int, [C.c_void_p, C.c_int, c_float_p, C.c_int, C.c_int, c_u32_p, c_float_p, c_i64_p]),
    "vdb_flat_search_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "vdb_flat_search_static_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "vdb_flat_distances": (C.c_int, [C.c_void_p, C.c_int, c_float_p, C.c_int, c_float_p]),
    "vdb_flat_distances_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "vdb_flat_range_search": (C.c_int, [C.c_void_p, C.c_int, c_float_p, C.c_float, c_u32_p, C.c_int64, c_float_p,
                                        c_i64_p, c_i64_p]),
    "vdb_assign": (C.c_int, [c_float_p, C.c_int64, c_float_p, C.c_int, C.c_int, C.c_int, c_i32_p, c_float_p]),
    "vdb_kmeans_accumulate": (C.c_int, [c_float_p, C.c_int64, C.c_int, c_i32_p, C.c_int, c_double_p, c_i64_p]),
    "vdb_kmeans_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vdb_kmeans_destroy": (C.c_int, [C.c_void_p]),
    "vdb_kmeans_init_rows": (C.c_int, [C.c_void_p, c_i64_p, C.c_int]),
    "vdb_kmeans_init_pp": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_i64_p]),
    "vdb_kmeans_set_centroids": (C.c_int, [C.c_void_p, c_float_p]),
    "vdb_kmeans_step": (C.c_int, [C.c_void_p, C.c_int, c_i64_p, c_double_p]),
    "vdb_kmeans_reseed": (C.c_int, [C.c_void_p, c_i32_p, c_i64_p, C.c_int, c_double_p]),
    "vdb_kmeans_commit": (C.c_int, [C.c_void_p]),
    "vdb_kmeans_centroids": (C.c_int, [C.c_void_p, c_float_p, c_double_p]),
    "vdb_kmeans_assignments": (C.c_int, [C.c_void_p, c_i32_p, c_float_p]),
    "vdb_ivf_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "vdb_ivf_destroy": (C.c_int, [C.c_void_p]),
    "vdb_ivf_set_centroids": (C.c_int, [C.c_void_p, c_float_p, C.c_int]),
    "vdb_ivf_build": (C.c_int, [C.c_void_p, c_float_p, C.c_int64, c_i32_p]),
    "vdb_ivf_build_lists": (C.c_int, [C.c_void_p, c_float_p, C.c_int64, c_i64_p, c_i64_p]),
    "vdb_ivf_build_numbered": (C.c_int, [C.c_void_p, c_float_p, C.c_int64, c_i64_p, c_i64_p]),
    "vdb_ivf_append": (C.c_int, [C.c_void_p, c_float_p, c_i64_p, c_i32_p, C.c_int64]),
    "vdb_ivf_remove_at": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64]),
    "vdb_ivf_layout": (C.c_int, [C.c_void_p, c_i64_p, c_i64_p, c_i64_p]),
    "vdb_ivf_search": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_int, c_u32_p, c_float_p, c_i64_p]),
    "vdb_ivf_search_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "vdb_ivf_search_ordered_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdb_ivf_probe": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, c_i32_p, c_float_p]),
    "vdb_ivf_last_scan_rows": (C.c_int, [C.c_void_p, c_i64_p]),
    "vdb_ivf_set_prefilter": (C.c_int, [C.c_void_p, C.c_int]),
    "vdb_ivf_search_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vdb_pq_search_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vdb_pq_encode": (C.c_int, [c_float_p, C.c_int64, c_float_p, C.c_int, C.c_int, C.c_int, c_u8_p]),
    "vdb_pq_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "vdb_pq_destroy": (C.c_int, [C.c_void_p]),
    "vdb_pq_set_codebook": (C.c_int, [C.c_void_p, c_float_p]),
    "vdb_pq_append_codes": (C.c_int, [C.c_void_p, c_u8_p, C.c_int64]),
    "vdb_pq_append_codes_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vdb_pq_append_rows": (C.c_int, [C.c_void_p, c_float_p, C.c_int64, c_u8_p]),
    "vdb_pq_truncate": (C.c_int, [C.c_void_p, C.c_int64]),
    "vdb_pq_set_code": (C.c_int, [C.c_void_p, C.c_int64, c_u8_p]),
    "vdb_pq_size": (C.c_int, [C.c_void_p, c_i64_p]),
    "vdb_pq_lut": (C.c_int, [C.c_void_p, c_float_p, c_float_p]),
    "vdb_pq_search": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, c_u32_p, c_float_p, c_i64_p]),
    "vdb_pq_search_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "vdb_ivfpq_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "vdb_ivfpq_destroy": (C.c_int, [C.c_void_p]),
    "vdb_ivfpq_set_quantizers": (C.c_int, [C.c_void_p, c_float_p, C.c_int, c_float_p]),
    "vdb_ivfpq_encode": (C.c_int, [C.c_void_p, c_float_p, C.c_int64, c_i32_p, c_u8_p]),
    "vdb_ivfpq_build": (C.c_int, [C.c_void_p, c_u8_p, C.c_int64, c_i64_p, c_i64_p]),
    "vdb_ivfpq_build_numbered": (C.c_int, [C.c_void_p, c_u8_p, C.c_int64, c_i64_p, c_i64_p]),
    "vdb_ivfpq_append": (C.c_int, [C.c_void_p, c_u8_p, c_i64_p, c_i32_p, C.c_int64]),
    "vdb_ivfpq_remove_at": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64]),
    "vdb_ivfpq_layout": (C.c_int, [C.c_void_p, c_i64_p, c_i64_p, c_i64_p]),
    "vdb_ivfpq_search": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, C.c_int, c_u32_p, c_float_p, c_i64_p]),
    "vdb_ivfpq_search_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "vdb_topk_merge_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "vdb_topk_merge_strided_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                             C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdb_topk_merge_keyed_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "vdb_topk_merge_keyed_strided_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None
_lock = threading.Lock()


class NativeError(RuntimeError):
    """A HIP runtime failure or an allocation failure inside the library."""


def load() -> C.CDLL:
    """Load libvdbcore.so and bind every declared symbol.  Raises if the
    library is missing: the HIP path is the only path."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C vectordb_amd/csrc`")
        # PyTorch ships its own libamdhip64.so.7; the HIP runtime that is loaded first
        # serves the whole process.  Tensors handed to the *_dev entry points must
        # come from the same runtime instance, so torch's copy goes first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def last_error() -> str:
    msg = load().vdb_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status: int) -> None:
    if status == VDB_OK:
        return
    msg = last_error()
    if status == VDB_ERR_INVALID:
        raise ValueError(msg)
    if status == VDB_ERR_STATE:
        raise RuntimeError(msg)
    if status == VDB_ERR_NOMEM:
        raise MemoryError(msg)
    raise NativeError(msg)


_initialised = False
_device: Optional[int] = None


def init(device: Optional[int] = None) -> None:
    """One process per GPU: LOCAL_RANK picks the device unless one is given.  The first call
    fixes the device; asking for another one later raises RuntimeError (vdb_init refuses)."""
    global _initialised, _device
    if device is None:
        if _initialised:
            return
        device = int(os.environ.get("LOCAL_RANK", "0"))
    check(load().vdb_init(int(device)))
    _initialised = True
    _device = int(device)


def current_device() -> Optional[int]:
    """The device the library was initialised on, or None before the first init()."""
    return _device


def device_info() -> dict:
    init()
    name = C.create_string_buffer(256)
    total, free = C.c_int64(0), C.c_int64(0)
    check(load().vdb_device_info(name, 256, C.byref(total), C.byref(free)))
    return {"name": name.value.decode(), "memory_total": total.value, "memory_free": free.value}


def synchronize() -> None:
    check(load().vdb_synchronize())


def measure_ceilings():
    """(device-to-device copy GB/s counting read + write, fp32 MFMA loop TFLOP/s, bf16 MFMA loop TFLOP/s,
    vector-ALU subtract/multiply/add loop in 10^12 operations/s)."""
    init()
    a, b, c, d = C.c_float(0.0), C.c_float(0.0), C.c_float(0.0), C.c_float(0.0)
    check(load().vdb_measure_ceilings(C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
    return a.value, b.value, c.value, d.value


def profile_enable(on: bool) -> None:
    check(load().vdb_profile_enable(1 if on else 0))


def profile_read():
    """(summed dominant-kernel milliseconds, launches) since the last read."""
    ms, n = C.c_float(0.0), C.c_int(0)
    check(load().vdb_profile_read(C.byref(ms), C.byref(n)))
    return ms.value, n.value


# ---------------------------------------------------------------- numpy helpers

def f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def fptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(c_float_p)


def pack_mask(mask) -> Optional[np.ndarray]:
    """Boolean row mask -> little-endian packed uint32 words (bit i%32 of word i//32)."""
    if mask is None:
        return None
    m = np.asarray(mask, dtype=bool)
    bits = np.packbits(m, bitorder="little")
    pad = (-len(bits)) % 4
    if pad:
        bits = np.concatenate([bits, np.zeros(pad, dtype=np.uint8)])
    return np.ascontiguousarray(bits.view("<u4"))


def u32ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(c_u32_p)


class DeviceMask:
    """Owner of a vdb_mask handle: a packed row filter that stays in HBM.  Every ``search`` of the handle classes
    below takes either packed host words (uploaded by that call) or one of these (nothing is uploaded)."""

    uploads = 0          # masks sent to the device by this process (what the filter-cache tests count)

    def __init__(self, mask_or_words, n_bits: Optional[int] = None):
        init()
        self.lib = load()
        a = np.asarray(mask_or_words)
        if a.dtype == bool:
            n_bits = len(a) if n_bits is None else n_bits
            a = pack_mask(a)
        words = np.ascontiguousarray(a, dtype=np.uint32)
        self.n_bits = int(len(words) * 32 if n_bits is None else n_bits)
        if (self.n_bits + 31) // 32 > len(words):
            raise ValueError(f"{len(words)} words do not hold {self.n_bits} bits")
        self.h = C.c_void_p()
        check(self.lib.vdb_mask_create(C.byref(self.h), u32ptr(words), self.n_bits))
        DeviceMask.uploads += 1

    @property
    def device_ptr(self) -> int:
        p = C.c_void_p()
        check(self.lib.vdb_mask_info(self.h, None, C.byref(p)))
        return p.value or 0

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.vdb_mask_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


STORAGE = {"f32": 0, "bf16": 1}
PREFILTER = {None: 0, "bf16": 1, "auto": 2}


class RowStore:
    """Owner of a vdb_rows handle (resident row matrix + norms); storage "f32" or "bf16"."""

    def __init__(self, dim: int, storage: str = "f32"):
        init()
        self.lib = load()
        self.dim = int(dim)
        if storage not in STORAGE:
            raise ValueError(f"storage must be one of {sorted(STORAGE)}, got {storage!r}")
        self.storage = storage
        self.h = C.c_void_p()
        check(self.lib.vdb_rows_create_typed(C.byref(self.h), self.dim, STORAGE[storage]))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.vdb_rows_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = C.c_int64(0)
        check(self.lib.vdb_rows_size(self.h, C.byref(n)))
        return n.value

    def append(self, rows: np.ndarray) -> None:
        rows = f32(rows)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"Vector dimension {rows.shape[-1]} != index dimension {self.dim}")
        check(self.lib.vdb_rows_append(self.h, fptr(rows), rows.shape[0]))

    def append_device(self, ptr: int, n: int, ld: int, stream: int = 0) -> None:
        check(self.lib.vdb_rows_append_dev(self.h, C.c_void_p(ptr), n, ld, C.c_void_p(stream)))

    def set_row(self, row: int, vec: np.ndarray) -> None:
        vec = f32(vec)
        check(self.lib.vdb_rows_set(self.h, int(row), fptr(vec)))

    def truncate(self, n: int) -> None:
        check(self.lib.vdb_rows_truncate(self.h, int(n)))

    def set_cosine_eps(self, eps: float) -> None:
        """Norm threshold of the cosine distance MATRIX of this store (pairwise_cosine's eps); searches keep 1e-8."""
        check(self.lib.vdb_rows_set_cosine_eps(self.h, float(eps)))

    def set_prefilter(self, mode: Optional[str]) -> None:
        """Batched searches behind a bf16 filter pass ("bf16"; "auto": while the bf16 copy of the rows leaves a tenth
        of the device memory free) or on the fp32 kernels alone (None): same results."""
        if mode not in PREFILTER:
            raise ValueError(f"prefilter must be None, 'bf16' or 'auto', got {mode!r}")
        check(self.lib.vdb_rows_set_prefilter(self.h, PREFILTER[mode]))

    def used_prefilter(self) -> bool:
        """True when the last search on this store ran the bf16 filter pass."""
        a, b = C.c_int(0), C.c_int(0)
        check(self.lib.vdb_flat_search_stats(self.h, C.byref(a), C.byref(b)))
        return a.value == 2

    def norms(self):
        n = len(self)
        sq, nr = np.empty(n, np.float32), np.empty(n, np.float32)
        check(self.lib.vdb_rows_norms(self.h, fptr(sq), fptr(nr)))
        return sq, nr

    def search(self, queries: np.ndarray, k: int, metric: int, mask_words: Optional[np.ndarray] = None):
        q = f32(queries)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        dist = np.empty((nq, k), np.float32)
        rows = np.empty((nq, k), np.int64)
        if isinstance(mask_words, DeviceMask):
            check(self.lib.vdb_flat_search_masked(self.h, metric, fptr(q), nq, k, mask_words.h, fptr(dist),
                                                  rows.ctypes.data_as(c_i64_p)))
        else:
            check(self.lib.vdb_flat_search(self.h, metric, fptr(q), nq, k, u32ptr(mask_words), fptr(dist),
                                           rows.ctypes.data_as(c_i64_p)))
        return dist, rows

    def search_stats(self):
        """(used_queue, overflowed) of the last search on this store."""
        a, b = C.c_int(0), C.c_int(0)
        check(self.lib.vdb_flat_search_stats(self.h, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)

    def search_device(self, q_ptr: int, nq: int, k: int, metric: int, out_dist_ptr: int, out_row_ptr: int,
                      mask_ptr: int = 0, stream: int = 0) -> None:
        check(self.lib.vdb_flat_search_dev(self.h, metric, C.c_void_p(q_ptr), nq, k, C.c_void_p(mask_ptr or None),
                                           C.c_void_p(out_dist_ptr), C.c_void_p(out_row_ptr), C.c_void_p(stream or None)))

    def search_device_static(self, q_ptr: int, nq: int, k: int, metric: int, out_dist_ptr: int, out_row_ptr: int,
                             mask_ptr: int = 0, stream: int = 0) -> None:
        """search_device without any allocation: workspaces come from the library's static scratch array."""
        check(self.lib.vdb_flat_search_static_dev(self.h, metric, C.c_void_p(q_ptr), nq, k, C.c_void_p(mask_ptr or None),
                                                  C.c_void_p(out_dist_ptr), C.c_void_p(out_row_ptr),
                                                  C.c_void_p(stream or None)))

    def distances(self, queries: np.ndarray, metric: int) -> np.ndarray:
        q = f32(queries)
        if q.ndim == 1:
            q = q[None, :]
        out = np.empty((q.shape[0], len(self)), np.float32)
        check(self.lib.vdb_flat_distances(self.h, metric, fptr(q), q.shape[0], fptr(out)))
        return out

    def range_search(self, query: np.ndarray, radius: float, metric: int, mask_words=None, cap: Optional[int] = None):
        q = f32(query)
        cap = len(self) if cap is None else int(cap)
        dist = np.empty(max(cap, 1), np.float32)
        rows = np.empty(max(cap, 1), np.int64)
        cnt = C.c_int64(0)
        check(self.lib.vdb_flat_range_search(self.h, metric, fptr(q), float(radius), u32ptr(mask_words), cap,
                                             fptr(dist), rows.ctypes.data_as(c_i64_p), C.byref(cnt)))
        return dist[:cnt.value], rows[:cnt.value]


def assign(x: np.ndarray, centroids: np.ndarray, metric: int):
    init()
    x, c = f32(x), f32(centroids)
    idx = np.empty(len(x), np.int32)
    dist = np.empty(len(x), np.float32)
    check(load().vdb_assign(fptr(x), len(x), fptr(c), len(c), x.shape[1], metric,
                            idx.ctypes.data_as(c_i32_p), fptr(dist)))
    return idx, dist


def kmeans_accumulate(x: np.ndarray, assign_idx: np.ndarray, n_clusters: int):
    init()
    x = f32(x)
    a = np.ascontiguousarray(assign_idx, dtype=np.int32)
    sums = np.empty((n_clusters, x.shape[1]), np.float64)
    counts = np.empty(n_clusters, np.int64)
    check(load().vdb_kmeans_accumulate(fptr(x), len(x), x.shape[1], a.ctypes.data_as(c_i32_p), n_clusters,
                                       sums.ctypes.data_as(c_double_p), counts.ctypes.data_as(c_i64_p)))
    return sums, counts


KMEANS_IVF, KMEANS_PQ = 0, 1


class KMeansTrainer:
    """Owner of a vdb_kmeans handle: k-means over columns [col0, col0 + dsub) of the rows of a RowStore
    that stay in HBM.  The store must outlive the trainer and must not be written meanwhile."""

    def __init__(self, store: "RowStore", col0: int, dsub: int, n_clusters: int):
        self.lib = load()
        self.store = store                       # keeps the training rows alive
        self.k, self.dsub = int(n_clusters), int(dsub)
        self.n = len(store)
        self.h = C.c_void_p()
        check(self.lib.vdb_kmeans_create(C.byref(self.h), store.h, int(col0), self.dsub, self.k))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.vdb_kmeans_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init_rows(self, rows) -> None:
        r = np.ascontiguousarray(rows, dtype=np.int64)
        check(self.lib.vdb_kmeans_init_rows(self.h, r.ctypes.data_as(c_i64_p), len(r)))

    def init_pp(self, first_row: int, uniforms) -> np.ndarray:
        u = np.ascontiguousarray(uniforms, dtype=np.float64)
        if len(u) != self.k - 1:
            raise ValueError(f"k-means++ over {self.k} clusters takes {self.k - 1} uniform draws, got {len(u)}")
        out = np.empty(self.k, np.int64)
        check(self.lib.vdb_kmeans_init_pp(self.h, int(first_row), u.ctypes.data_as(c_double_p), out.ctypes.data_as(c_i64_p)))
        return out

    def set_centroids(self, centroids) -> None:
        c = f32(centroids)
        if c.shape != (self.k, self.dsub):
            raise ValueError(f"centroids shape {c.shape} != {(self.k, self.dsub)}")
        check(self.lib.vdb_kmeans_set_centroids(self.h, fptr(c)))

    def step(self, form: int):
        """One Lloyd step into the pending centroids -> (member counts int64 [k], sum((new - old)^2))."""
        counts = np.empty(self.k, np.int64)
        shift = C.c_double(0.0)
        check(self.lib.vdb_kmeans_step(self.h, int(form), counts.ctypes.data_as(c_i64_p), C.byref(shift)))
        return counts, shift.value

    def reseed(self, clusters, rows) -> float:
        c = np.ascontiguousarray(clusters, dtype=np.int32)
        r = np.ascontiguousarray(rows, dtype=np.int64)
        shift = C.c_double(0.0)
        check(self.lib.vdb_kmeans_reseed(self.h, c.ctypes.data_as(c_i32_p), r.ctypes.data_as(c_i64_p), len(c), C.byref(shift)))
        return shift.value

    def commit(self) -> None:
        check(self.lib.vdb_kmeans_commit(self.h))

    def centroids(self, float64: bool = False) -> np.ndarray:
        c32 = np.empty((self.k, self.dsub), np.float32)
        if not float64:
            check(self.lib.vdb_kmeans_centroids(self.h, fptr(c32), None))
            return c32
        c64 = np.empty((self.k, self.dsub), np.float64)
        check(self.lib.vdb_kmeans_centroids(self.h, fptr(c32), c64.ctypes.data_as(c_double_p)))
        return c64

    def assignments(self, with_distances: bool = False):
        a = np.empty(self.n, np.int32)
        if not with_distances:
            check(self.lib.vdb_kmeans_assignments(self.h, a.ctypes.data_as(c_i32_p), None))
            return a
        d = np.empty(self.n, np.float32)
        check(self.lib.vdb_kmeans_assignments(self.h, a.ctypes.data_as(c_i32_p), fptr(d)))
        return a, d


def pq_encode(x: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    init()
    x, b = f32(x), f32(codebook)
    m, ksub, dsub = b.shape
    codes = np.empty((len(x), m), np.uint8)
    check(load().vdb_pq_encode(fptr(x), len(x), fptr(b), m, ksub, dsub, codes.ctypes.data_as(c_u8_p)))
    return codes


class IvfLists:
    """Owner of a vdb_ivf handle (centroids + inverted lists in list order)."""

    def __init__(self, dim: int, metric: int):
        init()
        self.lib = load()
        self.dim = int(dim)
        self.h = C.c_void_p()
        self._mu = threading.Lock()
        check(self.lib.vdb_ivf_create(C.byref(self.h), self.dim, metric))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.vdb_ivf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_centroids(self, centroids: np.ndarray) -> None:
        c = f32(centroids)
        check(self.lib.vdb_ivf_set_centroids(self.h, fptr(c), len(c)))

    def build(self, rows: np.ndarray, list_of_row: np.ndarray) -> None:
        rows = f32(rows).reshape(-1, self.dim)
        a = np.ascontiguousarray(list_of_row, dtype=np.int32)
        check(self.lib.vdb_ivf_build(self.h, fptr(rows), len(rows), a.ctypes.data_as(c_i32_p)))

    def build_lists(self, rows: np.ndarray, list_rows: np.ndarray, list_offsets: np.ndarray) -> None:
        rows = f32(rows).reshape(-1, self.dim)
        lr = np.ascontiguousarray(list_rows, dtype=np.int64)
        lo = np.ascontiguousarray(list_offsets, dtype=np.int64)
        check(self.lib.vdb_ivf_build_lists(self.h, fptr(rows), len(rows), lr.ctypes.data_as(c_i64_p),
                                           lo.ctypes.data_as(c_i64_p)))

    def build_numbered(self, rows: np.ndarray, row_numbers: np.ndarray, list_offsets: np.ndarray) -> None:
        """Rows already in list order; ``row_numbers[i]`` is what searches return for row i and what a mask is indexed by."""
        rows = f32(rows).reshape(-1, self.dim)
        rn = np.ascontiguousarray(row_numbers, dtype=np.int64)
        lo = np.ascontiguousarray(list_offsets, dtype=np.int64)
        check(self.lib.vdb_ivf_build_numbered(self.h, fptr(rows), len(rows), rn.ctypes.data_as(c_i64_p),
                                              lo.ctypes.data_as(c_i64_p)))

    def append(self, rows: np.ndarray, row_numbers, lists) -> None:
        """New rows behind the last row of their lists; the rest of the index stays where it is."""
        rows = f32(rows).reshape(-1, self.dim)
        rn = np.ascontiguousarray(row_numbers, dtype=np.int64).reshape(-1)
        ls = np.ascontiguousarray(lists, dtype=np.int32).reshape(-1)
        if not (len(rows) == len(rn) == len(ls)):
            raise ValueError("rows, row numbers and lists must have one entry per row")
        check(self.lib.vdb_ivf_append(self.h, fptr(rows), rn.ctypes.data_as(c_i64_p), ls.ctypes.data_as(c_i32_p), len(rows)))

    def remove_at(self, list_no: int, position: int) -> None:
        """The list's last row takes the place of the row at ``position``."""
        check(self.lib.vdb_ivf_remove_at(self.h, int(list_no), int(position)))

    def layout(self):
        """(rows in lists, positions in use, positions lost to places that lists have left)"""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self.lib.vdb_ivf_layout(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def search(self, queries: np.ndarray, k: int, n_probe: int, mask_words=None, with_scan_rows: bool = False):
        """-> (distances, rows), and with ``with_scan_rows`` the number of list rows this search scanned (the two
        calls as one step, so that searches of other threads on this handle do not come between them)."""
        q = f32(queries)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        dist = np.empty((nq, k), np.float32)
        rows = np.empty((nq, k), np.int64)
        with self._mu:
            if isinstance(mask_words, DeviceMask):
                check(self.lib.vdb_ivf_search_masked(self.h, fptr(q), nq, k, n_probe, mask_words.h, fptr(dist),
                                                     rows.ctypes.data_as(c_i64_p)))
            else:
                check(self.lib.vdb_ivf_search(self.h, fptr(q), nq, k, n_probe, u32ptr(mask_words), fptr(dist),
                                              rows.ctypes.data_as(c_i64_p)))
            if with_scan_rows:
                return dist, rows, self.last_scan_rows()
        return dist, rows

    def search_device(self, q_ptr: int, nq: int, k: int, n_probe: int, out_dist_ptr: int, out_row_ptr: int,
                      mask_ptr: int = 0, stream: int = 0, out_order_ptr: int = 0) -> None:
        if out_order_ptr:
            check(self.lib.vdb_ivf_search_ordered_dev(self.h, C.c_void_p(q_ptr), nq, k, n_probe, C.c_void_p(mask_ptr or None),
                                                      C.c_void_p(out_dist_ptr), C.c_void_p(out_row_ptr),
                                                      C.c_void_p(out_order_ptr), C.c_void_p(stream or None)))
            return
        check(self.lib.vdb_ivf_search_dev(self.h, C.c_void_p(q_ptr), nq, k, n_probe, C.c_void_p(mask_ptr or None),
                                          C.c_void_p(out_dist_ptr), C.c_void_p(out_row_ptr), C.c_void_p(stream or None)))

    def probe(self, queries: np.ndarray, n_probe: int):
        q = f32(queries)
        if q.ndim == 1:
            q = q[None, :]
        lists = np.empty((q.shape[0], n_probe), np.int32)
        dist = np.empty((q.shape[0], n_probe), np.float32)
        check(self.lib.vdb_ivf_probe(self.h, fptr(q), q.shape[0], n_probe, lists.ctypes.data_as(c_i32_p), fptr(dist)))
        return lists, dist

    def last_scan_rows(self) -> int:
        n = C.c_int64(0)
        check(self.lib.vdb_ivf_last_scan_rows(self.h, C.byref(n)))
        return n.value

    def set_prefilter(self, mode: Optional[str]) -> None:
        """Batched Euclidean searches behind the matrix-core scan over a bf16 copy of the member rows: "auto" (the
        default: while the copy leaves a tenth of the device free), "bf16" (always) or None (never).  Same results."""
        if mode not in PREFILTER:
            raise ValueError(f"prefilter must be None, 'bf16' or 'auto', got {mode!r}")
        check(self.lib.vdb_ivf_set_prefilter(self.h, PREFILTER[mode]))

    def search_stats(self):
        """(used the matrix-core scan, a queue overflowed and the streaming scan repeated the batch) of the last search."""
        a, b = C.c_int(0), C.c_int(0)
        check(self.lib.vdb_ivf_search_stats(self.h, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)


class PqCodes:
    """Owner of a vdb_pq handle (codebook + resident codes)."""

    def __init__(self, dim: int, m: int, ksub: int):
        init()
        self.lib = load()
        self.dim, self.m, self.ksub = int(dim), int(m), int(ksub)
        self.h = C.c_void_p()
        check(self.lib.vdb_pq_create(C.byref(self.h), self.dim, self.m, self.ksub))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.vdb_pq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = C.c_int64(0)
        check(self.lib.vdb_pq_size(self.h, C.byref(n)))
        return n.value

    def set_codebook(self, codebook: np.ndarray) -> None:
        b = f32(codebook)
        if b.shape != (self.m, self.ksub, self.dim // self.m):
            raise ValueError(f"codebook shape {b.shape} != {(self.m, self.ksub, self.dim // self.m)}")
        check(self.lib.vdb_pq_set_codebook(self.h, fptr(b)))

    def append_codes(self, codes: np.ndarray) -> None:
        c = np.ascontiguousarray(codes, dtype=np.uint8).reshape(-1, self.m)
        check(self.lib.vdb_pq_append_codes(self.h, c.ctypes.data_as(c_u8_p), len(c)))

    def append_codes_device(self, ptr: int, n: int, stream: int = 0) -> None:
        check(self.lib.vdb_pq_append_codes_dev(self.h, C.c_void_p(ptr), n, C.c_void_p(stream or None)))

    def append_rows(self, rows: np.ndarray) -> np.ndarray:
        rows = f32(rows).reshape(-1, self.dim)
        codes = np.empty((len(rows), self.m), np.uint8)
        check(self.lib.vdb_pq_append_rows(self.h, fptr(rows), len(rows), codes.ctypes.data_as(c_u8_p)))
        return codes

    def truncate(self, n: int) -> None:
        check(self.lib.vdb_pq_truncate(self.h, int(n)))

    def set_code(self, row: int, code: np.ndarray) -> None:
        c = np.ascontiguousarray(code, dtype=np.uint8)
        check(self.lib.vdb_pq_set_code(self.h, int(row), c.ctypes.data_as(c_u8_p)))

    def lut(self, query: np.ndarray) -> np.ndarray:
        q = f32(query)
        out = np.empty((self.m, self.ksub), np.float32)
        check(self.lib.vdb_pq_lut(self.h, fptr(q), fptr(out)))
        return out

    def search(self, queries: np.ndarray, k: int, mask_words=None):
        q = f32(queries)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        dist = np.empty((nq, k), np.float32)
        rows = np.empty((nq, k), np.int64)
        if isinstance(mask_words, DeviceMask):
            check(self.lib.vdb_pq_search_masked(self.h, fptr(q), nq, k, mask_words.h, fptr(dist),
                                                rows.ctypes.data_as(c_i64_p)))
        else:
            check(self.lib.vdb_pq_search(self.h, fptr(q), nq, k, u32ptr(mask_words), fptr(dist),
                                         rows.ctypes.data_as(c_i64_p)))
        return dist, rows

    def search_stats(self):
        """(went through the byte-table pass, that pass gave up and the fp32 scan repeated the batch) of the last search."""
        a, b = C.c_int(0), C.c_int(0)
        check(self.lib.vdb_pq_search_stats(self.h, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)

    def search_device(self, q_ptr: int, nq: int, k: int, out_dist_ptr: int, out_row_ptr: int,
                      mask_ptr: int = 0, stream: int = 0) -> None:
        check(self.lib.vdb_pq_search_dev(self.h, C.c_void_p(q_ptr), nq, k, C.c_void_p(mask_ptr or None),
                                         C.c_void_p(out_dist_ptr), C.c_void_p(out_row_ptr), C.c_void_p(stream or None)))


class IvfPqLists:
    """Owner of a vdb_ivfpq handle (coarse centroids, residual codebook, codes in inverted lists)."""

    def __init__(self, dim: int, m: int, ksub: int):
        init()
        self.lib = load()
        self.dim, self.m, self.ksub = int(dim), int(m), int(ksub)
        self.h = C.c_void_p()
        check(self.lib.vdb_ivfpq_create(C.byref(self.h), self.dim, self.m, self.ksub))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.vdb_ivfpq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_quantizers(self, centroids: np.ndarray, codebook: np.ndarray) -> None:
        c, b = f32(centroids), f32(codebook)
        if b.shape != (self.m, self.ksub, self.dim // self.m):
            raise ValueError(f"codebook shape {b.shape} != {(self.m, self.ksub, self.dim // self.m)}")
        check(self.lib.vdb_ivfpq_set_quantizers(self.h, fptr(c), len(c), fptr(b)))

    def encode(self, rows: np.ndarray):
        rows = f32(rows).reshape(-1, self.dim)
        lists = np.empty(len(rows), np.int32)
        codes = np.empty((len(rows), self.m), np.uint8)
        check(self.lib.vdb_ivfpq_encode(self.h, fptr(rows), len(rows), lists.ctypes.data_as(c_i32_p),
                                        codes.ctypes.data_as(c_u8_p)))
        return lists, codes

    def build(self, codes: np.ndarray, list_rows: np.ndarray, list_offsets: np.ndarray) -> None:
        c = np.ascontiguousarray(codes, dtype=np.uint8).reshape(-1, self.m)
        lr = np.ascontiguousarray(list_rows, dtype=np.int64)
        lo = np.ascontiguousarray(list_offsets, dtype=np.int64)
        check(self.lib.vdb_ivfpq_build(self.h, c.ctypes.data_as(c_u8_p), len(c), lr.ctypes.data_as(c_i64_p),
                                       lo.ctypes.data_as(c_i64_p)))

    def build_numbered(self, codes: np.ndarray, row_numbers: np.ndarray, list_offsets: np.ndarray) -> None:
        """Codes already in list order, a row number of the caller's choice for each."""
        c = np.ascontiguousarray(codes, dtype=np.uint8).reshape(-1, self.m)
        rn = np.ascontiguousarray(row_numbers, dtype=np.int64)
        lo = np.ascontiguousarray(list_offsets, dtype=np.int64)
        check(self.lib.vdb_ivfpq_build_numbered(self.h, c.ctypes.data_as(c_u8_p), len(c), rn.ctypes.data_as(c_i64_p),
                                                lo.ctypes.data_as(c_i64_p)))

    def append(self, codes: np.ndarray, row_numbers, lists) -> None:
        c = np.ascontiguousarray(codes, dtype=np.uint8).reshape(-1, self.m)
        rn = np.ascontiguousarray(row_numbers, dtype=np.int64).reshape(-1)
        ls = np.ascontiguousarray(lists, dtype=np.int32).reshape(-1)
        if not (len(c) == len(rn) == len(ls)):
            raise ValueError("codes, row numbers and lists must have one entry per row")
        check(self.lib.vdb_ivfpq_append(self.h, c.ctypes.data_as(c_u8_p), rn.ctypes.data_as(c_i64_p),
                                        ls.ctypes.data_as(c_i32_p), len(c)))

    def remove_at(self, list_no: int, position: int) -> None:
        """The codes behind ``position`` move up by one, in their order."""
        check(self.lib.vdb_ivfpq_remove_at(self.h, int(list_no), int(position)))

    def layout(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self.lib.vdb_ivfpq_layout(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def search(self, queries: np.ndarray, k: int, n_probe: int, mask_words=None):
        q = f32(queries)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        dist = np.empty((nq, k), np.float32)
        rows = np.empty((nq, k), np.int64)
        if isinstance(mask_words, DeviceMask):
            check(self.lib.vdb_ivfpq_search_masked(self.h, fptr(q), nq, k, n_probe, mask_words.h, fptr(dist),
                                                   rows.ctypes.data_as(c_i64_p)))
        else:
            check(self.lib.vdb_ivfpq_search(self.h, fptr(q), nq, k, n_probe, u32ptr(mask_words), fptr(dist),
                                            rows.ctypes.data_as(c_i64_p)))
        return dist, rows

    def search_device(self, q_ptr: int, nq: int, k: int, n_probe: int, out_dist_ptr: int, out_row_ptr: int,
                      mask_ptr: int = 0, stream: int = 0) -> None:
        check(self.lib.vdb_ivfpq_search_dev(self.h, C.c_void_p(q_ptr), nq, k, n_probe, C.c_void_p(mask_ptr or None),
                                            C.c_void_p(out_dist_ptr), C.c_void_p(out_row_ptr), C.c_void_p(stream or None)))


def topk_merge_device(dist_ptr: int, row_ptr: int, n_lists: int, nq: int, k: int,
                      out_dist_ptr: int, out_row_ptr: int, stream: int = 0) -> None:
    init()
    check(load().vdb_topk_merge_dev(C.c_void_p(dist_ptr), C.c_void_p(row_ptr), n_lists, nq, k,
                                    C.c_void_p(out_dist_ptr), C.c_void_p(out_row_ptr), C.c_void_p(stream or None)))


def topk_merge_strided_device(dist_ptr: int, row_ptr: int, list_stride_dist: int, list_stride_row: int, row_base_ptr: int,
                              n_lists: int, nq: int, k: int, out_dist_ptr: int, out_row_ptr: int, stream: int = 0) -> None:
    """vdb_topk_merge_dev over lists `list_stride_*` elements apart, with row_base[l] (device int64, or 0 for none)
    added to the rows of list l."""
    init()
    check(load().vdb_topk_merge_strided_dev(C.c_void_p(dist_ptr), C.c_void_p(row_ptr), list_stride_dist, list_stride_row,
                                            C.c_void_p(row_base_ptr or None), n_lists, nq, k,
                                            C.c_void_p(out_dist_ptr), C.c_void_p(out_row_ptr), C.c_void_p(stream or None)))


def topk_merge_keyed_device(dist_ptr: int, key_ptr: int, row_ptr: int, n_lists: int, nq: int, k: int,
                            out_dist_ptr: int, out_row_ptr: int, stream: int = 0) -> None:
    init()
    check(load().vdb_topk_merge_keyed_dev(C.c_void_p(dist_ptr), C.c_void_p(key_ptr), C.c_void_p(row_ptr), n_lists, nq, k,
                                          C.c_void_p(out_dist_ptr), C.c_void_p(out_row_ptr), C.c_void_p(stream or None)))


def topk_merge_keyed_strided_device(dist_ptr: int, key_ptr: int, row_ptr: int, list_stride_dist: int, list_stride_key_row: int,
                                    n_lists: int, nq: int, k: int, out_dist_ptr: int, out_row_ptr: int, stream: int = 0) -> None:
    """topk_merge_keyed_device over lists `list_stride_*` elements apart (blocks of one packed all-gather)."""
    init()
    check(load().vdb_topk_merge_keyed_strided_dev(C.c_void_p(dist_ptr), C.c_void_p(key_ptr), C.c_void_p(row_ptr),
                                                  list_stride_dist, list_stride_key_row, n_lists, nq, k,
                                                  C.c_void_p(out_dist_ptr), C.c_void_p(out_row_ptr), C.c_void_p(stream or None)))
