#!/usr/bin/env python3
"""Single-query latency through the C ABI with host buffers (what FlatIndex.search / IVFIndex.search /
PQIndex.search call): mean milliseconds per query over 300 searches of one query each.
    python tools/latency.py
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd import _native as N

N.init(0)
rng = np.random.default_rng(1)
out = {}


def loop(fn, q, n=300):
    for i in range(20):
        fn(q[i % len(q)])
    t0 = time.perf_counter()
    for i in range(n):
        fn(q[i % len(q)])
    return round((time.perf_counter() - t0) / n * 1e3, 4)


n, d = 1_000_000, 128
x = rng.standard_normal((n, d), dtype=np.float32)
q = rng.standard_normal((64, d), dtype=np.float32)
store = N.RowStore(d)
store.append(x)
out["flat_cosine_1Mx128_ms"] = loop(lambda v: store.search(v, 10, N.COSINE), q)
out["flat_manhattan_1Mx128_ms"] = loop(lambda v: store.search(v, 10, N.MANHATTAN), q)
out["flat_cosine_batch8_ms"] = loop(lambda v: store.search(q[:8], 10, N.COSINE), q)
store.close()

nlist = 1024
cent = x[:nlist].copy()
a, _ = N.assign(x, cent, N.EUCLIDEAN)
L = N.IvfLists(d, N.EUCLIDEAN)
L.set_centroids(cent)
L.build(x, a)
out["ivf_1Mx128_nlist1024_nprobe32_ms"] = loop(lambda v: L.search(v, 10, 32), q)
L.close()
print(json.dumps(out))
