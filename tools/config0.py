#!/usr/bin/env python3
"""BASELINE configs[0] through this package: Flat, 10k x 128 fp32, euclidean, 100 queries one at a
time, k = 10, through Collection.search (the reference's entry point A) and FlatIndex.search (B)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd.core import Collection
from vectordb_amd.index import FlatIndex

np.random.seed(42)
x = np.random.randn(10000, 128).astype(np.float32)
np.random.seed(123)
q = np.random.randn(100, 128).astype(np.float32)
coll = Collection("c", 128, metric="euclidean")
coll.add_batch([{"id": f"v{i}", "vector": x[i]} for i in range(len(x))])
flat = FlatIndex(dimension=128)
flat.add_batch([f"v{i}" for i in range(len(x))], x)
out = {}
for name, fn in (("collection_search", lambda v: coll.search(v, k=10)), ("flat_index_search", lambda v: flat.search(v, k=10))):
    for v in q[:10]:
        fn(v)
    t0 = time.perf_counter()
    for v in q:
        fn(v)
    dt = time.perf_counter() - t0
    out[name] = {"qps": round(100 / dt, 1), "ms_per_query": round(dt * 10, 4)}
t0 = time.perf_counter()
coll.search_batch(q, k=10)
out["collection_search_batch_100"] = {"qps": round(100 / (time.perf_counter() - t0), 1)}
print(json.dumps(out))
