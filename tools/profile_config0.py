#!/usr/bin/env python3
"""Where a single Collection.search call spends its host time (cProfile over 2000 calls, 10k x 128 rows)."""
import cProfile, pstats, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectordb_amd.core import Collection

np.random.seed(42)
x = np.random.randn(10000, 128).astype(np.float32)
q = np.random.randn(100, 128).astype(np.float32)
coll = Collection("c", 128, metric="euclidean")
coll.add_batch([{"id": f"v{i}", "vector": x[i]} for i in range(len(x))])
for v in q[:20]:
    coll.search(v, k=10)
pr = cProfile.Profile()
pr.enable()
for rep in range(20):
    for v in q:
        coll.search(v, k=10)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue())
