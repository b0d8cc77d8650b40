import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from vectordb_amd.index import PQIndex
rng = np.random.default_rng(31)
n, d, m = 1_000_000, 128, 16
x = rng.standard_normal((n, d), dtype=np.float32)
ix = PQIndex(dimension=d, n_subvectors=m, n_bits=8, n_iter=5, seed=1)
ix.train(x[:20_000])
ix.add_batch([f"r{i}" for i in range(n)], x, [{"cat": i % 100} for i in range(n)])
q = rng.standard_normal(d, dtype=np.float32)
f = lambda _id, meta: meta["cat"] == 7
ix.search(q, 10); ix.search(q, 10, filter_fn=f)
for name, kw in (("plain", {}), ("filtered", {"filter_fn": f}), ("plain", {}), ("plain no metadata", {"include_metadata": False})):
    t0 = time.perf_counter()
    for _ in range(50): ix.search(q, 10, **kw)
    print(name, (time.perf_counter() - t0) / 50 * 1e3, "ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): ix.search(q, 10)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
t0 = time.perf_counter()
for _ in range(50): ix._dev.search(q, 10, None)
print("dev.search", (time.perf_counter() - t0) / 50 * 1e3)
