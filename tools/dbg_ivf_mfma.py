import sys, numpy as np
sys.path.insert(0, '.')
from oracle import ref_numpy as R
from vectordb_amd import _native as nat
nat.init(0)
for d in (8, 32, 128):
    rng = np.random.RandomState(100 + d)
    n = 20000
    x = rng.randn(n, d).astype(np.float32)
    cent = x[rng.choice(n, 64, replace=False)].copy()
    q = rng.randn(70, d).astype(np.float32)
    assign = R.ivf_assign_batch(x, cent, "euclidean").astype(np.int32)
    lists = nat.IvfLists(d, nat.EUCLIDEAN); lists.set_centroids(cent); lists.build(x, assign)
    lists.set_prefilter("bf16")
    members = [np.flatnonzero(assign == l).astype(np.int64) for l in range(64)]
    dist, rows = lists.search(q, 10, 8)
    print(d, lists.search_stats())
    bad = 0
    for i in range(len(q)):
        rr, rd = R.ivf_search(q[i], x, cent, members, 10, 8, "euclidean")
        if not (np.array_equal(rows[i], rr) and np.array_equal(dist[i], rd.astype(np.float32))): bad += 1
    print("mismatching queries:", bad)
