#!/usr/bin/env python3
"""k-means++ at configs[3] size with the rounds over the bf16 copy and with the float32 rounds: the same rows must be drawn.
    python tools/check_kmpp_two_ways.py [n] [d] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vectordb_amd import _native as nat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
k = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
nat.init(0)
rng = np.random.RandomState(5)
x = rng.randn(n, d).astype(np.float32)
u = rng.random_sample(k - 1)
store = nat.RowStore(d)
store.append(x)
tr = nat.KMeansTrainer(store, 0, d, k)
out = []
for mode in ("bf16", "f32"):
    if mode == "f32":
        os.environ["VDB_KMPP_NO_BF16"] = "1"
    tr.init_pp(7, u)
    t0 = time.perf_counter()
    rows = tr.init_pp(7, u)
    print(mode, "%.3f s" % (time.perf_counter() - t0), flush=True)
    out.append(rows)
same = np.array_equal(out[0], out[1])
print("same rows:", same, "" if same else "first difference at draw %d" % int(np.flatnonzero(out[0] != out[1])[0]))
sys.exit(0 if same else 1)
