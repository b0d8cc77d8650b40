#!/usr/bin/env python3
"""Randomised parity soak of the Flat search against the CPU oracle (not collected by pytest):
    python tests/fuzz_flat.py --seconds 300 [--seed 0]
Random row counts (small sets, sets around the tile and sample-pass thresholds, large sets), dimensions,
batch sizes, k, metrics and filter densities go through check_search of the C-ABI tests: ids equal up to
ties at the k-th place, distances to 1e-4, Manhattan and Chebyshev bit for bit.  Prints one line per case
and stops at the first failure with the case's parameters."""
import argparse, os, sys, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from vectordb_amd import _native as nat
from test_gpu_cabi import check_search

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
nat.init(0)
rng = np.random.default_rng(args.seed)
metrics = ["euclidean", "euclidean_squared", "cosine", "dot", "manhattan", "chebyshev"]
t0 = time.time()
case = 0
while time.time() - t0 < args.seconds:
    case += 1
    kind = rng.integers(0, 4)
    if kind == 0:
        n = int(rng.integers(1, 600))
    elif kind == 1:
        n = int(rng.choice([255, 256, 257, 2047, 2048, 4096, 65535, 65536, 65537, 65536 + 129]))
    elif kind == 2:
        n = int(rng.integers(600, 70000))
    else:
        n = int(rng.integers(70000, 200000))
    d = int(rng.choice([1, 2, 3, 7, 8, 16, 31, 32, 33, 48, 64, 96, 100, 128, 130, 200, 256]))
    if n * d > 30_000_000:
        d = 32
    nq = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 17, 64, 127, 128, 129, 200]))
    k = int(rng.choice([1, 2, 5, 10, 32, 33, 64, 100]))
    metric = metrics[int(rng.integers(0, len(metrics)))]
    x = rng.standard_normal((n, d), dtype=np.float32)
    if rng.random() < 0.3 and n > 10:                      # duplicates: exact ties
        src = rng.integers(0, n, size=5)
        dst = rng.integers(0, n, size=5)
        x[dst] = x[src]
    q = rng.standard_normal((nq, d), dtype=np.float32)
    if rng.random() < 0.3:
        q[0] = x[int(rng.integers(0, n))]
    mask = None
    r = rng.random()
    if r < 0.2:
        mask = rng.random(n) < 0.5
    elif r < 0.3:
        mask = rng.random(n) < min(1.0, 20.0 / n)
    # the oracle walks the queries one by one: keep its share of the time bounded
    nq_check = min(nq, max(1, int(4_000_000 // (n * max(d, 8))) ))
    label = f"case {case}: n={n} d={d} nq={nq} (checked {nq_check}) k={k} {metric} mask={'none' if mask is None else int(mask.sum())}"
    try:
        if nq_check < nq:
            # the whole batch runs on the GPU; the first nq_check queries are compared
            store = nat.RowStore(d); store.append(x)
            dist_all, rows_all = store.search(q, k, nat.METRIC_CODES[metric], nat.pack_mask(mask)); store.close()
            dist, rows = check_search(nat, x, q[:nq_check], k, metric, mask=mask)
            if metric in ("manhattan", "chebyshev"):
                assert (rows_all[:nq_check] == rows).all() and (dist_all[:nq_check] == dist).all()
        else:
            check_search(nat, x, q, k, metric, mask=mask)
    except Exception as e:      # noqa: BLE001
        print("FAILED", label, "seed", args.seed, flush=True)
        raise
    print("ok", label, flush=True)
print(f"{case} cases, no failure")
