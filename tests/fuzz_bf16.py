#!/usr/bin/env python3
"""Randomised parity soak of the bf16 row store against the oracle on bf16-rounded inputs (not collected by
pytest):   python tests/fuzz_bf16.py --seconds 200 [--seed 0]
Row counts on both sides of the sample-pass threshold, dimensions with and without whole 128-byte K steps
(256 x 256 kernel, 128 x 128 kernels, list kernel), batches up to 600 queries, k up to 64, filters."""
import argparse, os, sys, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from vectordb_amd import _native as nat
from oracle import ref_numpy as R
from conftest import same_ids_or_near_ties
from test_gpu_cabi import assert_close, ids_atol, RTOL, _bf16_round

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
nat.init(0)
rng = np.random.default_rng(args.seed)
t0 = time.time()
case = 0
while time.time() - t0 < args.seconds:
    case += 1
    n = int(rng.choice([100, 5000, 65535, 65536, 65536 + 257, 100000, 150001]))
    d = int(rng.choice([8, 24, 64, 100, 128, 192, 256]))
    nq = int(rng.choice([4, 5, 64, 255, 256, 257, 300, 600]))
    k = int(rng.choice([1, 10, 32, 64]))
    metric = str(rng.choice(["euclidean", "euclidean_squared", "cosine", "dot"]))
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((nq, d), dtype=np.float32)
    mask = (rng.random(n) < 0.3) if rng.random() < 0.25 else None
    xr, qr = _bf16_round(x), _bf16_round(q)
    label = f"case {case}: n={n} d={d} nq={nq} k={k} {metric} mask={'none' if mask is None else int(mask.sum())}"
    store = nat.RowStore(d, storage="bf16")
    store.append(x)
    dist, rows = store.search(q, k, nat.METRIC_CODES[metric], nat.pack_mask(mask))
    stats = store.search_stats()
    store.close()
    try:
        for i in sorted(set([0, nq - 1, int(rng.integers(0, nq)), min(nq - 1, 256)])):
            ref_rows, ref_dist = R.flat_search(qr[i], xr, k, metric, mask=mask)
            full = R.query_distances(qr[i], xr, metric)
            m = len(ref_rows)
            assert (rows[i, m:] == -1).all()
            atol = ids_atol(metric, qr[i], xr, float(ref_dist[-1]))
            assert same_ids_or_near_ties(rows[i, :m], ref_rows, lambda r: full[r], float(ref_dist[-1]), rtol=RTOL, atol=atol)
            assert_close(metric, dist[i, :m], ref_dist, qr[i], xr)
    except Exception:       # noqa: BLE001
        print("FAILED", label, "seed", args.seed, flush=True)
        raise
    print("ok", label, "queue" if stats[0] else "lists", flush=True)
print(f"{case} cases, no failure")
