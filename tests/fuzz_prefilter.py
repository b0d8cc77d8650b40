#!/usr/bin/env python3
"""Randomised soak of the bf16 filter pass of the Flat search (not collected by pytest):
    python tests/fuzz_prefilter.py --seconds 300 [--seed 0]
Every case builds one fp32 row store and searches it twice, with the filter off and on; distances (as bit
patterns), rows and their order must be identical.  Row counts around the tile and sample thresholds (run with
VDB_SAMPLE_MIN_TILES=32 to take the queue kernels from 4096 rows on), dimensions 1-300, batches 4-300, k 1-32, four
metrics, filters, and data chosen to stress the rounding slack: Gaussian rows, tight clusters around the queries,
row norms over several decades, integer grids (exact ties by the hundred), duplicated and zero rows.  Prints one
line per case and stops at the first difference with the case's parameters."""
import argparse, os, sys, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from vectordb_amd import _native as nat

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
nat.init(0)
rng = np.random.default_rng(args.seed)
metrics = ["euclidean", "euclidean_squared", "cosine", "dot"]
min_tiles = int(os.environ.get("VDB_SAMPLE_MIN_TILES", "512"))
n_min = min_tiles * 128
t0 = time.time()
case = used = 0
while time.time() - t0 < args.seconds:
    case += 1
    kind = int(rng.integers(0, 3))
    if kind == 0:
        n = n_min + int(rng.choice([0, 1, 127, 128, 129, 255, 256, 257, 1000]))
    elif kind == 1:
        n = int(rng.integers(n_min, 3 * n_min))
    else:
        n = int(rng.integers(max(1, n_min - 300), n_min + 300))
    d = int(rng.choice([1, 2, 3, 7, 8, 16, 31, 32, 33, 48, 63, 64, 65, 96, 100, 128, 130, 200, 256, 300]))
    if n * d > 40_000_000:
        d = 32
    nq = int(rng.choice([4, 5, 8, 9, 17, 64, 127, 128, 129, 255, 256, 257, 300]))
    k = int(rng.choice([1, 2, 5, 10, 31, 32]))
    metric = metrics[int(rng.integers(0, 4))]
    style = int(rng.integers(0, 6))
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((nq, d), dtype=np.float32)
    if style == 1:        # tight clusters around some of the queries
        for i in range(min(nq, 6)):
            at = rng.choice(n, size=min(n, int(rng.integers(20, 500))), replace=False)
            x[at] = (float(rng.uniform(0.5, 3.0)) * q[i] + float(10.0 ** rng.uniform(-5, -1)) * rng.standard_normal((len(at), d))).astype(np.float32)
    elif style == 2:      # row norms over several decades
        x *= (10.0 ** rng.uniform(-3, 3, size=(n, 1))).astype(np.float32)
    elif style == 3:      # integer grid: exact ties
        x = rng.integers(-2, 3, size=(n, d)).astype(np.float32)
        q = rng.integers(-2, 3, size=(nq, d)).astype(np.float32)
    elif style == 4:      # duplicated rows, zero rows, a zero query
        src = rng.choice(n, size=min(n, 200), replace=False)
        x[rng.choice(n, size=len(src), replace=False)] = x[src]
        x[rng.choice(n, size=min(n, 5), replace=False)] = 0.0
        q[0] = 0.0
    elif style == 5:      # common scale far from one
        s = float(10.0 ** rng.uniform(-12, 12))
        x *= np.float32(s)
        q *= np.float32(10.0 ** rng.uniform(-6, 6))
    mask = None
    if rng.random() < 0.3:
        mask = rng.random(n) < float(rng.choice([0.9, 0.5, 0.05, 0.002]))
    words = nat.pack_mask(mask)
    code = nat.METRIC_CODES[metric]
    store = nat.RowStore(d)
    store.append(x)
    store.set_prefilter(None)
    d0, r0 = store.search(q, k, code, words)
    store.set_prefilter("bf16")
    d1, r1 = store.search(q, k, code, words)
    filt = store.used_prefilter()
    over = store.search_stats()[1]
    store.close()
    used += bool(filt)
    same = np.array_equal(r0, r1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32))
    print(f"case {case}: n={n} d={d} nq={nq} k={k} {metric} style={style} mask={'y' if mask is not None else 'n'} "
          f"filter={int(filt)} overflow={int(over)} {'ok' if same else 'DIFFERENT'}", flush=True)
    if not same:
        bad = np.flatnonzero((r0 != r1).any(axis=1) | (d0.view(np.uint32) != d1.view(np.uint32)).any(axis=1))
        print("first differing query", int(bad[0]), r0[bad[0]], r1[bad[0]], d0[bad[0]], d1[bad[0]])
        sys.exit(1)
print(f"{case} cases, {used} through the filter pass, no failure (every result identical) in {time.time() - t0:.0f} s")
