"""A/B of one Flat batch search with the bf16 filter pass off and on (same handle, device-resident inputs):
    python tools/ab_prefilter.py [n] [d] [metric] [nq]
Prints ms per batch for both and whether the results are identical."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from vectordb_amd import _native as N


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
    metric = sys.argv[3] if len(sys.argv) > 3 else "cosine"
    nq = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    k = 10
    N.init(0)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    store = N.RowStore(d)
    for o in range(0, n, 250_000):
        c = min(250_000, n - o)
        x = torch.randn(c, d, device=dev, generator=g)
        torch.cuda.synchronize()
        store.append_device(x.data_ptr(), c, d)
        torch.cuda.synchronize()
    q = torch.randn(nq, d, device=dev, generator=g)
    od = torch.empty(nq, k, device=dev)
    oi = torch.empty(nq, k, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    out = {}
    for mode in (None, "bf16"):
        store.set_prefilter(mode)
        for _ in range(3):
            store.search_device(q.data_ptr(), nq, k, N.METRIC_CODES[metric], od.data_ptr(), oi.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            store.search_device(q.data_ptr(), nq, k, N.METRIC_CODES[metric], od.data_ptr(), oi.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        out[mode] = (od.clone(), oi.clone())
        print(f"prefilter={mode}: {ms:.3f} ms per batch of {nq} ({nq / ms * 1e3:.0f} q/s), stats={store.search_stats()} "
              f"filter={store.used_prefilter()}", flush=True)
    same = torch.equal(out[None][0].view(torch.int32), out["bf16"][0].view(torch.int32)) and torch.equal(out[None][1], out["bf16"][1])
    print("identical:", same)


if __name__ == "__main__":
    main()
