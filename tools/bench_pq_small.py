#!/usr/bin/env python3
"""Single-query and small-batch PQ latency over code sets of different sizes (device pointers, 100 calls each)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vectordb_amd import _native as N
N.init(0)
dev = torch.device('cuda', 0)
m, d = 16, 128
rng = np.random.default_rng(7)
book = rng.standard_normal((m, 256, d // m), dtype=np.float32)
g = torch.Generator(device=dev); g.manual_seed(1)
for n in (100_000, 1_000_000, 4_000_000, 10_000_000):
    pq = N.PqCodes(d, m, 256)
    pq.set_codebook(book)
    codes = torch.randint(0, 256, (n, m), device=dev, dtype=torch.uint8)
    pq.append_codes_device(codes.data_ptr(), n, 0)
    torch.cuda.synchronize()
    for nq in (1, 8):
        q = torch.randn((nq, d), device=dev, generator=g)
        od = torch.empty((nq, 10), device=dev); orow = torch.empty((nq, 10), device=dev, dtype=torch.int64)
        for _ in range(5): pq.search_device(q.data_ptr(), nq, 10, od.data_ptr(), orow.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): pq.search_device(q.data_ptr(), nq, 10, od.data_ptr(), orow.data_ptr())
        torch.cuda.synchronize()
        print(f"n={n} nq={nq}: {(time.perf_counter() - t0) * 10:.4f} ms/batch  bytes/approx path: {pq.search_stats()}", flush=True)
    pq.close()
