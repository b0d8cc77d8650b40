import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vectordb_amd import _native as N
N.init(0)
dev = torch.device('cuda', 0)
n, m, d = 100_000_000, 16, 128
pq = N.PqCodes(d, m, 256)
rng = np.random.default_rng(7)
pq.set_codebook(rng.standard_normal((m, 256, d // m), dtype=np.float32))
for s in range(0, n, 10_000_000):
    codes = torch.randint(0, 256, (10_000_000, m), device=dev, dtype=torch.uint8)
    pq.append_codes_device(codes.data_ptr(), 10_000_000, 0)
    torch.cuda.synchronize()
del codes
g = torch.Generator(device=dev); g.manual_seed(1)
for nq in [int(a) for a in (sys.argv[1].split(',') if len(sys.argv) > 1 else '1,2,4,8,16,64'.split(','))]:
    q = torch.randn((nq, d), device=dev, generator=g)
    od = torch.empty((nq, 10), device=dev); orow = torch.empty((nq, 10), device=dev, dtype=torch.int64)
    for mode in (("bytes",) if len(sys.argv) > 2 else ("bytes", "fp32")):
        if mode == "fp32": os.environ["VDB_PQ_NO_BYTE_TABLES_NOW"] = "1"
        else: os.environ.pop("VDB_PQ_NO_BYTE_TABLES_NOW", None)
        for _ in range(3): pq.search_device(q.data_ptr(), nq, 10, od.data_ptr(), orow.data_ptr())
        torch.cuda.synchronize()
        N.profile_read(); N.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(10): pq.search_device(q.data_ptr(), nq, 10, od.data_ptr(), orow.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        N.profile_enable(False); ms, cnt = N.profile_read()
        print(nq, mode, "ms/batch %.4f" % (dt * 1e3), "kernel ms %.4f x %d" % (ms / max(cnt, 1), cnt // 10), pq.search_stats(), flush=True)
        if mode == "bytes": keep = (od.clone(), orow.clone())
        else: print("   same:", torch.equal(keep[0].view(torch.int32), od.view(torch.int32)) and torch.equal(keep[1], orow))
