import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
for nq in (8, 1, 8):
    q = torch.randn((nq, d), device=dev, generator=g)
    od = torch.empty((nq, 10), device=dev); orow = torch.empty((nq, 10), device=dev, dtype=torch.int64)
    line = []
    for rep in range(16):
        N.profile_read(); N.profile_enable(True)
        for _ in range(10): pq.search_device(q.data_ptr(), nq, 10, od.data_ptr(), orow.data_ptr())
        torch.cuda.synchronize()
        N.profile_enable(False); ms, cnt = N.profile_read()
        line.append("%.3f" % (ms / max(cnt, 1)))
    print(nq, " ".join(line), flush=True)
    time.sleep(0.5)
