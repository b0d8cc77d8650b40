import sys, numpy as np, torch
sys.path.insert(0, '.')
from vectordb_amd import _native as nat
nat.init(0)
dev = torch.device("cuda", 0)
rng = np.random.RandomState(0)
shards, nq, k = 8, 33, 10
d = np.sort(rng.rand(shards, nq, k).astype(np.float32), axis=2)
r = rng.randint(0, 100000, size=(shards, nq, k)).astype(np.int64)
gd = torch.from_numpy(d).to(dev); gr = torch.from_numpy(r).to(dev)
od = torch.empty((nq, k), device=dev, dtype=torch.float32); orow = torch.empty((nq, k), device=dev, dtype=torch.int64)
nat.topk_merge_device(gd.data_ptr(), gr.data_ptr(), shards, nq, k, od.data_ptr(), orow.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
od = od.cpu().numpy(); orow = orow.cpu().numpy()
bad = 0
for i in range(nq):
    dd = d[:, i, :].reshape(-1); rr = r[:, i, :].reshape(-1)
    o = np.lexsort((rr, dd))[:k]
    if not (np.array_equal(orow[i], rr[o]) and np.array_equal(od[i], dd[o])): bad += 1; print(i, orow[i], rr[o])
print("bad", bad)
