import ctypes as C, numpy as np, torch, sys
sys.path.insert(0, ".")
from vectordb_amd import _native as N
N.init(0)
lib = N.load()
n, d, nlist, nprobe = 2_000_000, 128, 4096, 32
rng = np.random.default_rng(42)
x = rng.standard_normal((n, d), dtype=np.float32)
cent = x[rng.choice(n, nlist, replace=False)].copy()
assign, _ = N.assign(x, cent, N.EUCLIDEAN)
lists = N.IvfLists(d, N.EUCLIDEAN)
lists.set_centroids(cent)
lists.build(x, assign)
q = torch.randn(1, d, device="cuda")
od = torch.empty(1, 10, device="cuda"); orow = torch.empty(1, 10, dtype=torch.int64, device="cuda")
for _ in range(5):
    lists.search_device(q.data_ptr(), 1, 10, nprobe, od.data_ptr(), orow.data_ptr())
torch.cuda.synchronize()
buf = (C.c_longlong * 32)()
lib.vdb_debug_read.argtypes = [C.c_void_p]
lib.vdb_debug_read(buf)
t = list(buf)
names = ["norm", "wave lists", "store+sync", "rank merge", "scan", "offsets+items"]
print("plan:", ", ".join("%s %.1f" % (names[i], (t[11 + i] - t[10 + i]) / 100) for i in range(6)), "us")
