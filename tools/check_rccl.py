#!/usr/bin/env python3
"""One-rank RCCL rehearsal of the sharded search's collective: the packed all-gather of uint8 blocks and
the in-place merge, through a real process group (world size 1 on a one-GPU box).
    python tools/check_rccl.py
"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
from vectordb_amd import _native
from vectordb_amd.sharded import HipFlatBackend, ShardedFlat
_native.init(0)

n, d, nq, k = 70000, 32, 33, 10
g = torch.Generator(device=dev); g.manual_seed(1)
rows = torch.randn((n, d), device=dev, generator=g)
q = torch.randn((nq, d), device=dev, generator=g)
be = HipFlatBackend(d, _native.COSINE, prefilter=None)
be.add_device(rows)
single = ShardedFlat(be, n, 0, 1)
sd, sr = single.search(q, k)
sd, sr = sd.clone(), sr.clone()
# the world>1 code path with one rank: gather of one block, merge of one list
idx = ShardedFlat(be, n, 0, 1)
idx.world = 1
local, gathered, ld, lr, gd, gr, base, od, orow = idx._buffers(nq, k, dev)
be.local_search(q, k, ld, lr)
dist.all_gather_into_tensor(gathered, local)
be.merge(gd, gr, k, od, orow, base)
torch.cuda.synchronize()
assert torch.equal(orow, sr) and torch.equal(od, sd)
print("rccl all_gather_into_tensor(uint8) + packed merge ok:", tuple(gathered.shape), gathered.dtype)
dist.destroy_process_group()
