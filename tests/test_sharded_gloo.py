"""The row-sharded search path with world_size 2 over gloo on CPU.

The per-shard search and the merge are GPU kernels in production; here the
oracle stands in for both so that what is tested is the distributed plumbing:
the row partition, local -> global row numbers, the all-gather layout
[rank][query][k], and that the merged result equals the single-shard result
including the tie rule (equal distances -> lower global row)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_numpy as R
from vectordb_amd.sharded import ShardedFlat, shard_range


class OracleBackend:
    def __init__(self, rows, metric):
        self.rows, self.metric = rows, metric

    def local_search(self, queries, k, out_dist, out_rows):
        out_dist.fill_(float("inf"))
        out_rows.fill_(-1)
        for i, q in enumerate(queries.numpy()):
            r, d = R.flat_search(q, self.rows, k, self.metric)
            out_rows[i, :len(r)] = torch.from_numpy(r)
            out_dist[i, :len(r)] = torch.from_numpy(np.asarray(d, dtype=np.float32))

    def merge(self, all_dist, all_rows, k, out_dist, out_rows):
        world, nq, _ = all_dist.shape
        d = all_dist.permute(1, 0, 2).reshape(nq, -1).numpy()
        r = all_rows.permute(1, 0, 2).reshape(nq, -1).numpy()
        for i in range(nq):
            valid = r[i] >= 0
            order = np.lexsort((r[i][valid], d[i][valid]))[:k]
            out_dist[i].fill_(float("inf")); out_rows[i].fill_(-1)
            out_dist[i, :len(order)] = torch.from_numpy(d[i][valid][order])
            out_rows[i, :len(order)] = torch.from_numpy(r[i][valid][order])


def make_data():
    rng = np.random.RandomState(17)
    x = rng.randn(1001, 24).astype(np.float32)
    x[700] = x[20]                 # an exact tie across the two shards
    x[999] = x[20]
    q = rng.randn(9, 24).astype(np.float32)
    q[0] = x[20]
    return x, q


def worker(rank, world, port, metric, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, q = make_data()
        lo, hi = shard_range(len(x), world, rank)
        index = ShardedFlat(OracleBackend(x[lo:hi], metric), len(x), rank, world)
        d, r = index.search(torch.from_numpy(q), k)
        ret[rank] = (d.numpy().copy(), r.numpy().copy())
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_ranges_cover_rows():
    for n in (0, 1, 7, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("metric,k", [("manhattan", 10), ("euclidean", 5), ("manhattan", 600)])
def test_two_ranks_equal_single_shard(metric, k):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), metric, k, ret), nprocs=world, join=True)
    x, q = make_data()
    for rank in range(world):
        d, r = ret[rank]
        for i in range(len(q)):
            ref_r, ref_d = R.flat_search(q[i], x, k, metric)
            n = len(ref_r)
            if metric == "manhattan":
                np.testing.assert_array_equal(r[i, :n], ref_r)          # exact path: same rows, same order, ties by global row
                np.testing.assert_array_equal(d[i, :n], ref_d)
            else:
                assert set(r[i, :n]) == set(ref_r)
                np.testing.assert_allclose(d[i, :n], ref_d, rtol=1e-5, atol=1e-5)
            assert (r[i, n:] == -1).all()
    assert list(ret[0][1][0][:3]) == [20, 700, 999]
