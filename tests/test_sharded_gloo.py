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

    def merge(self, all_dist, all_rows, k, out_dist, out_rows, row_base=None):
        world, nq, _ = all_dist.shape
        if row_base is not None:            # local -> global row numbers; -1 stays -1
            all_rows = torch.where(all_rows >= 0, all_rows + row_base.view(-1, 1, 1), all_rows)
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


# --------------------------------------------------------------------------- IVF lists over two ranks
from vectordb_amd.sharded import ShardedIVF, assign_lists_balanced, local_ivf_layout, kmeans_update_allreduce


class OracleIvfBackend:
    """The oracle's IVF search over the lists one rank owns; keys = probe rank << 32 | position."""

    def __init__(self, rows, centroids, members, owner, rank, metric):
        self.rows, self.centroids, self.metric = rows, centroids, metric
        self.members = [m if owner[l] == rank else np.zeros(0, dtype=np.int64) for l, m in enumerate(members)]

    def local_search(self, queries, k, n_probe, out_dist, out_rows, out_order):
        out_dist.fill_(float("inf")); out_rows.fill_(-1); out_order.fill_(-1)
        for i, q in enumerate(queries.numpy()):
            probes = R.ivf_probe_order(q, self.centroids, self.metric, n_probe)
            cand, keys = [], []
            for p, l in enumerate(probes):
                cand.append(self.members[l])
                keys.append((np.int64(p) << 32) | np.arange(len(self.members[l]), dtype=np.int64))
            cand, keys = np.concatenate(cand), np.concatenate(keys)
            d = R.rows_scalar_metric(q, self.rows[cand], self.metric).astype(np.float32)
            order = np.argsort(d, kind="stable")[:k]
            n = len(order)
            out_dist[i, :n] = torch.from_numpy(d[order])
            out_rows[i, :n] = torch.from_numpy(cand[order])
            out_order[i, :n] = torch.from_numpy(keys[order])

    def merge(self, all_dist, all_order, all_rows, k, out_dist, out_rows):
        world, nq, _ = all_dist.shape
        d = all_dist.permute(1, 0, 2).reshape(nq, -1).numpy()
        o = all_order.permute(1, 0, 2).reshape(nq, -1).numpy()
        r = all_rows.permute(1, 0, 2).reshape(nq, -1).numpy()
        for i in range(nq):
            valid = r[i] >= 0
            order = np.lexsort((o[i][valid], d[i][valid]))[:k]
            out_dist[i].fill_(float("inf")); out_rows[i].fill_(-1)
            out_dist[i, :len(order)] = torch.from_numpy(d[i][valid][order])
            out_rows[i, :len(order)] = torch.from_numpy(r[i][valid][order])


def make_ivf_data():
    rng = np.random.RandomState(5)
    x = rng.randn(1500, 16).astype(np.float32)
    x[900] = x[30]                       # equal distances in one list
    cent = x[rng.choice(len(x), 24, replace=False)].copy()
    assign = R.ivf_assign_batch(x, cent, "euclidean")
    members = [np.flatnonzero(assign == l).astype(np.int64) for l in range(len(cent))]
    q = rng.randn(7, 16).astype(np.float32)
    q[0] = x[30]
    return x, cent, assign, members, q


def ivf_worker(rank, world, port, k, n_probe, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, cent, assign, members, q = make_ivf_data()
        owner = assign_lists_balanced([len(m) for m in members], world)
        index = ShardedIVF(OracleIvfBackend(x, cent, members, owner, rank, "euclidean"), rank, world)
        d, r = index.search(torch.from_numpy(q), k, n_probe)
        # one k-means update with the rows split in halves
        lo, hi = shard_range(len(x), world, rank)
        sums = torch.zeros((len(cent), x.shape[1]), dtype=torch.float64)
        counts = torch.zeros(len(cent), dtype=torch.int64)
        sums.index_add_(0, torch.from_numpy(assign[lo:hi].astype(np.int64)), torch.from_numpy(x[lo:hi]).double())
        counts.index_add_(0, torch.from_numpy(assign[lo:hi].astype(np.int64)), torch.ones(hi - lo, dtype=torch.int64))
        new, total = kmeans_update_allreduce(sums, counts, torch.from_numpy(cent))
        ret[rank] = (d.numpy().copy(), r.numpy().copy(), new.numpy().copy(), total.numpy().copy())
    finally:
        dist.destroy_process_group()


def test_lists_balanced_and_layout():
    lengths = [50, 10, 40, 30, 20, 0, 5]
    owner = assign_lists_balanced(lengths, 3)
    load = [sum(l for l, o in zip(lengths, owner) if o == r) for r in range(3)]
    assert sum(load) == sum(lengths) and max(load) - min(load) <= 10
    list_of_row = np.array([0, 1, 0, 2, 1, 0, 2, 2])
    own = np.array([0, 1, 0])
    rows0, lr0, off0 = local_ivf_layout(list_of_row, 3, own, 0)
    assert rows0.tolist() == [0, 2, 3, 5, 6, 7]                       # global rows of lists 0 and 2
    assert off0.tolist() == [0, 3, 3, 6]
    assert rows0[lr0].tolist() == [0, 2, 5, 3, 6, 7]                  # list 0 then list 2, insertion order inside


@pytest.mark.parametrize("k,n_probe", [(5, 4), (40, 24)])
def test_two_ranks_ivf_equal_single(k, n_probe):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(ivf_worker, args=(world, free_port(), k, n_probe, ret), nprocs=world, join=True)
    x, cent, assign, members, q = make_ivf_data()
    for rank in range(world):
        d, r, new, total = ret[rank]
        for i in range(len(q)):
            ref_r, ref_d = R.ivf_search(q[i], x, cent, members, k, n_probe, "euclidean")
            n = len(ref_r)
            np.testing.assert_array_equal(r[i, :n], ref_r)              # same rows in the same order, ties in visiting order
            np.testing.assert_array_equal(d[i, :n], ref_d.astype(np.float32))
            assert (r[i, n:] == -1).all()
        assert total.sum() == len(x)
        ref_new = np.stack([x[assign == l].astype(np.float64).mean(0) if (assign == l).any() else cent[l]
                            for l in range(len(cent))]).astype(np.float32)
        np.testing.assert_allclose(new, ref_new, rtol=1e-6, atol=1e-6)
